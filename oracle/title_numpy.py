"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's title scorer and of the DAE_title
mix (citations relative to /root/reference):

  models/title_models/Char_CNN.py:6-75   character embedding -> parallel "wide" convolutions over the
                                         title (one per filter size, VALID) -> ReLU -> max over time ->
                                         concat -> dropout -> sigmoid(features . Output_W + Output_b)
  models/DAEs.py:153-181 (App. B.6)      y = title_score * w_title + dae_score * w_playlist with
                                         x_count = row_sum * input_keep_prob,
                                         w_title = u / (u + x_count + 1e-10), w_playlist = x_count / (same)

PARITY UNPINNED like the rest of the TensorFlow boundary (oracle/dae_oracle.c header).  One documented
assumption: title indices are padded with -1 (spotify_reader.py:36) and tf.nn.embedding_lookup on the
GPU returns a ZERO vector for an out-of-range id (the CPU kernel raises); padding therefore embeds to 0.
Float64 internally so that it is a trustworthy yardstick for the fp32 kernels (gradients included).
"""
import numpy as np


def xavier_normal(rng, shape, fan_in, fan_out):
    """tf.contrib.layers.xavier_initializer(uniform=False): N(0, 2 / (fan_in + fan_out)), truncated at 2 sd
    in TF; the restatement draws a plain normal (only used to make test weights)."""
    return (rng.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out))).astype(np.float32)


def make_params(n_char, emb, filter_sizes, filter_num, n_output, seed=0):
    """The variables Char_CNN.py creates, under their TF names."""
    rng = np.random.default_rng(seed)
    p = {"char_embedding": xavier_normal(rng, (n_char, emb), n_char, emb)}
    for i, fs in enumerate(filter_sizes):
        p["Conv_W%d" % i] = xavier_normal(rng, (fs, emb, 1, filter_num), fs * emb, fs * emb * filter_num)
        p["Conv_b%d" % i] = xavier_normal(rng, (filter_num,), filter_num, 1)
    d = filter_num * len(filter_sizes)
    p["Output_W"] = xavier_normal(rng, (d, n_output), d, n_output)
    p["Output_b"] = xavier_normal(rng, (n_output,), n_output, 1)
    return p


def embed(titles, E):
    titles = np.asarray(titles, dtype=np.int64)
    ok = (titles >= 0) & (titles < E.shape[0])
    out = E.astype(np.float64)[np.where(ok, titles, 0)]
    out[~ok] = 0.0                                                   # padding (-1) embeds to zero
    return out                                                       # [B, L, emb]


def features(titles, params, filter_sizes, return_argmax=False):
    """Char_CNN.py:31-62 -> [B, n_sizes * filter_num] (before dropout)."""
    x = embed(titles, params["char_embedding"])
    B, L, _ = x.shape
    feats, args = [], []
    for i, fs in enumerate(filter_sizes):
        W = params["Conv_W%d" % i].astype(np.float64)[:, :, 0, :]   # [fs, emb, F]
        b = params["Conv_b%d" % i].astype(np.float64)
        P = L - fs + 1
        conv = np.stack([np.einsum("bdc,dcf->bf", x[:, p:p + fs, :], W) for p in range(P)], axis=1) + b
        conv = np.maximum(conv, 0.0)                                 # :49-50
        feats.append(conv.max(axis=1))                               # :56 one-max pooling
        args.append(conv.argmax(axis=1))
    f = np.concatenate(feats, axis=1)
    return (f, np.concatenate(args, axis=1)) if return_argmax else f


def forward(titles, params, filter_sizes, keep_mask=None, keep_prob=1.0):
    """-> (features after dropout, logits, title_score) ; Char_CNN.py:64-72."""
    f = features(titles, params, filter_sizes) / keep_prob
    if keep_mask is not None:
        f = f * keep_mask
    z = f @ params["Output_W"].astype(np.float64) + params["Output_b"].astype(np.float64)
    return f, z, 1.0 / (1.0 + np.exp(-z))


def mix_weights(row_sum, input_keep_prob, titles_use):
    """DAEs.py:159-162.  Returns (w_title, w_playlist) as float32 columns, computed in fp32 like the graph."""
    F = np.float32
    x_count = np.asarray(row_sum, F).reshape(-1, 1) * F(input_keep_prob)
    u = np.asarray(titles_use, F).reshape(-1, 1)
    deno = u + x_count + F(1e-10)
    return (u / deno).astype(F), (x_count / deno).astype(F)


def mix(title_score, dae_score, w_title, w_playlist):
    """DAEs.py:180."""
    return title_score * w_title + dae_score * w_playlist


def grads(titles, params, filter_sizes, dae_score, y, w_title, w_playlist, n_batch, keep_mask=None, keep_prob=1.0):
    """Gradients of DAEs.py:193-195 (weighted BCE of the MIXED score, mean over n_batch) w.r.t. the title
    variables only (the DAE constants are frozen, DAEs.py:165-171).  float64."""
    x = embed(titles, params["char_embedding"])
    B, L, Ec = x.shape
    f0, arg = features(titles, params, filter_sizes, return_argmax=True)
    km = np.ones_like(f0) if keep_mask is None else keep_mask.astype(np.float64)
    f = f0 / keep_prob * km
    Wo = params["Output_W"].astype(np.float64)
    z = f @ Wo + params["Output_b"].astype(np.float64)
    st = 1.0 / (1.0 + np.exp(-z))
    wt = np.asarray(w_title, np.float64); wp = np.asarray(w_playlist, np.float64)
    yp = st * wt + np.asarray(dae_score, np.float64) * wp
    eps = 1e-10
    yv = np.asarray(y, np.float64)
    cost = -np.sum(yv * np.log(yp + eps) + 0.55 * (1 - yv) * np.log(1 - yp + eps)) / n_batch
    dyp = -(yv / (yp + eps) - 0.55 * (1 - yv) / (1 - yp + eps)) / n_batch
    dz = dyp * wt * st * (1 - st)
    g = {"Output_W": f.T @ dz, "Output_b": dz.sum(axis=0)}
    df0 = (dz @ Wo.T) * km / keep_prob
    gE = np.zeros_like(params["char_embedding"], dtype=np.float64)
    titles = np.asarray(titles, np.int64)
    F = params["Conv_b0"].shape[0]
    for i, fs in enumerate(filter_sizes):
        W = params["Conv_W%d" % i].astype(np.float64)[:, :, 0, :]
        gW = np.zeros_like(W); gb = np.zeros(F)
        for b in range(B):
            for fi in range(F):
                d = df0[b, i * F + fi]
                if f0[b, i * F + fi] <= 0.0 or d == 0.0:            # ReLU gate (max of relu: 0 -> no gradient)
                    continue
                p = arg[b, i * F + fi]
                gW[:, :, fi] += d * x[b, p:p + fs, :]
                gb[fi] += d
                for dp in range(fs):
                    t = titles[b, p + dp]
                    if 0 <= t < gE.shape[0]:
                        gE[t] += d * W[dp, :, fi]
        g["Conv_W%d" % i] = gW[:, :, None, :]
        g["Conv_b%d" % i] = gb
    g["char_embedding"] = gE
    return cost, g, dict(features=f0, argmax=arg, z=z, title_score=st, y_pred=yp)
