"""TEST INFRASTRUCTURE ONLY -- literal dense numpy restatement of models/DAEs.py (reference).

PARITY UNPINNED at the TensorFlow boundary (see oracle/dae_oracle.c header): TensorFlow 1.x is not
available, the reference has no tests/vectors for this path.  This module follows the reference
statement by statement in the reference's own DENSE formulation, and is the yardstick the canonical
C oracle (dae_oracle.c) is checked against within fp32 re-association tolerance.

Citations are relative to /root/reference.
"""
import numpy as np

F = np.float32


def sparse_to_dense(positions, values, n_batch, n_input):
    """DAEs.py:33-35 -- tf.sparse_tensor_to_dense(validate_indices=False): ASSIGNMENT, entries
    applied in order so the LAST duplicate wins (assumption documented there)."""
    x = np.zeros((n_batch, n_input), dtype=F)
    positions = np.asarray(positions, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values, dtype=F).reshape(-1)
    if values.size == 1 and positions.shape[0] != 1:
        values = np.full(positions.shape[0], values[0], dtype=F)
    for (r, c), v in zip(positions, values):        # in order: last wins
        x[r, c] = v
    return x


def sigmoid(z):
    z = np.asarray(z, dtype=F)
    return (F(1.0) / (F(1.0) + np.exp(-z, dtype=F))).astype(F)


def forward(x, W_enc, b_enc, W_dec, b_dec, input_keep_mask=None, ikp=1.0,
            hidden_keep_mask=None, kp=1.0):
    """DAEs.py:40-42, 64-70, 73-77/141-145.  Masks are 0/1 arrays (None = keep all) standing in
    for floor(keep_prob + uniform)."""
    x = np.asarray(x, dtype=F)
    xd = x / F(ikp)
    if input_keep_mask is not None:
        xd = xd * input_keep_mask.astype(F)
    s = xd.sum(axis=1, keepdims=True, dtype=F)                      # :41
    xh = xd / (s + F(1e-10))                                        # :42
    h = sigmoid(xh @ W_enc + b_enc)                                 # :66-67
    h = h / F(kp)
    if hidden_keep_mask is not None:
        h = h * hidden_keep_mask.astype(F)                          # :68
    logits = h @ W_dec.T + b_dec                                    # :75 / :143
    return xh, h, logits.astype(F)


def loss_from_pred(y, y_pred, n_batch):
    """DAEs.py:98-100 (reg term added by the caller)."""
    y = np.asarray(y, dtype=F)
    L = -np.sum(y * np.log(y_pred + F(1e-10)) +
                F(0.55) * (F(1.0) - y) * np.log(F(1.0) - y_pred + F(1e-10)), axis=1, dtype=F)
    return F(L.sum(dtype=F) / F(n_batch))


def l2_loss(*tensors):
    """DAEs.py:79-82 / :147-150 -- tf.nn.l2_loss(t) = sum(t**2)/2."""
    return F(sum(float(np.sum(np.square(t.astype(np.float64)))) / 2.0 for t in tensors))


def grads(x, y, W_enc, b_enc, W_dec, b_dec, n_batch, tied, reg_lambda=0.0,
          input_keep_mask=None, ikp=1.0, hidden_keep_mask=None, kp=1.0):
    """Hand-derived gradient of DAEs.py:98-100 cost w.r.t. d_params (float64 internally so it is
    a trustworthy reference for the fp32 GPU kernels)."""
    D = np.float64
    x = np.asarray(x, D); y = np.asarray(y, D)
    We = W_enc.astype(D); be = b_enc.astype(D); Wd = W_dec.astype(D); bd = b_dec.astype(D)
    xd = x / ikp
    if input_keep_mask is not None:
        xd = xd * input_keep_mask
    s = xd.sum(axis=1, keepdims=True)
    xh = xd / (s + 1e-10)
    pre = xh @ We + be
    sg = 1.0 / (1.0 + np.exp(-pre))
    hm = np.ones_like(sg) if hidden_keep_mask is None else hidden_keep_mask.astype(D)
    h = sg / kp * hm
    z = h @ Wd.T + bd
    p = 1.0 / (1.0 + np.exp(-z))
    eps = 1e-10
    L = -np.sum(y * np.log(p + eps) + 0.55 * (1 - y) * np.log(1 - p + eps), axis=1)
    cost = L.sum() / n_batch
    dLdp = -(y / (p + eps) - 0.55 * (1 - y) / (1 - p + eps))
    dz = dLdp * p * (1 - p) / n_batch
    gWd = dz.T @ h
    gbd = dz.sum(axis=0)
    dh = dz @ Wd
    dpre = dh * hm / kp * sg * (1 - sg)
    gWe = xh.T @ dpre
    gbe = dpre.sum(axis=0)
    if tied:
        gWe = gWe + gWd
        gWd = None
        cost += reg_lambda * 0.5 * ((We ** 2).sum() + (bd ** 2).sum() + (be ** 2).sum())
        gWe = gWe + reg_lambda * We
    else:
        cost += reg_lambda * 0.5 * ((We ** 2).sum() + (bd ** 2).sum() + (be ** 2).sum()
                                    + (Wd ** 2).sum())
        gWe = gWe + reg_lambda * We
        gWd = gWd + reg_lambda * Wd
    gbd = gbd + reg_lambda * bd
    gbe = gbe + reg_lambda * be
    return dict(cost=cost, gW_enc=gWe, gb_enc=gbe, gW_dec=gWd, gb_dec=gbd, y_pred=p, h=h)


def adam_tf(p, m, v, g, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (DAEs.py:102) as TF1's ApplyAdam functor computes it, all in fp32
    (epsilon OUTSIDE the bias correction, dense update):
        alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t)      (beta powers = fp32 running products)
        m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  p -= (m * alpha) / (sqrt(v) + eps)
    Returns new (p, m, v)."""
    p = p.astype(F); m = m.astype(F); v = v.astype(F); g = g.astype(F)
    b1, b2 = F(beta1), F(beta2)
    b1p, b2p = F(1.0), F(1.0)
    for _ in range(int(t)):
        b1p = F(b1p * b1); b2p = F(b2p * b2)
    alpha = F(F(lr) * np.sqrt(F(1.0) - b2p, dtype=F) / (F(1.0) - b1p))
    m = (m + (g - m) * (F(1.0) - b1)).astype(F)
    v = (v + (g * g - v) * (F(1.0) - b2)).astype(F)
    p = (p - (m * alpha) / (np.sqrt(v, dtype=F) + F(eps))).astype(F)
    return p, m, v


def cand_generate(scores, seed, k=500):
    """main_challenge.py:28-36 / metrics.py:59-68, literally (slow: list.remove)."""
    cand = np.argsort(-1 * scores)
    cand = cand.tolist()
    for i in seed:
        try:
            cand.remove(i)
        except ValueError:
            pass
    return cand[:k]


def topk_valid_under_reference_rule(scores_row, seed, picked, k=500):
    """Set-parity check (SURVEY finding 5): `picked` is a valid reference answer iff it contains no
    seed, has min(k, available) entries, and every picked score >= every non-picked non-seed
    score.  Returns (ok, boundary_tie) where boundary_tie says the k-th and (k+1)-th scores tie
    (then several sets are equally valid under numpy's unspecified tie order)."""
    n = scores_row.shape[0]
    seedset = set(int(s) for s in seed if 0 <= int(s) < n)
    picked = [int(p) for p in picked if p >= 0]
    avail = n - len(seedset)
    if len(picked) != min(k, avail) or len(set(picked)) != len(picked):
        return False, False
    if seedset & set(picked):
        return False, False
    mask = np.ones(n, dtype=bool)
    mask[list(seedset)] = False
    mask[picked] = False
    lo_picked = scores_row[picked].min() if picked else np.inf
    hi_rest = scores_row[mask].max() if mask.any() else -np.inf
    return bool(lo_picked >= hi_rest), bool(lo_picked == hi_rest)


class NumpyTrainStages:
    """TEST INFRASTRUCTURE: float64 restatement of the three stages of the vocabulary-sharded
    training step (include/dae_hip.h dae_train_shard_*), dense formulation, keep
    probabilities 1.0 only.  Stands in for the device kernels in the gloo tests of
    sharding.ShardedTrainer and is the yardstick of the GPU stage tests.  Tensors are torch CPU
    tensors (written in place), matching HipTrainStages' call signature."""

    def __init__(self, V):
        self.V = V

    def _xhat(self, x, B):
        rp, col, val = (np.asarray(t) for t in x)
        xd = np.zeros((B, self.V), np.float64)
        for r in range(B):
            xd[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
        return xd / (xd.sum(axis=1, keepdims=True) + 1e-10)                   # DAEs.py:41-42

    def encode(self, x, W_enc, lo, hi, ikp, seed, pre):
        assert ikp == 1.0
        xh = self._xhat(x, pre.shape[0])
        pre.copy_(_t(xh[:, lo:hi] @ W_enc.numpy().astype(np.float64)))        # DAEs.py:66

    def decode(self, pre, b_enc, y, W_enc, W_dec, b_dec, lo, hi, n_batch, tied, kp, seed, lam,
               gW_out, gb_dec, dh, cost):
        assert kp == 1.0
        D = np.float64
        B = pre.shape[0]
        Wd = (W_enc if tied else W_dec).numpy().astype(D)
        sg = 1.0 / (1.0 + np.exp(-(pre.numpy().astype(D) + b_enc.numpy().astype(D))))   # :67
        rp, col, val = (np.asarray(t) for t in y)
        yd = np.zeros((B, self.V), D)
        for r in range(B):
            yd[r, col[rp[r]:rp[r + 1]]] = val[rp[r]:rp[r + 1]]
        yd = yd[:, lo:hi]
        p = 1.0 / (1.0 + np.exp(-(sg @ Wd.T + b_dec.numpy().astype(D))))               # :75 / :143
        L = -np.sum(yd * np.log(p + 1e-10) + 0.55 * (1 - yd) * np.log(1 - p + 1e-10))   # :98-99
        dz = -(yd / (p + 1e-10) - 0.55 * (1 - yd) / (1 - p + 1e-10)) * p * (1 - p) / n_batch
        l2 = 0.5 * ((W_enc.numpy().astype(D) ** 2).sum() + (b_dec.numpy().astype(D) ** 2).sum())
        if lo == 0:
            l2 += 0.5 * (b_enc.numpy().astype(D) ** 2).sum()
        if not tied:
            l2 += 0.5 * (Wd ** 2).sum()
        cost.copy_(_t(np.array([L / n_batch + lam * l2])))
        gW_out.copy_(_t(dz.T @ sg))
        gb_dec.copy_(_t(dz.sum(axis=0)))
        dh.copy_(_t(dz @ Wd))
        self._sg = sg

    def finish(self, dh, x, W_enc, b_enc, W_dec, b_dec, lo, hi, tied, ikp, kp, seed, lam,
               gW_enc, gb_enc, gW_dec, gb_dec):
        D = np.float64
        sg = self._sg
        dpre = dh.numpy().astype(D) * sg * (1 - sg)
        gb_enc.copy_(_t(dpre.sum(axis=0) + lam * b_enc.numpy().astype(D)))
        ge = self._xhat(x, dh.shape[0])[:, lo:hi].T @ dpre
        if tied:
            ge = ge + gW_enc.numpy().astype(D)
        gW_enc.copy_(_t(ge + lam * W_enc.numpy().astype(D)))
        if not tied:
            gW_dec.copy_(_t(gW_dec.numpy().astype(D) + lam * W_dec.numpy().astype(D)))
        gb_dec.copy_(_t(gb_dec.numpy().astype(D) + lam * b_dec.numpy().astype(D)))

    def adam(self, p, m, v, g, lr, t):
        p2, m2, v2 = adam_tf(p.numpy(), m.numpy(), v.numpy(), g.numpy(), lr, t)
        p.copy_(_t(p2)); m.copy_(_t(m2)); v.copy_(_t(v2))


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
