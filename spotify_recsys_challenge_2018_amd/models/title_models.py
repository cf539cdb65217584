"""The title scorer of the challenge path (reference models/title_get.py, models/title_models/Char_CNN.py;
citations relative to /root/reference) on the MI355X library: SURVEY.md 8f row 2.

    characters [B, 25] -> char_embedding [41, E] -> one wide convolution per filter size -> ReLU -> max over
    time -> concat (n_sizes * filter_num features) -> dropout -> sigmoid(features . Output_W + Output_b)

The per-title part is `dae_title_features` (csrc/title.hip); the vocabulary-wide output layer is the decoder
GEMM of the scoring path with hidden = the feature count, on a second library context that holds the prepacked
Output_W^T.  Variables keep their TensorFlow names (Char_CNN.py:20, :45-46, :69-70) so that a pickle of
`get_params()` reads like the graph; the reference itself saves a TF checkpoint (main_train.py:247), which this
environment cannot read or write -- the pickle is this package's own format.  Char_LSTM (the alternative
scorer, never selected by the shipped configs) is not implemented.
"""
import ctypes
import pickle

import os

import numpy as np

from .. import _lib


def get_model(conf):
    """models/title_get.py:11-26."""
    if conf.char_model != 'Char_CNN':
        raise NotImplementedError("char_model %r: only Char_CNN is implemented (the shipped configs use it)"
                                  % (conf.char_model,))
    return Char_CNN(conf, [[conf.filter_num, fs, -1] for fs in conf.filter_size])


class Char_CNN:
    def __init__(self, conf, conv_layers):
        self.embedding = int(conf.char_emb)                      # Char_CNN.py:9-13
        self.input_len = int(conf.strmaxlen)
        self.output_dim = int(conf.n_output)
        self.char_size = int(conf.charsize)
        self.conv_layers = conv_layers
        if self.embedding <= 0:
            raise NotImplementedError("char_emb = 0 (one-hot input, Char_CNN.py:26-28) is not implemented")
        if any(c[2] != -1 for c in conv_layers):
            raise NotImplementedError("intermediate max-pooling (Char_CNN.py:51-54) is not used by title_get.py")
        self.filter_sizes = [int(c[1]) for c in conv_layers]
        self.filter_num = int(conv_layers[0][0])
        self.n_feat = self.filter_num * len(self.filter_sizes)
        # feature row length the GEMM kernels take: a multiple of 64 lets the backward GEMMs own two hidden tiles
        # per wave (400 features -> 448; 416 measured 4 % slower, 512 no better)
        self.ld = (self.n_feat + 63) // 64 * 64
        self.device_index = int(getattr(conf, "device_index", 0))
        self.init_seed = int(getattr(conf, "title_init_seed", 0))
        self.learning_rate = float(getattr(conf, "title_lr", 0.001))
        self.ctx = None
        self.p = {}
        self._packed_dirty = True
        self._rng = np.random.RandomState(int(getattr(conf, "dropout_seed", 1234)) + 1)
        self._fs = (ctypes.c_int32 * len(self.filter_sizes))(*self.filter_sizes)

    # -- variables ------------------------------------------------------------------------------------
    def names(self):
        n = ["char_embedding"]
        for i in range(len(self.filter_sizes)):
            n += ["Conv_W%d" % i, "Conv_b%d" % i]
        return n + ["Output_W", "Output_b"]

    def _host_init(self):
        """tf.contrib.layers.xavier_initializer(uniform=False) for every variable, biases included (Char_CNN.py:19,
        :44-46, :68-70): a normal of stddev sqrt(2 / (fan_in + fan_out)) TRUNCATED at two standard deviations
        (TF redraws the samples outside; the stddev is that of the untruncated normal).  For a rank-1 variable TF
        takes fan_in = fan_out = its length: stddev sqrt(1 / n)."""
        rng = np.random.default_rng(self.init_seed)

        def xn(shape, fan_in, fan_out):
            x = rng.standard_normal(shape)
            bad = np.abs(x) > 2.0
            while bad.any():                                     # truncated normal: redraw the tails
                x[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(x) > 2.0
            return (x * np.sqrt(2.0 / (fan_in + fan_out))).astype(np.float32)
        E, F, V = self.embedding, self.filter_num, self.output_dim
        p = {"char_embedding": xn((self.char_size, E), self.char_size, E)}
        for i, fs in enumerate(self.filter_sizes):
            p["Conv_W%d" % i] = xn((fs, E, 1, F), fs * E, fs * E * F)
            p["Conv_b%d" % i] = xn((F,), F, F)
        p["Output_W"] = xn((self.n_feat, V), self.n_feat, V)
        p["Output_b"] = xn((V,), V, V)
        return p

    def fit(self, params=None):
        """Create the context and put the variables on the device (host dict in TF layout, or fresh)."""
        import torch
        self.ctx = _lib.Context(self.device_index)
        self.set_params(params if params is not None else self._host_init())

    def set_params(self, host):
        import torch
        dev = torch.device("cuda", self.device_index)
        E, F, V = self.embedding, self.filter_num, self.output_dim
        want = {"char_embedding": (self.char_size, E), "Output_W": (self.n_feat, V), "Output_b": (V,)}
        for i, fs in enumerate(self.filter_sizes):
            want["Conv_W%d" % i] = (fs, E, 1, F)
            want["Conv_b%d" % i] = (F,)
        for n, shp in want.items():
            if tuple(np.shape(host[n])) != shp:
                raise ValueError("title variable %s has shape %s, expected %s" % (n, np.shape(host[n]), shp))

        def dev_t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        self.p = {"char_embedding": dev_t(host["char_embedding"])}
        # the conv variables back to back, as dae_title_features takes them
        self.p["conv_w"] = dev_t(np.concatenate([np.asarray(host["Conv_W%d" % i], np.float32).reshape(-1)
                                                 for i in range(len(self.filter_sizes))]))
        self.p["conv_b"] = dev_t(np.concatenate([np.asarray(host["Conv_b%d" % i], np.float32).reshape(-1)
                                                 for i in range(len(self.filter_sizes))]))
        # Output_W^T, zero padded to the feature row length: the decoder layout [V, hidden]
        wt = np.zeros((V, self.ld), np.float32)
        wt[:, :self.n_feat] = np.asarray(host["Output_W"], np.float32).T
        self.p["Output_WT"] = dev_t(wt)
        self.p["Output_b"] = dev_t(host["Output_b"])
        self._packed_dirty = True
        self._params_gen = self.__dict__.get("_params_gen", 0) + 1      # (a cached dae_pipeline holds images of the old variables)
        self._drop_features_table()

    def get_params(self):
        """Host dict under the TF variable names and shapes."""
        E, F = self.embedding, self.filter_num
        out = {"char_embedding": self.p["char_embedding"].cpu().numpy()}
        cw, cb = self.p["conv_w"].cpu().numpy(), self.p["conv_b"].cpu().numpy()
        off = 0
        for i, fs in enumerate(self.filter_sizes):
            n = fs * E * F
            out["Conv_W%d" % i] = cw[off:off + n].reshape(fs, E, 1, F).copy()
            out["Conv_b%d" % i] = cb[i * F:(i + 1) * F].copy()
            off += n
        out["Output_W"] = self.p["Output_WT"].cpu().numpy()[:, :self.n_feat].T.copy()
        out["Output_b"] = self.p["Output_b"].cpu().numpy()
        return out

    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.get_params(), f)

    def load(self, path):
        with open(path, "rb") as f:
            self.set_params(pickle.load(f))

    # -- forward ----------------------------------------------------------------------------------------
    def _titles_dev(self, titles, n_rows, side_stream_of=None):
        import torch
        t = np.full((n_rows, self.input_len), -1, np.int32)
        src = np.asarray(titles, np.int64).reshape(-1, self.input_len) if len(titles) else np.zeros((0, self.input_len))
        t[:len(src)] = src[:n_rows]
        if side_stream_of is not None:                 # streamed scoring: upload on the model's copy stream
            return side_stream_of._to_dev(t, torch.int32, side_stream=True)
        return torch.from_numpy(t).to(torch.device("cuda", self.device_index))

    def _ensure_features_table(self):
        """Inference with the variables as they are NOW: the convolutions as a table over (filter size, offset, character)
        (dae_title_prepack_features) -- rebuilt whenever the variables changed since it was made."""
        if self.__dict__.get("_ftab_gen") != self._params_gen:
            P = _lib._ptr
            self.ctx.bind_stream()
            self.ctx.check(self.ctx.lib.dae_title_prepack_features(
                self.ctx.h, P(self.p["char_embedding"]), self.char_size, self.embedding, P(self.p["conv_w"]), self._fs,
                len(self.filter_sizes), self.filter_num))
            self._ftab_gen = self._params_gen

    def _drop_features_table(self):
        """The variables are about to change (a training step, set_params): no call may read the old table."""
        if self.__dict__.get("_ftab_gen") is not None and getattr(self, "ctx", None) is not None and self.ctx.h:
            self.ctx.check(self.ctx.lib.dae_title_prepack_features(self.ctx.h, None, 0, 0, None, None, 0, 0))
        self._ftab_gen = None

    def features(self, titles, n_rows, keep_prob=1.0, seed=0, keep_for_backward=False, side_stream_of=None, d_titles=None):
        """Char_CNN.py:23-63 -> feat [n_rows, ld] (CUDA); with keep_for_backward also (argmax, raw)."""
        import torch
        self.ctx.bind_stream()
        if keep_prob == 1.0 and not keep_for_backward:
            self._ensure_features_table()
        dev = self.p["conv_w"].device
        d_t = self._titles_dev(titles, n_rows, side_stream_of) if d_titles is None else d_titles    # (staged with the feed)
        feat = torch.empty((n_rows, self.ld), dtype=torch.float32, device=dev)
        arg = torch.empty((n_rows, self.n_feat), dtype=torch.int32, device=dev) if keep_for_backward else None
        raw = torch.empty((n_rows, self.n_feat), dtype=torch.float32, device=dev) if keep_for_backward else None
        P = _lib._ptr
        self.ctx.check(self.ctx.lib.dae_title_features(
            self.ctx.h, P(d_t), n_rows, self.input_len, P(self.p["char_embedding"]), self.char_size, self.embedding,
            P(self.p["conv_w"]), P(self.p["conv_b"]), self._fs, len(self.filter_sizes), self.filter_num,
            float(keep_prob), int(seed), P(feat), self.ld, P(arg), P(raw)))
        return (feat, d_t, arg, raw) if keep_for_backward else feat

    def _ensure_packed(self, dtype=_lib.DAE_DTYPE_F32, features_table=True):
        if features_table:                   # (scoring; a training step passes False: its forward keeps argmax / raw features)
            self._ensure_features_table()
        if self._packed_dirty:
            self._packed = set()
            self._packed_dirty = False
        if dtype not in self.__dict__.setdefault("_packed", set()):
            self.ctx.bind_stream()
            self.ctx.prepack_decoder(self.p["Output_WT"], self.p["Output_b"], 0, self.output_dim, dtype)
            self._packed.add(dtype)
            # "bf16" and "exact_bf16" share the context's bf16 image: the exact prepack serves both, a plain one drops the bounds
            if dtype == _lib.DAE_DTYPE_BF16_EXACT:
                self._packed.add(_lib.DAE_DTYPE_BF16)
            elif dtype == _lib.DAE_DTYPE_BF16:
                self._packed.discard(_lib.DAE_DTYPE_BF16_EXACT)

    def score(self, titles, n_rows, keep_prob=1.0, seed=0):
        """`model_title.output`: sigmoid(features . Output_W + Output_b) as a dense [n_rows, n_output] CUDA tensor."""
        import torch
        self._ensure_packed()
        feat = self.features(titles, n_rows, keep_prob, seed)
        out = torch.empty((n_rows, self.output_dim), dtype=torch.float32, device=feat.device)
        self.ctx.decode_dense(feat, out, apply_sigmoid=True)
        return out

    # -- training (the DAE is frozen: only these variables move, DAEs.py:165-171, :198) -----------------
    def _train_state(self):
        import torch
        if getattr(self, "_adam", None) is None:
            self._step = 0
            self._tvars = ["char_embedding", "conv_w", "conv_b", "Output_WT", "Output_b"]
            self._grads = {n: torch.zeros_like(self.p[n]) for n in self._tvars}
            self._adam = {n: (torch.zeros_like(self.p[n]), torch.zeros_like(self.p[n])) for n in self._tvars}
        return self._grads

    def backward_and_step(self, feat, d_titles, arg, raw, z_title, dae_score, y_csr, w_t, w_p, n_batch, keep_prob,
                          seed, cost_out):
        """Gradients of the mixed-score loss w.r.t. the title variables, then one TF1-Adam step on each."""
        import torch
        g = self._train_state()
        ctx, lib, P = self.ctx, self.ctx.lib, _lib._ptr
        B, V = z_title.shape
        dfeat = torch.empty((B, self.ld), dtype=torch.float32, device=feat.device)
        ctx.check(lib.dae_title_loss_backward(
            ctx.h, P(z_title), int(z_title.stride(0)), P(dae_score), int(dae_score.stride(0)),
            P(y_csr[0]), P(y_csr[1]), P(y_csr[2]), P(w_t), P(w_p), B, V, int(n_batch),
            P(feat), self.ld, P(self.p["Output_WT"]), P(g["Output_WT"]), P(g["Output_b"]), P(dfeat), P(cost_out)))
        ctx.check(lib.dae_title_conv_backward(
            ctx.h, P(d_titles), B, self.input_len, P(self.p["char_embedding"]), self.char_size, self.embedding,
            P(self.p["conv_w"]), self._fs, len(self.filter_sizes), self.filter_num, P(arg), P(raw), P(dfeat),
            self.ld, float(keep_prob), int(seed), P(g["char_embedding"]), P(g["conv_w"]), P(g["conv_b"])))
        self._step += 1
        for n in self._tvars:
            m, v = self._adam[n]
            ctx.check(lib.dae_adam_step(ctx.h, P(self.p[n]), P(m), P(v), P(g[n]), self.p[n].numel(),
                                        self.learning_rate, 0.9, 0.999, 1e-8, self._step))
        self._packed_dirty = True
        self._params_gen = self.__dict__.get("_params_gen", 0) + 1
        self._drop_features_table()

    def __str__(self):
        return '\n'.join(["Wide CNN", "Embedding Size : " + str(self.embedding),
                          "Number of Filters : " + str(self.filter_num), "Conv Layers : " + str(self.conv_layers)])
