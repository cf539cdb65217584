"""Host-side mirror of the reference's models/DAEs.py object protocol on top of the
C ABI of libdae_hip.so.  PyTorch-ROCm is used for device memory and streams only; every hot
operation is a hand-written HIP kernel behind include/dae_hip.h.

Reference protocol (models/DAEs.py, all citations relative to /root/reference):
    model = DAE_tied(conf) | DAE(conf)          # :13 / :114   reads conf.save/batch/n_input/...
    model.fit()                                  # :84          builds the graph
    sess.run(model.init_op)                      # variables -> device
    sess.run(model.y_pred, feed_dict={model.x_positions:..., model.x_ones:...,
             model.keep_prob: 1.0, model.input_keep_prob: 1.0})          # main_train.py:66-68
    sess.run([model.optimizer, model.cost], feed_dict={... y_positions, y_ones, keep_prob,
             input_keep_prob})                                           # main_train.py:204-207
    model.save_model(sess)                       # :107-111  pickle [enc_W, dec_W, enc_b, dec_b]

The same calls work here through `Session` below.  Direct methods (`predict`, `recommend`,
`train_step`) are what the drivers of this repo use; `recommend` is the fused
encode -> decode -> top-k path that never materialises the [batch, n_input] matrix.
"""
import ctypes
import pickle

import numpy as np

from .. import _lib


# [BASE] decode_dtype values -> the C ABI's decode arithmetic (include/dae_hip.h)
_DECODE_DTYPES = {"f32": _lib.DAE_DTYPE_F32, "bf16": _lib.DAE_DTYPE_BF16, "exact_bf16": _lib.DAE_DTYPE_BF16_EXACT}


def _title_dtype(dtype, model=None):
    """Arithmetic of a title-mixed launch.  exact_bf16 has its own two-GEMM path (dae_mix_topk_exact, csrc/mixexact.hip)
    built for the shipped shapes -- DAE hidden 256, title feature rows of 448; any other model runs its titled launches
    on the fp32 kernels (the lists exact_bf16 promises are the fp32 lists either way)."""
    if dtype != _lib.DAE_DTYPE_BF16_EXACT:
        return dtype
    tm = getattr(model, "title_model", None)
    if (tm is not None and model.n_hidden == 256 and tm.ld == 448 and not getattr(model, "title_exact_off", False)):
        return dtype
    if tm is not None and not getattr(model, "title_exact_off", False) and not model.__dict__.get("_title_fp32_said"):
        # [DAE] hidden, [TITLE] filter_num / filter_size are free keys of the schema (main.py:46, :73-75): say ONCE that this
        # model's titled launches cost the fp32 kernels' time (about 6 x), instead of silently
        model.__dict__["_title_fp32_said"] = True
        import sys
        print("[dae] decode_dtype = exact_bf16: the exact title mix is built for hidden = 256 and 448-wide title feature rows; "
              "this model has hidden = %d, %d-wide rows -- its titled launches run on the fp32 kernels (same lists, slower)"
              % (model.n_hidden, tm.ld), file=sys.stderr)
    return _lib.DAE_DTYPE_F32


class _Placeholder:
    """Stands in for a tf.placeholder: only used as a feed_dict key."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<placeholder %s>" % self.name


class _Fetch:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<fetch %s>" % self.name


SEEDS_FROM_INPUT = "input"      # recommend(seeds=SEEDS_FROM_INPUT): the seeds are the track columns of the input feed


def coo_to_csr(positions, values, n_rows, n_cols=None):
    """DAEs.py:33-35 semantics for the kernels: the reference scatters COO (row, col) -> value
    into a dense matrix by ASSIGNMENT with validate_indices=False, so duplicates are the norm and
    the LAST occurrence wins.  Returns CSR with columns ascending per row:
    (row_ptr int32 [n_rows+1], col int32 [nnz], val float32 [nnz]).  Explicit zeros are kept out
    (a zero entry contributes nothing to the row sum or the gather)."""
    pos = np.asarray(positions, dtype=np.int64).reshape(-1, 2)
    n = pos.shape[0]
    vals = np.asarray(values, dtype=np.float32).reshape(-1)
    if vals.size == 1 and n != 1:
        vals = np.full(n, vals[0], dtype=np.float32)
    if vals.size != n:
        raise ValueError("positions (%d) and values (%d) differ in length" % (n, vals.size))
    if n == 0:
        return (np.zeros(n_rows + 1, dtype=np.int32), np.zeros(0, dtype=np.int32),
                np.zeros(0, dtype=np.float32))
    rows, cols = pos[:, 0], pos[:, 1]
    if rows.min() < 0 or rows.max() >= n_rows:
        raise ValueError("row index out of range [0,%d)" % n_rows)
    if cols.min() < 0 or (n_cols is not None and cols.max() >= n_cols):
        raise ValueError("column index out of range")
    order = np.lexsort((np.arange(n), cols, rows))          # row, col, then feed order
    r, c, v = rows[order], cols[order], vals[order]
    last = np.ones(n, dtype=bool)
    last[:-1] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])        # last duplicate of each (row, col)
    keep = last & (v != 0.0)
    r, c, v = r[keep], c[keep], v[keep]
    row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(row_ptr, r + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    return row_ptr.astype(np.int32), c.astype(np.int32), v.astype(np.float32)


def seeds_to_csr(seeds, n_rows, n_tracks):
    """Per-row seed track lists (main_challenge.py:31-35 `cand.remove(i)`) -> CSR of sorted unique
    in-range track ids.  One lexsort over all rows (the per-row np.unique loop was the largest host cost
    of a scoring call)."""
    import itertools
    used = seeds[:n_rows] if len(seeds) > n_rows else seeds
    lens = np.zeros(n_rows, dtype=np.int64)
    lens[:len(used)] = np.fromiter(map(len, used), dtype=np.int64, count=len(used))
    total = int(lens.sum())
    if total == 0:
        return np.zeros(n_rows + 1, dtype=np.int32), np.zeros(0, dtype=np.int32)
    flat = np.fromiter(itertools.chain.from_iterable(used), dtype=np.int64, count=total)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lens)
    ok = (flat >= 0) & (flat < n_tracks)
    flat, rows = flat[ok], rows[ok]
    order = np.lexsort((flat, rows))
    flat, rows = flat[order], rows[order]
    first = np.ones(flat.size, dtype=bool)
    first[1:] = (rows[1:] != rows[:-1]) | (flat[1:] != flat[:-1])
    flat, rows = flat[first], rows[first]
    row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(np.bincount(rows, minlength=n_rows))
    return row_ptr.astype(np.int32), flat.astype(np.int32)


class DAE_tied:
    """Tied-weight denoising autoencoder (reference DAEs.py:13-111)."""

    tied = True

    def __init__(self, conf):
        self.save_dir = conf.save                       # DAEs.py:15
        self.n_batch = int(conf.batch)                  # :17
        self.n_input = int(conf.n_input)                # :18
        self.n_hidden = int(conf.hidden)                # :19
        self.learning_rate = float(conf.lr)             # :20
        self.reg_lambda = float(conf.reg_lambda)        # :21
        self.n_tracks = int(getattr(conf, "n_tracks", self.n_input))
        self.device_index = int(getattr(conf, "device_index", 0))
        self.init_seed = int(getattr(conf, "init_seed", 0))

        # feed_dict keys (DAEs.py:23-30)
        self.x_positions = _Placeholder("x_positions")
        self.x_ones = _Placeholder("x_ones")
        self.y_positions = _Placeholder("y_positions")
        self.y_ones = _Placeholder("y_ones")
        self.keep_prob = _Placeholder("keep_prob")
        self.input_keep_prob = _Placeholder("input_keep_prob")

        self.y_pred = None
        self.cost = None
        self.optimizer = None
        self.init_op = None
        self.weights = {}
        self.biases = {}
        self.d_params = []
        self.ctx = None
        self._adam = None
        self._step = 0
        self._packed_dirty = {_lib.DAE_DTYPE_F32: True, _lib.DAE_DTYPE_BF16: True, _lib.DAE_DTYPE_BF16_EXACT: True}
        self._packed_cols = {}
        # decode arithmetic of recommend(): "f32" (fp32 MFMA, bit-exact path), "bf16" (BASELINE configs[4]) or
        # "exact_bf16" (BASELINE north_star: bf16 MFMA filter + fp32 recomputation of the survivors -- the fp32
        # path's lists, bit for bit, at close to the bf16 rate)
        dd = str(getattr(conf, "decode_dtype", "f32"))
        if dd not in _DECODE_DTYPES:
            raise ValueError("decode_dtype %r: one of %s" % (dd, sorted(_DECODE_DTYPES)))
        self.decode_dtype = _DECODE_DTYPES[dd]
        # arithmetic of the training forward GEMM: "f32" (default) or "bf16" (BASELINE configs[3]: bf16 operands,
        # fp32 accumulate; loss, backward GEMMs, parameters and Adam stay fp32)
        self.train_dtype = _lib.DAE_DTYPE_BF16 if str(getattr(conf, "train_dtype", "f32")) == "bf16" \
            else _lib.DAE_DTYPE_F32
        # untied model, reg_lambda == 0: the encoder's gradient is row-sparse, and its dense Adam runs through
        # dae_adam_rows_* (bit-identical parameters, no HBM passes over the rows without gradient); "dense" keeps
        # dae_adam_step on the whole matrix
        self.encoder_adam = str(getattr(conf, "encoder_adam", "rows"))
        self.decoder_adam = str(getattr(conf, "decoder_adam", "fused"))        # "fused" (dae_arm_decoder_adam) | "dense"
        # ... and every `rows_adam_flush_every` steps all rows are brought up to date, which bounds how many missed
        # steps a rarely seen row has to replay when it is next named (a replay is sequential per element)
        self.rows_adam_flush_every = int(getattr(conf, "rows_adam_flush_every", 32))
        self.rows_adam_table = int(getattr(conf, "rows_adam_table", 1 << 16))      # alpha per step; doubles when full
        self._lazy = None
        self._rng = np.random.RandomState(int(getattr(conf, "dropout_seed", 1234)))
        self.device_csr = bool(getattr(conf, "device_csr", True))
        self._csr_status = None
        self._sharded = None          # sharding.ShardedTrainer when training is row-sharded over ranks
        self._params_stale = False
        self._score_shard = None      # shard_scoring(): this rank's vocabulary columns + the ShardedRanker

    # -- parameters ---------------------------------------------------------------------------------
    def _xavier(self, rng, shape):
        """tf.contrib.layers.xavier_initializer() (uniform): U(+-sqrt(6/(fan_in+fan_out)))."""
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def _host_init(self):
        """DAEs.py:53-61: encoder_h Xavier, biases zero; decoder IS the encoder matrix."""
        rng = np.random.default_rng(self.init_seed)
        W = self._xavier(rng, (self.n_input, self.n_hidden))
        return [W, W, np.zeros(self.n_hidden, np.float32), np.zeros(self.n_input, np.float32)]

    def init_weight(self):
        import torch
        dev = torch.device("cuda", self.device_index)
        enc_W, dec_W, enc_b, dec_b = self._host_init()
        for a, shp in ((enc_W, (self.n_input, self.n_hidden)), (dec_W, (self.n_input, self.n_hidden)),
                       (enc_b, (self.n_hidden,)), (dec_b, (self.n_input,))):
            if tuple(a.shape) != shp:
                raise ValueError("initial value has shape %s, expected %s" % (a.shape, shp))
        self.weights["encoder_h"] = torch.from_numpy(np.ascontiguousarray(enc_W, np.float32)).to(dev)
        if self.tied:
            self.weights["decoder_h"] = self.weights["encoder_h"]
        else:
            self.weights["decoder_h"] = torch.from_numpy(np.ascontiguousarray(dec_W, np.float32)).to(dev)
        self.biases["encoder_b"] = torch.from_numpy(np.ascontiguousarray(enc_b, np.float32)).to(dev)
        self.biases["decoder_b"] = torch.from_numpy(np.ascontiguousarray(dec_b, np.float32)).to(dev)
        self.d_params = [self.weights["encoder_h"], self.weights["decoder_h"],
                         self.biases["encoder_b"], self.biases["decoder_b"]]
        self._mark_dirty()

    def fit(self):
        """DAEs.py:84-105.  Creates the device context, the parameters and the fetch handles."""
        self.ctx = _lib.Context(self.device_index)
        if self.train_dtype != _lib.DAE_DTYPE_F32:
            self.ctx.set_train_dtype(self.train_dtype)
        self.init_weight()
        self.y_pred = _Fetch("y_pred")
        self.cost = _Fetch("cost")
        self.optimizer = _Fetch("optimizer")
        self.init_op = _Fetch("init_op")
        self.hidden = _Fetch("hidden")

    # -- helpers ------------------------------------------------------------------------------------
    def _to_dev(self, a, dtype, side_stream=False):
        """Host array -> device tensor.  side_stream (the training loop): uploaded on a COPY stream.  A copy from pageable memory is stream-ordered
        and blocks the host until it has happened; on the compute stream that means "until everything queued before
        it has run", so the training loop could never get ahead of the GPU (0.25 ms per feed measured).  On its own
        stream the copy waits for earlier copies only; the compute stream waits for its event.  (The scoring loop
        stages its feeds inside the library: csrc/pipeline.hip.)"""
        import torch
        src = torch.from_numpy(np.ascontiguousarray(a))
        dev = torch.device("cuda", self.device_index)
        if not side_stream:                  # calls that fetch their result anyway (recommend, predict, ...)
            return src.to(dev, dtype=dtype, non_blocking=False)
        cs = self.__dict__.get("_copy_stream")
        if cs is None:
            cs = self._copy_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(self.device_index)
        with torch.cuda.stream(cs):
            t = src.to(dev, dtype=dtype, non_blocking=False)
            ev = cs.record_event()
        cur.wait_event(ev)
        t.record_stream(cur)
        return t

    @staticmethod
    def _feed_arrays(positions, values):
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.int64).reshape(-1, 2))
        vals = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
        if vals.size != 1 and vals.size != pos.shape[0]:
            raise ValueError("positions (%d) and values (%d) differ in length" % (pos.shape[0], vals.size))
        return pos, vals

    def _upload_csr(self, positions, values, side_stream=False, ctx=None, n_rows=None):
        """The feed (COO in feed order, duplicates allowed) -> device CSR.  Default: upload the raw feed
        and build the CSR on the GPU (dae_coo_to_csr, csrc/csr.hip); `device_csr = False` keeps the numpy
        restatement `coo_to_csr` (same result entry for entry; it also range-checks eagerly)."""
        import torch
        if self.device_csr:
            pos, vals = self._feed_arrays(positions, values)
            d_pos = self._to_dev(pos if pos.shape[0] else np.zeros((1, 2), np.int64), torch.int64, side_stream)[:pos.shape[0]]
            d_val = self._to_dev(vals if vals.size else np.zeros(1, np.float32), torch.float32, side_stream)
            rp, c, v, status = (ctx or self.ctx).coo_to_csr(d_pos, d_val, n_rows or self.n_batch, self.n_input)
            # checked lazily (no sync on the scoring path).  A flag is written on the stream that is current NOW -- the
            # main stream or the second scoring lane's -- so it travels with an event recorded behind its writer: whoever
            # reads it later, on whichever stream, waits for exactly that
            cur = torch.cuda.current_stream(self.device_index)
            pending = (self._csr_status or []) + [(status, cur.record_event())]
            if len(pending) > 64:                               # un-fetched training steps: fold on the device
                pending = [(self._fold_status(pending), cur.record_event())]
            self._csr_status = pending
            return rp, c, v
        rp, c, v = coo_to_csr(positions, values, n_rows or self.n_batch, self.n_input)
        if c.size == 0:          # keep valid device pointers for empty batches
            c = np.zeros(1, np.int32); v = np.zeros(1, np.float32)
        return self._to_dev(rp, torch.int32), self._to_dev(c, torch.int32), self._to_dev(v, torch.float32)

    def _fold_status(self, pending):
        """max over the pending range flags, on the current stream, ordered behind every flag's writer."""
        import torch
        cur = torch.cuda.current_stream(self.device_index)
        for st, ev in pending:
            cur.wait_event(ev)
            st.record_stream(cur)                               # allocated on its writer's stream, read (and freed) here
        return torch.cat([st for st, _ev in pending]).max().reshape(1)

    def _check_feed(self):
        """After the results of a call have been fetched: the device CSR builder skips entries whose row / column is
        out of range and raises the flag checked here."""
        if self._csr_status:
            bad = int(self._fold_status(self._csr_status).item())
            self._csr_status = None
            if bad:
                raise ValueError("feed holds a row or column index out of range (rows [0,%d) of each feed, columns [0,%d))"
                                 % (self.n_batch, self.n_input))

    check_feed = _check_feed

    def _mark_dirty(self):
        """The packed decoder images of the context(s) no longer hold the current weights."""
        self._packed_dirty = {_lib.DAE_DTYPE_F32: True, _lib.DAE_DTYPE_BF16: True, _lib.DAE_DTYPE_BF16_EXACT: True}
        self._packed_cols = {}
        self._weights_gen = self.__dict__.get("_weights_gen", 0) + 1

    # -- multi-GPU scoring (BASELINE.json configs[2]): vocabulary columns sharded over the ranks ----
    def shard_scoring(self, rank, world, group=None, exchange="allgather", tau_exchange=True):
        """`recommend` through sharding.ShardedRanker: this rank decodes and ranks only the vocabulary columns
        `sharding.scoring_shard(n_tracks, n_input, world, rank)` (W_enc is replicated, so every rank computes the same hidden
        activations with no collective), the per-shard top-k lists meet in ONE exchange (RCCL all-gather, or an
        all-to-all of the rows each rank owns) and are merged with the same key -- exact, since a shard's top-k
        holds its share of the global top-k.  Every rank feeds the same batches.

        exchange = "allgather": `recommend` returns all rows on every rank (main_challenge: rank 0 writes).
        exchange = "alltoall":  `recommend` returns the rows this rank owns (`owned_rows()`); n_batch is rounded
        up to a multiple of the world size (the extra rows are empty).
        tau_exchange: the shards' thresholds meet before the filter launches (one more collective of 4 bytes per row
        and rank; sharding.ShardedRanker "threshold exchange"): same lists, less selection work per shard."""
        from ..sharding import scoring_shard
        if exchange not in ("allgather", "alltoall"):
            raise ValueError("unknown exchange %r" % (exchange,))
        if exchange == "alltoall" and self.n_batch % world:
            if self.ctx is not None:
                raise _lib.DaeError("shard_scoring(alltoall) must round n_batch up before fit()")
            self.n_batch += world - self.n_batch % world
        shard = scoring_shard(self.n_tracks, self.n_input, world, rank)
        self._score_shard = {"rank": int(rank), "world": int(world), "group": group, "exchange": exchange,
                             "cols": shard, "rank_bound": shard[0][1], "rankers": {}, "tau_exchange": bool(tau_exchange)}
        self._mark_dirty()

    def owned_rows(self):
        """Rows of a batch whose final top-k `recommend` returns on this rank: all of them, except under
        shard_scoring(exchange="alltoall")."""
        sh = self._score_shard
        if sh is None or sh["exchange"] != "alltoall":
            return 0, self.n_batch
        from ..sharding import row_owner_bounds
        return row_owner_bounds(self.n_batch, sh["world"], sh["rank"])

    def _shard_ranker(self, dtype):
        sh = self._score_shard
        if dtype not in sh["rankers"]:
            from ..sharding import HipRankStages, ShardedRanker
            # the rank's image holds its track slice under the real ids, then its artist slice: rank only below the bound
            st = HipRankStages(self.ctx, self.weights["encoder_h"], self.biases["encoder_b"], sh["rank_bound"], dtype)
            two = dict(local_begin=st.local_begin, local_finish=st.local_finish) if sh.get("tau_exchange") else {}
            sh["rankers"][dtype] = ShardedRanker(st.local_topk, st.merge, group=sh["group"], exchange=sh["exchange"], **two)
        return sh["rankers"][dtype]

    # -- multi-GPU training: rows of W_enc / W_dec / b_dec sharded over the ranks ---------
    def shard_training(self, rank, world, group=None):
        """Train through sharding.ShardedTrainer: this rank updates only its vocabulary rows (two
        [batch, hidden] all-reduces per step over RCCL).  Inference keeps a full replica per rank
        (W_enc replicated); it is refreshed from the shards by `sync_params`, which the
        scoring / persistence entry points call when training has run since the last refresh."""
        from ..sharding import HipTrainStages, ShardedTrainer
        self._sharded = ShardedTrainer(self.get_params(), self.n_batch, self.learning_rate,
                                       self.reg_lambda, self.tied, HipTrainStages(self.ctx),
                                       device=self.weights["encoder_h"].device, rank=rank, world=world,
                                       group=group, seed=int(self._rng.randint(0, 2 ** 31 - 1)))

    def _flush_rows_adam(self):
        """Every encoder row current for the last training step (rows without gradient lag behind until someone
        needs the whole matrix: evaluation, saving, sharding)."""
        lz = self._lazy
        if lz is None or lz["flushed"] == self._step:
            return
        self.ctx.bind_stream()
        m, v = self._adam["encoder_h"]
        P = _lib._ptr
        self.ctx.check(self.ctx.lib.dae_adam_rows_flush(
            self.ctx.h, P(self.weights["encoder_h"]), P(m), P(v), P(lz["state"]), P(lz["tab"]), lz["tab"].numel(),
            self.n_input, self.n_hidden, 0.9, 0.999, 1e-8, self._step))
        lz["flushed"] = self._step

    def sync_params(self):
        """Parameters current on this rank: pending row updates of the encoder applied, and -- collective when
        sharded (every rank must call it at the same point) -- the replica refreshed from the shards."""
        self._flush_rows_adam()
        if self._sharded is None or not self._params_stale:
            return
        enc_W, dec_W, enc_b, dec_b = self._sharded.gather_params(as_numpy=False)     # stays on the device
        self.weights["encoder_h"].copy_(enc_W)
        if not self.tied:
            self.weights["decoder_h"].copy_(dec_W)
        self.biases["encoder_b"].copy_(enc_b)
        self.biases["decoder_b"].copy_(dec_b)
        self._params_stale = False
        self._mark_dirty()

    def _ensure_packed(self, dtype=_lib.DAE_DTYPE_F32, cols=None):
        """The context's packed decoder image of `dtype` holds columns `cols` (default: all) of the current
        weights.  (A context keeps one image per dtype: a sharded `recommend` and a dense `predict` on the same
        model re-tile when they alternate -- 0.13 ms.)"""
        self.sync_params()
        shard = cols is not None and isinstance(cols[0], (tuple, list))        # sharding.scoring_shard: (tracks, artists)
        cols = (0, self.n_input) if cols is None else (tuple(map(tuple, cols)) if shard else (int(cols[0]), int(cols[1])))
        if self._packed_dirty[dtype] or self._packed_cols.get(dtype) != cols:
            self.ctx.bind_stream()
            if shard:
                from ..sharding import prepack_scoring_shard
                _bound, self._shard_rows = prepack_scoring_shard(self.ctx, self.weights["decoder_h"],
                                                                 self.biases["decoder_b"], cols, dtype)
            else:
                self.ctx.prepack_decoder(self.weights["decoder_h"], self.biases["decoder_b"], cols[0], cols[1], dtype)
            self._packed_dirty[dtype] = False
            self._packed_cols[dtype] = cols
            pg = self.__dict__.setdefault("_pack_gen", {})               # bumps whenever a slot of the context is re-tiled
            slot = "f32" if dtype == _lib.DAE_DTYPE_F32 else "bf16"
            pg[slot] = pg.get(slot, 0) + 1
            # "bf16" and "exact_bf16" share the context's bf16 image: the exact prepack serves both, a plain bf16
            # prepack drops the bounds the exact mode needs
            if dtype == _lib.DAE_DTYPE_BF16_EXACT:
                self._packed_dirty[_lib.DAE_DTYPE_BF16] = False
                self._packed_cols[_lib.DAE_DTYPE_BF16] = cols
            elif dtype == _lib.DAE_DTYPE_BF16:
                self._packed_dirty[_lib.DAE_DTYPE_BF16_EXACT] = True

    def encode(self, x_positions, x_ones, keep_prob=1.0, input_keep_prob=1.0, seed=0):
        """DAEs.py:40-42 + :64-70 -> hidden [n_batch, n_hidden] (torch CUDA tensor)."""
        import torch
        self.sync_params()
        self.ctx.bind_stream()
        rp, c, v = self._upload_csr(x_positions, x_ones)
        h = torch.empty((self.n_batch, self.n_hidden), dtype=torch.float32,
                        device=self.weights["encoder_h"].device)
        self.ctx.encode(rp, c, v, self.weights["encoder_h"], self.biases["encoder_b"], h,
                        ikp=input_keep_prob, kp=keep_prob, seed=seed)
        return h

    def predict(self, x_positions, x_ones, keep_prob=1.0, input_keep_prob=1.0, seed=0):
        """sess.run(model.y_pred, ...) : dense scores [n_batch, n_input] float32 (host array)."""
        import torch
        self._ensure_packed()
        h = self.encode(x_positions, x_ones, keep_prob, input_keep_prob, seed)
        out = torch.empty((self.n_batch, self.n_input), dtype=torch.float32, device=h.device)
        self.ctx.decode_dense(h, out, apply_sigmoid=True)
        res = out.cpu().numpy()
        self._check_feed()
        return res

    def _dtype_of(self, dtype):
        if dtype is None:
            return self.decode_dtype
        if dtype in _DECODE_DTYPES:
            return _DECODE_DTYPES[dtype]
        if dtype in _DECODE_DTYPES.values():
            return int(dtype)
        raise ValueError("decode dtype %r: one of %s" % (dtype, sorted(_DECODE_DTYPES)))

    def _seed_csr_dev(self, seeds, csr, side_stream=False, ctx=None, n_rows=None):
        """Seed lists -> device CSR.  `seeds` is a list of per-row track-id lists (main_challenge.py:31-35), or
        SEEDS_FROM_INPUT: the seeds are the playlist's own tracks -- what both reference drivers pass -- and are cut
        out of the input CSR on the device (dae_seeds_from_csr): no per-row list handling, no uploads."""
        import torch
        if isinstance(seeds, str):
            if seeds != SEEDS_FROM_INPUT:
                raise ValueError("seeds: a list of per-row id lists, or SEEDS_FROM_INPUT")
            return (ctx or self.ctx).seeds_from_csr(csr[0], csr[1], self.n_tracks)
        srp, sc = seeds_to_csr(seeds, n_rows or self.n_batch, self.n_tracks)
        if sc.size == 0:
            sc = np.zeros(1, np.int32)
        return self._to_dev(srp, torch.int32, side_stream), self._to_dev(sc, torch.int32, side_stream)

    def _submit(self, x_positions, x_ones, seeds, k, dtype, side_stream, titles=None, titles_use=None, ctx=None,
                n_rows=None):
        """Enqueue one batch of the fused scoring path on the current stream; nothing is fetched.
        -> (score, idx, done event).  `ctx`: the library context to run on (default: the model's); `n_rows`: rows of
        this launch when it is not the model's batch."""
        import torch
        ctx = ctx or self.ctx
        nb = n_rows or self.n_batch
        dev = self.weights["encoder_h"].device
        csr = self._upload_csr(x_positions, x_ones, side_stream=side_stream, ctx=ctx, n_rows=nb)
        d_srp, d_sc = self._seed_csr_dev(seeds, csr, side_stream, ctx=ctx, n_rows=nb)
        score = torch.empty((nb, k), dtype=torch.float32, device=dev)
        idx = torch.empty((nb, k), dtype=torch.int32, device=dev)
        ctx.score_topk(csr[0], csr[1], csr[2], self.weights["encoder_h"], self.biases["encoder_b"],
                       self.n_tracks, d_srp, d_sc, k, score, idx, dtype=dtype)
        ev = torch.cuda.current_stream(self.device_index).record_event()
        return score, idx, ev

    def recommend(self, x_positions, x_ones, seeds, k=500, n_rows=None, dtype=None):
        """Fused scoring path: encode -> decode -> top-k (track columns, seeds removed).
        Equivalent of main_challenge.py:80-90 / main_train.py:66-89 without the dense matrix.
        Returns (idx [n_rows,k] int32 with -1 padding, score [n_rows,k] float32)."""
        import torch
        dtype = self._dtype_of(dtype)
        sh = self._score_shard
        self._ensure_packed(dtype, None if sh is None else sh["cols"])
        self.ctx.bind_stream()
        n_rows = self.n_batch if n_rows is None else n_rows
        if sh is not None:
            # vocabulary-sharded: local top-k over this rank's columns, one exchange, merge (sharding.ShardedRanker)
            csr = self._upload_csr(x_positions, x_ones)
            d_srp, d_sc = self._seed_csr_dev(seeds, csr)
            score, idx = self._shard_ranker(dtype).rank_batch((csr[0], csr[1], csr[2], d_srp, d_sc), k)
            if dtype == _lib.DAE_DTYPE_BF16_EXACT:
                # the exact mode's BOUND GUARD on a shard: a violation on ANY rank leaves the merged lists unproven (a column the
                # bf16 filter wrongly dropped may have been that rank's contribution) -> the ranks agree on the flag (one 4-byte
                # all-reduce per batch) and re-score the batch with the fp32 kernels together
                import torch
                import torch.distributed as dist
                n_bad, col = self.ctx.exact_guard_read()
                flag = torch.tensor([1 if n_bad else 0], dtype=torch.int32)
                if dist.is_available() and dist.is_initialized() and dist.get_world_size(sh["group"]) > 1:
                    if dist.get_backend(sh["group"]) != "gloo":
                        flag = flag.to(idx.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=sh["group"])
                if int(flag.item()):
                    import warnings
                    warnings.warn("exact_bf16 (vocabulary shard): the bound guard fired on %s: this batch is re-scored with the fp32 "
                                  "kernels on every rank" % ("this rank, e.g. column %d" % col if n_bad else "another rank"))
                    self._guard_fallbacks = self.__dict__.get("_guard_fallbacks", 0) + 1
                    self._ensure_packed(_lib.DAE_DTYPE_F32, sh["cols"])
                    score, idx = self._shard_ranker(_lib.DAE_DTYPE_F32).rank_batch((csr[0], csr[1], csr[2], d_srp, d_sc), k)
            r0, r1 = self.owned_rows()
            n_own = max(0, min(r1, n_rows) - r0)
            res = idx[:n_own].cpu().numpy(), score[:n_own].cpu().numpy()
            self._check_feed()
            return res
        score, idx, _ev = self._submit(x_positions, x_ones, seeds, k, dtype, side_stream=False)
        res = idx[:n_rows].cpu().numpy(), score[:n_rows].cpu().numpy()
        self._check_feed()
        if dtype == _lib.DAE_DTYPE_BF16_EXACT and type(self)._submit is DAE._submit:
            # the exact mode's BOUND GUARD (include/dae_hip.h dae_exact_guard_read): a recomputed survivor outside the interval
            # the bf16 filter promised means the lists of this call are unproven -- the same call again on the fp32 kernels
            n_bad, col = self.ctx.exact_guard_read()
            if n_bad:
                import warnings
                warnings.warn("exact_bf16: the bound guard fired (%d survivors, e.g. column %d): this batch is re-scored with "
                              "the fp32 kernels" % (n_bad, col))
                self._guard_fallbacks = self.__dict__.get("_guard_fallbacks", 0) + 1
                return self.recommend(x_positions, x_ones, seeds, k=k, n_rows=n_rows, dtype=_lib.DAE_DTYPE_F32)
        return res

    def recommend_iter(self, feeds, k=500, dtype=None, want_scores=True):
        """`recommend` over a stream of batches with the host and the device overlapped (the loop of
        main_challenge.py:72-93 / main_train.py:62-96): `feeds` yields (x_positions, x_ones, seeds, n_rows) (DAE_title:
        + titles, titles_use); the generator yields (idx [n_rows,k], score [n_rows,k] or None) in order, one pair per feed.
        The loop itself is the library's (`dae_pipeline_*`, csrc/pipeline.hip): consecutive feeds are scored in
        ONE launch of up to 1024 rows (2048 in the bf16 modes; `_coalesce_count`: the reference's batches of 150 / 250 rows
        pad to 256 alone), three library contexts take the launches in turn, the CSR of launch n + 1 is built while launch n
        scores, the lists leave through the copy engine.  Rows are scored independently: every feed gets the bits
        `recommend` returns for it alone (tests/test_gpu_stream_loop.py)."""
        if self._score_shard is not None:
            for x_positions, x_ones, seeds, n_rows in feeds:         # the exchange is a collective: no run-ahead
                idx, score = self.recommend(x_positions, x_ones, seeds, k=k, n_rows=n_rows, dtype=dtype)
                yield idx, (score if want_scores else None)
            return
        dtype = self._dtype_of(dtype)
        # THE LOOP LIVES IN THE LIBRARY (include/dae_hip.h dae_pipeline_*): this generator copies feeds in and hands views of
        # pinned result blocks out; a library-owned thread issues the launches.  (Rounds 2 - 5 kept an interpreter loop beside it
        # -- staging rings, lane contexts, fetch streams, 270 lines of closures; round 6 retired it: the pipeline takes plain and
        # titled feeds, and what it does not take goes through `recommend` feed by feed, in order.)
        titled_native = (getattr(self, "title_model", None) is not None and getattr(self.title_model, "ctx", None) is not None
                         and type(self)._submit is DAE_title._submit)
        if self.device_csr and (type(self)._submit is DAE._submit or titled_native):
            yield from self._recommend_iter_native(feeds, k, _title_dtype(dtype, self) if titled_native else dtype, want_scores)
            return
        for f in feeds:                  # (device_csr = False: host-built CSRs -- the per-batch call)
            x_positions, x_ones, seeds, n_rows = f[:4]
            kw = {} if len(f) <= 4 else {"titles": f[4], "titles_use": f[5] if len(f) > 5 else None}
            idx, score = self.recommend(x_positions, x_ones, seeds, k=k, n_rows=n_rows, dtype=dtype, **kw)
            yield idx, (score if want_scores else None)

    def _native_pipe(self, dtype, k, want_scores):
        """The model's dae_pipeline for (dtype, k, scores wanted): created on first use, again after the weights changed."""
        tm = getattr(self, "title_model", None)
        if tm is not None and getattr(tm, "ctx", None) is None:
            tm = None
        key = (int(dtype), int(k), bool(want_scores), self.n_batch)
        # (dae_set_exact_margin on the model's contexts -- the guard's test hook -- reaches the pipeline's own images as well)
        margins = [getattr(c, "_exact_margin", 1.0) for c in ([self.ctx] + ([] if tm is None else [tm.ctx]))]
        margin = next((m_ for m_ in margins if m_ != 1.0), 1.0)
        gen = (self.__dict__.get("_weights_gen", 0), None if tm is None else tm.__dict__.get("_params_gen", 0), margin)
        cache = self.__dict__.setdefault("_pipes", {})
        ent = cache.get(key)
        if ent is not None and (ent[0] != gen or ent[1].h is None):
            ent[1].close()
            ent = None
        if ent is None:
            import torch
            # one pipeline at a time per model: each holds (2 lanes + 2) staging slots of pinned + device memory and a dozen
            # result blocks -- a caller that alternates dtypes pays a re-creation, not half a gigabyte of pinned memory
            for key_, (_g, old) in list(cache.items()):
                if getattr(old, "_users", 0) > 0:            # a recommend_iter generator of another (dtype, k, scores) key is still
                    continue                                 # running on it (two loops interleaved): it closes with that loop
                if self.__dict__.get("keep_pipelines"):      # (diagnosis: scripts/probe/row_diag.py)
                    self.__dict__.setdefault("_old_pipes", []).append(old)
                else:
                    old.close()
                cache.pop(key_, None)
            self._flush_rows_adam()
            torch.cuda.current_stream(self.device_index).synchronize()      # the weights are final before another thread reads them
            # (titled launches: the fp32 rule -- 5 feeds of 150 = 750 rows of the 96- / 128-row groups, as the Python loop ran them)
            group = self._coalesce_count(dtype if tm is None else None) * self.n_batch
            if tm is not None:
                group = min(group, 4096)
            # three lanes: measured best in every mode (batch 256, playlists/s through the loop, 2 / 3 / 4 lanes: fp32 0.98 /
            # 1.29 / 1.06 M, exact_bf16 4.1 / 4.8 / 4.2 M, bf16 5.3 / 6.3 / 5.6 M -- profiles/r04_notes.md)
            lanes = int(self.__dict__.get("n_lanes") or 3)
            pipe = _lib.Pipeline(self.weights["encoder_h"], self.biases["encoder_b"], self.weights["decoder_h"],
                                 self.biases["decoder_b"], self.n_tracks, dtype=dtype, k=k, group_rows=group,
                                 max_nnz=max(1 << 18, group * 1024), lanes=lanes, want_scores=want_scores,
                                 device_index=self.device_index, title=tm)
            if margin != 1.0 and dtype == _lib.DAE_DTYPE_BF16_EXACT:
                pipe.exact_margin(margin)
            ent = cache[key] = (gen, pipe)
        return ent[1]

    def _recommend_iter_native(self, feeds, k, dtype, want_scores):
        pipe = self._native_pipe(dtype, k, want_scores)
        key = (int(dtype), int(k), bool(want_scores), self.n_batch)

        from collections import deque
        rows_out = deque()           # rows each pending feed asked for (a feed is fed as the graph's n_batch rows, DAEs.py:34)
        nb_full = self.n_batch
        # results arrive a LAUNCH at a time: asking the pipeline after every feed whether something is ready is a foreign call
        # per feed for an answer that changes once per launch (the caller's thread is what bounds the bf16 loops: round 6)
        per_launch = max(1, pipe.group_rows // max(1, self.n_batch))
        n_fed = 0

        def out(r):
            n = rows_out.popleft()
            if n == nb_full:
                return r[0], (r[1] if want_scores else None)
            return r[0][:n], (r[1][:n] if want_scores else None)
        clean = False
        fallbacks0 = pipe.stats()["guard_fallbacks"] if dtype == _lib.DAE_DTYPE_BF16_EXACT else 0
        pipe._users = getattr(pipe, "_users", 0) + 1             # (_native_pipe never closes a pipeline a loop is running on)
        try:
            for f in feeds:
                x_positions, x_ones, seeds, n_rows = f[:4]
                n = self.n_batch if n_rows is None else int(n_rows)
                titles = use = None
                if len(f) > 5 and f[4] is not None and f[5] is not None and pipe.title_len is not None:
                    u = np.asarray(f[5], np.float32).reshape(-1)[:self.n_batch]
                    if u.size and np.any(u):                     # (titles_use all zero: the plain DAE, as DAE_title.recommend decides)
                        L_ = pipe.title_len
                        titles = np.full((self.n_batch, L_), -1, np.int32)
                        t = f[4]
                        nt = min(len(t), self.n_batch)
                        if nt:
                            if not isinstance(t, np.ndarray):
                                t = [([-1] * L_) if x is None else x for x in t[:nt]]
                            titles[:nt] = np.asarray(t, np.int64).reshape(-1, L_)[:nt]
                        use = np.zeros(self.n_batch, np.float32)
                        use[:min(u.size, nt)] = u[:nt]                                # (no title, no use)
                unfit = len(f) > 4 and pipe.title_len is None
                if not unfit:                                    # a feed larger than a launch slot: through recommend(), not a DaeError
                    nnz_f = int(np.shape(x_positions)[0]) if np.ndim(x_positions) == 2 else len(x_positions)
                    unfit = nnz_f > pipe.max_nnz
                if not (isinstance(seeds, str) and seeds == SEEDS_FROM_INPUT) or unfit or n > self.n_batch:
                    pipe.flush()                                 # a feed the pipeline does not take: in order, through recommend()
                    while pipe.pending:
                        yield out(pipe.poll(True))
                    kw = {} if len(f) <= 4 else {"titles": f[4], "titles_use": f[5] if len(f) > 5 else None}
                    idx, score = self.recommend(x_positions, x_ones, seeds, k=k, n_rows=n_rows, dtype=dtype, **kw)
                    yield idx, (score if want_scores else None)
                    continue
                while not pipe.submit(x_positions, x_ones, self.n_batch, titles, use):      # every lane full: hand the oldest lists out first
                    yield out(pipe.poll(True))
                rows_out.append(n)
                n_fed += 1
                if n_fed % per_launch == 0 or titles is not None:
                    while True:                                  # ... and whatever else is ready, without waiting
                        r = pipe.poll(False)
                        if r is None:
                            break
                        yield out(r)
            pipe.flush()
            while pipe.pending:
                yield out(pipe.poll(True))
            clean = True
        except _lib.DaeError as e:
            if "out of range" in str(e) or "outside" in str(e):
                raise ValueError(str(e))
            raise
        finally:
            pipe._users -= 1
            if dtype == _lib.DAE_DTYPE_BF16_EXACT and pipe.h is not None:
                n_fb = pipe.stats()["guard_fallbacks"] - fallbacks0
                if n_fb > 0:         # (the lists that went out are the fp32 kernels': the pipeline re-scored those launches itself)
                    import warnings
                    self._guard_fallbacks = self.__dict__.get("_guard_fallbacks", 0) + n_fb
                    warnings.warn("exact_bf16: the bound guard fired in %d launch(es) of the streamed loop: they were re-scored with "
                                  "the fp32 kernels" % n_fb)
            cache = self.__dict__.get("_pipes", {})
            if not clean:        # an error, or a consumer that stopped early: feeds may be queued -- this pipeline is not reused
                if cache.get(key, (None, None))[1] is pipe:
                    cache.pop(key, None)
                pipe.close()
            elif pipe._users == 0 and len(cache) > 1 and cache.get(key, (None, None))[1] is pipe:
                cache.pop(key, None)                             # another key's pipeline was created while this loop ran: one stays
                pipe.close()

    def _coalesce_count(self, dtype=None):
        """Feeds per launch of the streamed loop.  fp32 decode (the device is the limit): the count (<= 8, <= 1024 rows)
        that wastes the fewest padded rows of the 128-row groups the decode works in; more feeds on a tie.  150 -> 5
        (750 of 768 rows), 250 -> 4 (1000 of 1024), 256 -> 4, batches of >= 512 rows stay alone.  bf16 / exact_bf16
        (the host is the limit: ~0.3 ms of Python per launch against 0.2 ms of kernels for 1 024 rows): as many feeds
        as fit 2 048 rows, at most 8 (150 -> 8: 2.4 -> 3.3 M playlists/s).  `model.coalesce = n` overrides."""
        forced = self.__dict__.get("coalesce")
        if forced:
            return max(1, int(forced))
        nb = self.n_batch
        if dtype is not None and dtype != _lib.DAE_DTYPE_F32:
            return max(1, min(8, 2048 // nb))
        best, best_eff = 1, 0.0
        for m in range(1, 9):
            if m > 1 and m * nb > 1024:
                break
            eff = m * nb / float(-(-m * nb // 128) * 128)
            if eff >= best_eff - 1e-9:
                best, best_eff = m, max(eff, best_eff)
        return best

    def train_step(self, x_positions, x_ones, y_positions, y_ones, keep_prob, input_keep_prob, fetch_cost=True):
        """sess.run([model.optimizer, model.cost], ...) (main_train.py:204-213) -> cost (float).
        `fetch_cost=False` returns the cost as a 0-dim DEVICE tensor and does not wait for the step: the host
        builds the next batch while this one runs (main_train accumulates the tensor and fetches it once per
        epoch); the feed's range flag is then checked by the next fetching call or `check_feed()`."""
        import torch
        self.ctx.bind_stream()
        if self._sharded is not None:
            x = self._upload_csr(x_positions, x_ones)
            y = self._upload_csr(y_positions, y_ones)
            self._params_stale = True
            cost = self._sharded.train_step(x, y, keep_prob, input_keep_prob, fetch_cost=fetch_cost)
            if fetch_cost:
                self._check_feed()
            return cost
        dev = self.weights["encoder_h"].device
        if self._adam is None:
            self._grads = {}
            self._adam = {}
            names = ["encoder_h", "encoder_b", "decoder_b"] + ([] if self.tied else ["decoder_h"])
            for n in names:
                p = self.weights[n] if n in self.weights else self.biases[n]
                self._grads[n] = torch.zeros_like(p)
                self._adam[n] = (torch.zeros_like(p), torch.zeros_like(p))
            self._cost = torch.zeros(1, dtype=torch.float32, device=dev)
        xr, xc, xv = self._upload_csr(x_positions, x_ones, side_stream=True)
        yr, yc, yv = self._upload_csr(y_positions, y_ones, side_stream=True)
        seed = int(self._rng.randint(0, 2 ** 31 - 1))
        g = self._grads
        lib, ctx = self.ctx.lib, self.ctx
        P = _lib._ptr
        lz = None
        if not self.tied and self.reg_lambda == 0.0 and self.encoder_adam == "rows":
            if self._lazy is None:
                self._lazy = {"state": torch.zeros(2 * self.n_input, dtype=torch.int32, device=dev),
                              "tab": torch.zeros(max(8, int(self.rows_adam_table)), dtype=torch.float32, device=dev),
                              "flushed": self._step}
                ctx.check(lib.dae_set_enc_grad_prezeroed(ctx.h, 1))
            lz = self._lazy
            if self._step + 2 >= lz["tab"].numel():                       # alpha of every step so far
                lz["tab"] = torch.cat([lz["tab"], torch.zeros_like(lz["tab"])])
            # the rows this step's input names (the CSR's column array, length read on the device from row_ptr[B])
            # become current BEFORE the encode reads them
            rows_arg = (P(xc), ctypes.c_void_p(xr.data_ptr() + 4 * self.n_batch), int(xc.numel()))
            m_e, v_e = self._adam["encoder_h"]
            ctx.check(lib.dae_adam_rows_begin(ctx.h, P(self.weights["encoder_h"]), P(m_e), P(v_e), P(lz["state"]),
                                              P(lz["tab"]), lz["tab"].numel(), self.n_input, self.n_hidden,
                                              rows_arg[0], rows_arg[1], rows_arg[2], 0.9, 0.999, 1e-8, self._step + 1))
        # untied, reg_lambda == 0, hidden a multiple of 128: the dense Adam update of W_dec is applied inside the
        # decoder-gradient kernel (bit-identical parameters; the gradient never reaches memory)
        fuse_dec = (not self.tied and self.reg_lambda == 0.0 and self.n_hidden % 128 == 0
                    and self.decoder_adam == "fused")
        if fuse_dec:
            m_d, v_d = self._adam["decoder_h"]
            ctx.check(lib.dae_arm_decoder_adam(ctx.h, P(m_d), P(v_d), self.learning_rate, 0.9, 0.999, 1e-8,
                                               self._step + 1))
        ctx.check(lib.dae_train_forward_backward(
            ctx.h, P(xr), P(xc), P(xv), P(yr), P(yc), P(yv),
            P(self.weights["encoder_h"]), P(self.biases["encoder_b"]),
            P(self.weights["decoder_h"]), P(self.biases["decoder_b"]),
            self.n_input, self.n_hidden, self.n_batch, self.n_batch, 1 if self.tied else 0,
            float(input_keep_prob), float(keep_prob), seed, float(self.reg_lambda),
            P(g["encoder_h"]), P(g["encoder_b"]),
            P(g["decoder_h"]) if not self.tied else None, P(g["decoder_b"]), P(self._cost)))
        self._step += 1
        for n, grad in g.items():
            p = self.weights[n] if n in self.weights else self.biases[n]
            m, v = self._adam[n]
            if fuse_dec and n == "decoder_h":
                continue
            if lz is not None and n == "encoder_h":
                ctx.check(lib.dae_adam_rows_apply(ctx.h, P(p), P(m), P(v), P(grad), P(lz["state"]), P(lz["tab"]),
                                                  lz["tab"].numel(), self.n_input, self.n_hidden, rows_arg[0],
                                                  rows_arg[1], rows_arg[2], self.learning_rate, 0.9, 0.999, 1e-8,
                                                  self._step))
                if self.rows_adam_flush_every > 0 and self._step - lz["flushed"] >= self.rows_adam_flush_every:
                    self._flush_rows_adam()
                continue
            ctx.check(lib.dae_adam_step(ctx.h, P(p), P(m), P(v), P(grad), p.numel(),
                                        self.learning_rate, 0.9, 0.999, 1e-8, self._step))
        self._mark_dirty()
        if not fetch_cost:
            return self._cost[0].clone()
        cost = float(self._cost.item())
        self._check_feed()
        return cost

    # -- persistence ----------------------------------------------------------------------------------
    def get_params(self):
        """sess.run(model.d_params): [enc_W, dec_W (enc_W again when tied), enc_b, dec_b]."""
        self.sync_params()
        return [p.detach().cpu().numpy() for p in self.d_params]

    def save_model(self, sess=None):
        """DAEs.py:107-111: pickle of the four float32 arrays in d_params order."""
        with open(self.save_dir, "wb") as f:
            pickle.dump(self.get_params(), f)


class DAE(DAE_tied):
    """Untied DAE, optionally initialised from a pretrain pickle (reference DAEs.py:114-150)."""

    tied = False

    def __init__(self, conf):
        DAE_tied.__init__(self, conf)
        self.initval_dir = conf.initval               # DAEs.py:117

    def _host_init(self):
        if str(self.initval_dir).endswith("NULL"):    # DAEs.py:120 (Conf joins the dir in front)
            rng = np.random.default_rng(self.init_seed)
            return [self._xavier(rng, (self.n_input, self.n_hidden)),
                    self._xavier(rng, (self.n_input, self.n_hidden)),
                    np.zeros(self.n_hidden, np.float32), np.zeros(self.n_input, np.float32)]
        with open(self.initval_dir, "rb") as f:       # DAEs.py:130-135
            emb = pickle.load(f)
        return [np.array(emb[0], np.float32), np.array(emb[1], np.float32),
                np.array(emb[2], np.float32), np.array(emb[3], np.float32)]


class Session:
    """Minimal stand-in for tf.Session so that reference-style driver code runs unchanged:
    `sess.run(model.y_pred, feed_dict=...)`, `sess.run([model.optimizer, model.cost], ...)`,
    `sess.run(model.init_op)`, `sess.run(model.d_params)`."""

    def __init__(self, model=None):
        self.model = model

    def run(self, fetches, feed_dict=None):
        m = self.model
        if m is None:
            raise ValueError("Session needs the model: Session(model)")
        feed = feed_dict or {}

        def g(ph, default=None):
            return feed.get(ph, default)

        if fetches is m.init_op:
            return None
        if fetches is m.d_params or (isinstance(fetches, list) and fetches and fetches == m.d_params):
            return m.get_params()
        if fetches is m.y_pred:
            return m.predict(g(m.x_positions), g(m.x_ones), g(m.keep_prob, 1.0),
                             g(m.input_keep_prob, 1.0))
        if isinstance(fetches, (list, tuple)) and len(fetches) == 2 and fetches[0] is m.optimizer \
                and fetches[1] is m.cost:
            cost = m.train_step(g(m.x_positions), g(m.x_ones), g(m.y_positions), g(m.y_ones),
                                g(m.keep_prob, 1.0), g(m.input_keep_prob, 1.0))
            return None, cost
        raise ValueError("unsupported fetches %r" % (fetches,))


class DAE_title(DAE):
    """The challenge-time model (reference DAEs.py:153-201): a FROZEN DAE (its four arrays from `conf.DAEval`,
    :165-171) whose scores are mixed with the title scorer's,

        y = title_score * w_title + dae_score * w_playlist                                   (:180)
        x_count = row_sum(x) * input_keep_prob;  w_title = u / (u + x_count + 1e-10);  w_playlist = x_count / (same)

    with u = titles_use in {0, 1} per row (:156-162).  u = 0 gives w_playlist = 1.0f exactly, i.e. the plain
    DAE -- those batches take the fused scoring path.  With titles the two vocabulary-wide
    score matrices are materialised and mixed (`dae_mix_scores`), then ranked by `dae_topk_dense`: correct but
    unfused (fusing the second GEMM into the threshold path is the follow-up)."""

    def __init__(self, conf, title_model):
        DAE.__init__(self, conf)
        self.initval_dir = conf.DAEval                    # DAEs.py:157: the frozen weights
        self.title_model = title_model

    def _row_sums(self, x_positions, x_ones):
        rp, _c, v = coo_to_csr(x_positions, x_ones, self.n_batch, self.n_input)
        rows = np.repeat(np.arange(self.n_batch), np.diff(rp))
        return np.bincount(rows, weights=v.astype(np.float64), minlength=self.n_batch).astype(np.float32)

    def _mix_weights(self, csr, titles_use, input_keep_prob=1.0, seed=0, side_stream=False, n_rows=None, u_dev=None):
        """DAEs.py:159-162 on the device: x_count = row_sum(x) * input_keep_prob; w_title = u / (u + x_count + 1e-10),
        w_playlist = x_count / (same) -- fp32 operations in the reference's order.  -> (w_title, w_playlist) [n_batch]."""
        import torch
        rp, c, v = csr
        dev = rp.device
        nb = n_rows or self.n_batch
        P = _lib._ptr
        if u_dev is None:
            u = np.zeros(nb, np.float32)
            tu = np.asarray(titles_use, np.float32).reshape(-1)
            u[:len(tu)] = tu[:nb]
            u = self._to_dev(u, torch.float32, side_stream)
        else:
            u = u_dev                                         # (already on the device)
        w_t = torch.empty(nb, dtype=torch.float32, device=dev)
        w_p = torch.empty(nb, dtype=torch.float32, device=dev)
        # one launch (dae_mix_weights: the row sums of dae_row_sums and the four elementwise operations behind them)
        self.ctx.check(self.ctx.lib.dae_mix_weights(self.ctx.h, P(rp), P(c), P(v), nb, float(input_keep_prob), int(seed),
                                                    P(u), P(w_t), P(w_p)))
        return w_t, w_p

    def mixed_scores(self, x_positions, x_ones, titles, titles_use, input_keep_prob=1.0, title_keep_prob=1.0,
                     seed=0):
        """sess.run(model.y_pred, ...) of the title graph: dense mixed scores [n_batch, n_input] (CUDA tensor).  The
        dense fetch of the protocol; `recommend` never builds these matrices."""
        import torch
        self._ensure_packed()
        self.ctx.bind_stream()
        csr = self._upload_csr(x_positions, x_ones)
        h = torch.empty((self.n_batch, self.n_hidden), dtype=torch.float32, device=self.weights["encoder_h"].device)
        self.ctx.encode(csr[0], csr[1], csr[2], self.weights["encoder_h"], self.biases["encoder_b"], h,
                        ikp=input_keep_prob, kp=1.0, seed=seed)
        y = torch.empty((self.n_batch, self.n_input), dtype=torch.float32, device=h.device)
        self.ctx.decode_dense(h, y, apply_sigmoid=True)
        ts = self.title_model.score(titles, self.n_batch, title_keep_prob, seed)
        w_t, w_p = self._mix_weights(csr, titles_use, input_keep_prob, seed)
        self.ctx.bind_stream()
        P = _lib._ptr
        self.ctx.check(self.ctx.lib.dae_mix_scores(self.ctx.h, P(ts), int(ts.stride(0)), P(y), int(y.stride(0)),
                                                   P(w_t), P(w_p), self.n_batch, self.n_input))
        return y

    def train_step(self, x_positions, x_ones, y_positions, y_ones, keep_prob, input_keep_prob, titles=None,
                   titles_use=None, title_keep_prob=1.0):
        """sess.run([model.optimizer, model.cost], ...) of the title graph (main_train.py:214-221): one Adam step
        on the title variables; the DAE arrays stay as loaded.  Returns the cost."""
        import torch
        if titles is None:
            raise ValueError("DAE_title trains the title scorer: titles are required")
        tm = self.title_model
        seed = int(self._rng.randint(0, 2 ** 31 - 1))
        self._ensure_packed()
        self.ctx.bind_stream()
        tm.ctx.bind_stream()
        dev = self.weights["encoder_h"].device
        # frozen DAE forward with its dropouts (model.keep_prob = conf.kp, input_keep_prob; main_train.py:219-220)
        xr, xc, xv = self._upload_csr(x_positions, x_ones)
        yr, yc, yv = self._upload_csr(y_positions, y_ones)
        h = torch.empty((self.n_batch, self.n_hidden), dtype=torch.float32, device=dev)
        self.ctx.encode(xr, xc, xv, self.weights["encoder_h"], self.biases["encoder_b"], h,
                        ikp=input_keep_prob, kp=keep_prob, seed=seed)
        dae = torch.empty((self.n_batch, self.n_input), dtype=torch.float32, device=dev)
        self.ctx.decode_dense(h, dae, apply_sigmoid=True)
        # mixing weights (DAEs.py:159-162) from the dropped-out row sums
        s = torch.empty(self.n_batch, dtype=torch.float32, device=dev)
        P = _lib._ptr
        self.ctx.check(self.ctx.lib.dae_row_sums(self.ctx.h, P(xr), P(xc), P(xv), self.n_batch,
                                                 float(input_keep_prob), seed, P(s)))
        u = torch.zeros(self.n_batch, dtype=torch.float32, device=dev)
        tu = np.ones(self.n_batch, np.float32) if titles_use is None else np.asarray(titles_use, np.float32).reshape(-1)
        u[:min(len(tu), self.n_batch)] = torch.from_numpy(tu[:self.n_batch]).to(dev)
        x_count = s * float(np.float32(input_keep_prob))
        deno = u + x_count + 1e-10
        w_t, w_p = (u / deno).contiguous(), (x_count / deno).contiguous()
        # title forward, keeping what the backward pass needs
        tm._ensure_packed(features_table=False)
        feat, d_titles, arg, raw = tm.features(titles, self.n_batch, title_keep_prob, seed, keep_for_backward=True)
        zt = torch.empty((self.n_batch, self.n_input), dtype=torch.float32, device=dev)
        tm.ctx.decode_dense(feat, zt, apply_sigmoid=False)
        if getattr(self, "_tcost", None) is None:
            self._tcost = torch.zeros(1, dtype=torch.float32, device=dev)
        tm.backward_and_step(feat, d_titles, arg, raw, zt, dae, (yr, yc, yv), w_t, w_p, self.n_batch,
                             title_keep_prob, seed, self._tcost)
        cost = float(self._tcost.item())
        self._check_feed()
        return cost

    def _submit(self, x_positions, x_ones, seeds, k, dtype, side_stream, titles=None, titles_use=None, ctx=None,
                n_rows=None):
        """One batch enqueued, nothing fetched (`n_rows`: rows of a launch that coalesces several feeds).  Without titles in use the mix reduces to the plain DAE (w_playlist is
        exactly 1.0f, App. B.6): the fused path of the base class.  With titles: the DAE term of the track columns
        goes to a transposed scratch (dae_decode_mix_term) and the title scorer's context runs the fused threshold
        path on sigmoid(z_title) * w_title + that term (dae_set_score_mix) -- no [batch, n_input] matrix of either."""
        if titles is None or titles_use is None or not np.any(np.asarray(titles_use)):
            return DAE._submit(self, x_positions, x_ones, seeds, k, dtype, side_stream, ctx=ctx, n_rows=n_rows)
        import torch
        tm = self.title_model
        tm.ctx.bind_stream()
        dtype = _title_dtype(dtype, self)
        if dtype == _lib.DAE_DTYPE_BF16_EXACT and self.__dict__.get("_mix_exact_pause", 0) > 0:
            # rows of this model keep overflowing the refine launch's candidate buffers (a flat DAE bias: no prior for the
            # threshold sample): the exact launches are paused for a while instead of being run and re-scored every time
            self._mix_exact_pause -= 1
            self._ensure_packed(_lib.DAE_DTYPE_F32)
            tm._ensure_packed(_lib.DAE_DTYPE_F32)
            dtype = _lib.DAE_DTYPE_F32
        nb = n_rows or self.n_batch
        dev = self.weights["encoder_h"].device
        d_titles = d_use = None
        # (Measured and not kept: the launch's two independent preambles -- CSR build / encode / seeds / mixing weights, and the
        # title features -- on streams of their own, joined before dae_mix_topk_exact: 1.12 - 1.17 M playlists/s against
        # 1.22 M without.  The loop is bound by the host's ~0.6 ms per launch, and the fork adds events and stream switches.)
        feat = None
        csr = self._upload_csr(x_positions, x_ones, side_stream=side_stream, n_rows=nb)
        h = torch.empty((nb, self.n_hidden), dtype=torch.float32, device=dev)
        self.ctx.encode(csr[0], csr[1], csr[2], self.weights["encoder_h"], self.biases["encoder_b"], h)
        w_t, w_p = self._mix_weights(csr, titles_use, side_stream=side_stream, n_rows=nb, u_dev=d_use)
        if dtype == _lib.DAE_DTYPE_BF16_EXACT:
            # both GEMMs on bf16 operands in one launch per pass, the survivors recomputed in fp32 (csrc/mixexact.hip)
            if feat is None:
                feat = tm.features(titles, nb, side_stream_of=self if side_stream else None, d_titles=d_titles)
            d_srp, d_sc = self._seed_csr_dev(seeds, csr, side_stream, n_rows=nb)
            score = torch.empty((nb, k), dtype=torch.float32, device=dev)
            idx = torch.empty((nb, k), dtype=torch.int32, device=dev)
            gw = torch.empty(2, dtype=torch.int32, device=dev)
            tm.ctx.mix_topk_exact(self.ctx, feat, h, w_t, w_p, self.n_tracks, d_srp, d_sc, k, score, idx, guard_out=gw)
            ev = torch.cuda.current_stream(self.device_index).record_event()
            # the guard words travel with the lists; what a re-scoring in fp32 needs, should they have moved
            idx._mix_guard = (gw, (x_positions, x_ones, seeds, titles, titles_use, n_rows))
            return score, idx, ev
        nt32 = min((self.n_tracks + 31) // 32 * 32, self.n_input)
        y1T = torch.empty((nt32, nb), dtype=torch.float32, device=dev)
        self.ctx.decode_mix_term(h, w_p, self.n_tracks, y1T, dtype=dtype)
        feat = tm.features(titles, nb, side_stream_of=self if side_stream else None, d_titles=d_titles)
        d_srp, d_sc = self._seed_csr_dev(seeds, csr, side_stream, n_rows=nb)
        score = torch.empty((nb, k), dtype=torch.float32, device=dev)
        idx = torch.empty((nb, k), dtype=torch.int32, device=dev)
        tm.ctx.set_score_mix(y1T, w_t)
        try:
            tm.ctx.decode_topk(feat, self.n_tracks, d_srp, d_sc, k, score, idx, out_kind=_lib.DAE_OUT_LOGIT, dtype=dtype)
        finally:
            tm.ctx.set_score_mix()
        ev = torch.cuda.current_stream(self.device_index).record_event()
        return score, idx, ev

    def _mix_guard_fired(self, idx, k, words=None, score=None):
        """A launch of the exact title mix whose guard words moved is not trusted (a recomputed logit left the interval
        the bf16 launch promised for it, or a row overflowed its candidate list): -> (score, idx) of the same feed through
        the fp32 kernels, or None when the launch stands."""
        tag = getattr(idx, "_mix_guard", None)
        if tag is None:
            return None
        gw, (x_positions, x_ones, seeds, titles, titles_use, n_rows) = tag
        n_bad, col = (int(v) for v in (gw.cpu() if words is None else words))
        # (the shadow of the cumulative words lives on the context and is resynchronised by every dae_exact_guard_read on it:
        # ADVICE r4 -- a read elsewhere used to leave this comparison one reset behind)
        seen = getattr(self.title_model.ctx, "_guard_seen", 0)
        if not self.title_model.ctx.guard_moved(n_bad):
            self._mix_overflow_streak = 0
            return None
        import warnings
        import torch
        self._ensure_packed(_lib.DAE_DTYPE_F32)
        self.title_model._ensure_packed(_lib.DAE_DTYPE_F32)
        n_new = n_bad - seen
        nb = n_rows or self.n_batch
        if col == -3:
            # rows whose survivors overflow the refine launch's list return no recommendations (idx -1): when every event of
            # this launch is such a row (e.g. a title-only playlist under a flat title scorer: its scores cannot be told apart
            # by any bound), ONLY those rows are re-scored with the fp32 kernels and patched in
            rows = np.nonzero(idx[:nb, 0].cpu().numpy() < 0)[0]
            use = np.asarray(titles_use, np.float32).reshape(-1)
            if 0 < len(rows) == n_new and len(rows) <= nb // 2 and np.all(use[rows] > 0):
                P = np.asarray(x_positions, np.int64).reshape(-1, 2)
                sel = np.isin(P[:, 0], rows)
                newid = np.full(nb, -1, np.int64)
                newid[rows] = np.arange(len(rows))
                P2 = np.stack([newid[P[sel, 0]], P[sel, 1]], 1) if sel.any() else np.zeros((0, 2), np.int64)
                o = np.asarray(x_ones, np.float32).reshape(-1)
                O2 = o[sel] if o.size == len(P) else o
                seeds2 = seeds if isinstance(seeds, str) else [list(seeds[r]) if r < len(seeds) else [] for r in rows]
                T2 = np.asarray(titles, np.int64).reshape(-1, self.title_model.input_len)[rows]
                warnings.warn("exact_bf16 title mix: the bound guard fired (%d row(s) overflow the refine launch): those rows are "
                              "re-scored with the fp32 kernels" % len(rows))
                self._guard_row_fallbacks = self.__dict__.get("_guard_row_fallbacks", 0) + len(rows)
                score2, idx2, _ev = self._submit(P2, O2, seeds2, k, _lib.DAE_DTYPE_F32, False, T2, use[rows], n_rows=len(rows))
                rt = torch.from_numpy(rows).to(idx.device)
                idx_n, score_n = idx.clone(), score.clone()
                idx_n[rt] = idx2[:len(rows)]
                score_n[rt] = score2[:len(rows)]
                self._mix_overflow_streak = 0
                return score_n, idx_n
        if col in (-2, -3):                     # not a broken bound, but many rows the refine launch cannot hold: after two such
            self._mix_overflow_streak = self.__dict__.get("_mix_overflow_streak", 0) + 1      # launches in a row the mode pauses
            if self._mix_overflow_streak >= 2:
                self._mix_exact_pause = 64      # launches on the fp32 kernels before it is tried again
                self._mix_overflow_streak = 0
        warnings.warn("exact_bf16 title mix: the bound guard fired (%s): this launch is re-scored with the fp32 kernels"
                      % ("rows overflow the refine launch" if col in (-2, -3) else "column %d" % col))
        self._guard_fallbacks = self.__dict__.get("_guard_fallbacks", 0) + 1
        score, idx2, _ev = self._submit(x_positions, x_ones, seeds, k, _lib.DAE_DTYPE_F32, False, titles, titles_use,
                                        n_rows=n_rows)
        return score, idx2

    def recommend(self, x_positions, x_ones, seeds, k=500, n_rows=None, dtype=None, titles=None, titles_use=None):
        """Top-k of the MIXED score (DAEs.py:176-181 + main_challenge.py:26-41) without either [batch, n_input] matrix
        (see `_submit`).  Same operations in the same order as `mixed_scores` + dae_topk_dense, hence the same bits
        (fp32).  dtype "bf16" (or [BASE] decode_dtype = bf16) runs BOTH vocabulary-wide GEMMs -- the DAE decoder and
        the title scorer's output layer -- on bf16 operands with fp32 accumulate; encode, title features, sigmoids,
        the mix, the threshold and the ranking stay fp32."""
        if titles is None or titles_use is None or not np.any(np.asarray(titles_use)):
            return DAE.recommend(self, x_positions, x_ones, seeds, k=k, n_rows=n_rows, dtype=dtype)
        if self._score_shard is not None:
            raise _lib.DaeError("title-mixed batches are not vocabulary-sharded: run --challenge with titles on one "
                                "GPU per process group (playlist partitioning), or without the title variables")
        dtype = _title_dtype(self._dtype_of(dtype), self)
        self._ensure_packed(dtype)
        self.title_model._ensure_packed(dtype)
        self.ctx.bind_stream()
        score, idx, _ev = self._submit(x_positions, x_ones, seeds, k, dtype, False, titles, titles_use)
        n_rows = self.n_batch if n_rows is None else n_rows
        redo = self._mix_guard_fired(idx, k, None, score)
        if redo is not None:
            score, idx = redo
        res = idx[:n_rows].cpu().numpy(), score[:n_rows].cpu().numpy()
        self._check_feed()
        return res
