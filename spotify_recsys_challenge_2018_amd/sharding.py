"""Vocabulary (output-column) sharding of the scoring path across the GPUs of one node
(BASELINE.json configs[2]).  One process per GPU, torch.distributed over RCCL.

Rank g owns the decoder rows (vocabulary columns) [lo_g, hi_g) -- tile aligned so every shard's
packed image starts on a 32-column MFMA tile -- and W_enc is replicated, so every rank computes
the same hidden activations with no collective.  The only exchange step of the path is that of the
per-shard top-k lists ((logit, column) x k per playlist): an all-gather after which every rank merges
the G lists of every row, or an all-to-all after which every rank merges the G lists of the rows it
owns -- same (logit desc, column asc) rule either way.  The merge is exact because each shard's
top-k contains that shard's share of the global top-k.
"""
import numpy as np

TILE = 32


def shard_bounds(n_cols, world, rank, align=TILE):
    """Columns [lo, hi) of shard `rank`: contiguous, tile-aligned starts, sizes differ by at most
    one tile."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank %r/%r" % (world, rank))
    n_tiles = (n_cols + align - 1) // align
    base, extra = divmod(n_tiles, world)
    lo_t = rank * base + min(rank, extra)
    hi_t = lo_t + base + (1 if rank < extra else 0)
    return min(lo_t * align, n_cols), min(hi_t * align, n_cols)


def all_shard_bounds(n_cols, world, align=TILE):
    return [shard_bounds(n_cols, world, g, align) for g in range(world)]


def scoring_shard(n_tracks, n_input, world, rank):
    """The columns rank `rank` of a vocabulary-sharded SCORING job decodes: an equal, tile-aligned slice of the TRACK
    columns (the only ones that are ranked, main_challenge.py:87) and an equal slice of the artist columns (decoded
    because the reference decodes every column, DAEs.py:143; never ranked) -> ((t_lo, t_hi), (a_lo, a_hi)).
    Splitting [0, n_input) as one range (round 2) left the last ranks with artist columns only: same GEMM work, but
    empty candidate lists there and all of the selection work on the first ranks."""
    t_lo, t_hi = shard_bounds(n_tracks, world, rank)
    a_lo, a_hi = shard_bounds(n_input - n_tracks, world, rank)
    return (t_lo, t_hi), (n_tracks + a_lo, n_tracks + a_hi)


def prepack_scoring_shard(ctx, W_dec, b_dec, shard, dtype=0):
    """Prepack `ctx`'s decoder image with the rows of `shard` (= scoring_shard(...)): the rank keeps ONE contiguous
    copy [track rows | artist rows] of its part of the decoder (a shard owner holds 1/G of the matrix).  The image's
    global columns are [t_lo, t_lo + n): the track part under its real ids, the artist part under ids that are never
    emitted because the rank bound the scoring calls pass is t_hi (the returned value: use it as `n_tracks` in
    dae_score_topk / HipRankStages).  Seed lists and results stay in GLOBAL track ids.  Returns (rank_bound, keepalive)."""
    import torch
    (t_lo, t_hi), (a_lo, a_hi) = shard
    W_loc = torch.cat([W_dec[t_lo:t_hi], W_dec[a_lo:a_hi]]).contiguous()
    b_loc = torch.cat([b_dec[t_lo:t_hi], b_dec[a_lo:a_hi]]).contiguous()
    if W_loc.shape[0] == 0:
        raise ValueError("empty scoring shard %r" % (shard,))
    ctx.prepack_decoder_rows(W_loc, b_loc, t_lo, dtype)
    return t_hi, (W_loc, b_loc)


def gather_shard_topk(local_logit, local_idx, group=None, out=None):
    """All-gather the per-shard candidate lists: [B,k] -> [G,B,k] (logit fp32, column int32).
    One collective per tensor; with RCCL over xGMI each rank sends its 2 x B x k x 4 bytes once."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    B, k = local_logit.shape
    if out is None:
        out = (torch.empty((world * B, k), dtype=local_logit.dtype, device=local_logit.device),
               torch.empty((world * B, k), dtype=local_idx.dtype, device=local_idx.device))
    g_logit, g_idx = out                       # rank-major concatenation along dim 0
    _collective(dist.all_gather_into_tensor, g_logit, local_logit.contiguous(), group)
    _collective(dist.all_gather_into_tensor, g_idx, local_idx.contiguous(), group)
    return g_logit.view(world, B, k), g_idx.view(world, B, k)


def _collective(fn, dst, src, group):
    """fn(dst, src, group=group).  RCCL takes device tensors as they are; gloo (CPU tests, and the one-GPU
    rehearsal of the N > 1 flow) has no device all-gather / all-to-all, so device tensors go through the host."""
    import torch.distributed as dist
    if src.is_cuda and dist.get_backend(group) == "gloo":
        d = dst.cpu()
        fn(d, src.cpu(), group=group)
        dst.copy_(d)
    else:
        fn(dst, src, group=group)


def row_owner_bounds(n_rows, world, rank):
    """Playlists [lo, hi) whose FINAL top-k rank `rank` produces in the "alltoall" exchange (equal blocks)."""
    if n_rows % world:
        raise ValueError("the row-owner exchange needs n_rows (%d) divisible by the world size (%d)" % (n_rows, world))
    per = n_rows // world
    return rank * per, (rank + 1) * per


def exchange_shard_topk(local_logit, local_idx, group=None, out=None):
    """All-to-all of the per-shard candidate lists: rank r keeps, from every shard, only the rows it OWNS
    (`row_owner_bounds`): [B,k] -> [G, B/G, k].  Same lists as the all-gather restricted to the owned rows, so
    the merge is the same and exact; each rank sends (G-1)/G of its 2 x B x k x 4 bytes ONCE in total instead of
    once per peer, and merges B/G rows instead of B.  xGMI is point-to-point: every byte crosses one link."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    B, k = local_logit.shape
    row_owner_bounds(B, world, 0)
    if out is None:
        out = (torch.empty_like(local_logit), torch.empty_like(local_idx))
    x_logit, x_idx = out                       # source-rank-major along dim 0
    _collective(dist.all_to_all_single, x_logit, local_logit.contiguous(), group)
    _collective(dist.all_to_all_single, x_idx, local_idx.contiguous(), group)
    return x_logit.view(world, B // world, k), x_idx.view(world, B // world, k)


class ShardedRanker:
    """decode + top-k over this rank's vocabulary shard, exchange, merge -- the N > 1 form of
    main_challenge.py:80-90 (one batch) that `DAE.shard_scoring` and bench.py both run.

    `local_topk(feed, k) -> (logit [B,k], idx [B,k])` and `merge(g_logit, g_idx) -> (score, idx)` are
    the two device operations (HipRankStages below: dae_score_topk / dae_decode_topk with DAE_OUT_LOGIT
    over the rank's prepacked columns, and dae_topk_merge); they are injected so the exchange logic is
    testable with gloo on CPU.  `feed` is whatever local_topk takes (the batch's device CSR + seeds, or
    hidden activations).

    exchange = "allgather": every rank ends with the merged top-k of ALL B rows (BASELINE.json configs[2] as
    written).  exchange = "alltoall": every rank ends with the merged top-k of the rows it owns
    (`row_owner_bounds`) -- 1/G of the exchange bytes and of the merge work; the job's output is the
    concatenation over ranks.

    bufs: optional preallocated (logit, idx) receive buffers of the exchange ([G*B,k] for the all-gather, [B,k]
    for the all-to-all), reused by every call.  gather: optional replacement of the collective,
    `gather(l_logit, l_idx) -> (g_logit [G,rows,k], g_idx)` (single-process tests that hold every shard).

    THRESHOLD EXCHANGE (optional: `local_begin` / `local_finish` given): a shard alone can only bound the row's k-th
    largest logit by ITS OWN k-th largest, so it returns k candidates per row although only ~k/G of them can be in the
    merged list.  `local_begin(feed, k) -> tau [B]` runs the shard's encode + threshold sample (dae_score_topk_begin),
    the shards' bounds meet in one all-gather of 4 bytes per row and rank (8 KB per rank at 2048 rows), their
    element-wise maximum -- still a lower bound of the global k-th largest -- goes to `local_finish(feed, k, tau) ->
    (logit, idx)` (dae_score_topk_finish), and the lists come back holding only what can still matter (padded with
    -inf / -1): the same merged result, less selection work per shard.  `gather_tau(tau) -> [G, B]` replaces that
    collective in single-process tests."""

    def __init__(self, local_topk, merge, group=None, exchange="allgather", bufs=None, gather=None,
                 local_begin=None, local_finish=None, gather_tau=None):
        if exchange not in ("allgather", "alltoall"):
            raise ValueError("unknown exchange %r" % (exchange,))
        if (local_begin is None) != (local_finish is None):
            raise ValueError("local_begin and local_finish go together")
        self.local_topk = local_topk
        self.merge = merge
        self.group = group
        self.exchange = exchange
        self.bufs = bufs
        self.gather = gather
        self.local_begin, self.local_finish, self.gather_tau = local_begin, local_finish, gather_tau
        self._tau_buf = None
        self._marks = None            # profile_phases(True): per call, the stream events between the stages

    # -- where a step's time goes (bench.py `phases`): events on the caller's stream between the stages -------------------
    def profile_phases(self, on=True):
        """Record a stream event after every stage of the following rank_batch calls (the collectives are ordered on the
        caller's stream, so the pairs bracket them).  read_phases() returns the mean milliseconds per stage."""
        self._marks = [] if on else None

    def _mark(self, marks):
        if marks is not None:
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)

    def read_phases(self):
        """{local_ms (encode + threshold sample [+ filter + selection without the threshold exchange]), tau_exchange_ms,
        filter_select_ms, exchange_ms, merge_ms, calls}: means over the calls since profile_phases(True); synchronises."""
        import torch
        calls = self._marks or []
        self._marks = [] if self._marks is not None else None
        if not calls:
            return None
        torch.cuda.synchronize()
        names = ["local_ms", "tau_exchange_ms", "filter_select_ms", "exchange_ms", "merge_ms"]
        tot = dict.fromkeys(names, 0.0)
        for ev in calls:
            for i, n in enumerate(names):
                tot[n] += ev[i].elapsed_time(ev[i + 1])
        out = {n: round(v / len(calls), 4) for n, v in tot.items()}
        out["calls"] = len(calls)
        return out

    def _exchange_tau(self, tau):
        """[B] per-shard lower bounds -> their maximum over the shards (device tensor, [B])."""
        import torch
        if self.gather_tau is not None:
            return torch.amax(self.gather_tau(tau), dim=0)
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        if self._tau_buf is None or self._tau_buf.numel() != world * tau.numel() or self._tau_buf.device != tau.device:
            self._tau_buf = torch.empty(world * tau.numel(), dtype=tau.dtype, device=tau.device)
        _collective(dist.all_gather_into_tensor, self._tau_buf, tau.contiguous(), self.group)
        return torch.amax(self._tau_buf.view(world, -1), dim=0)

    def rank_batch(self, feed, k):
        m = [] if self._marks is not None else None
        self._mark(m)
        if self.local_begin is not None:
            tau = self.local_begin(feed, k)
            self._mark(m)
            tau = self._exchange_tau(tau)
            self._mark(m)
            l_logit, l_idx = self.local_finish(feed, k, tau)
        else:
            l_logit, l_idx = self.local_topk(feed, k)
            self._mark(m); self._mark(m)
        self._mark(m)
        if self.gather is not None:
            g_logit, g_idx = self.gather(l_logit, l_idx)
        elif self.exchange == "alltoall":
            g_logit, g_idx = exchange_shard_topk(l_logit, l_idx, self.group, out=self.bufs)
        else:
            g_logit, g_idx = gather_shard_topk(l_logit, l_idx, self.group, out=self.bufs)
        self._mark(m)
        res = self.merge(g_logit, g_idx)
        self._mark(m)
        if m is not None:
            self._marks.append(m)
        return res


class HipRankStages:
    """The two device stages of vocabulary-sharded scoring through the C ABI (include/dae_hip.h), on the context
    whose prepacked decoder image holds this rank's columns [lo, hi):

      local_topk   dae_score_topk(DAE_OUT_LOGIT): encode (W_enc is replicated: every rank computes the same hidden
                   activations, no collective) -> decode + top-k over the shard; GLOBAL column ids come back
      merge        dae_topk_merge: G lists per row -> the global top-k, same (logit desc, column asc) key

    feed = (row_ptr, col, val, seed_row_ptr, seed_col) device tensors of the WHOLE batch.  Output tensors are
    allocated once per (rows, k) and reused."""

    def __init__(self, ctx, W_enc, b_enc, n_tracks, dtype=0, out_kind=0):
        self.ctx, self.W_enc, self.b_enc = ctx, W_enc, b_enc
        self.n_tracks, self.dtype, self.out_kind = int(n_tracks), int(dtype), int(out_kind)
        self._loc, self._out = {}, {}
        self._tau = None

    def _pair(self, cache, rows, k):
        import torch
        key = (int(rows), int(k))
        if key not in cache:
            dev = self.W_enc.device
            cache[key] = (torch.empty(key, dtype=torch.float32, device=dev),
                          torch.empty(key, dtype=torch.int32, device=dev))
        return cache[key]

    def local_topk(self, feed, k):
        from ._lib import DAE_OUT_LOGIT
        rp, col, val, srp, sc = feed
        logit, idx = self._pair(self._loc, rp.numel() - 1, k)
        self.ctx.score_topk(rp, col, val, self.W_enc, self.b_enc, self.n_tracks, srp, sc, k, logit, idx,
                            out_kind=DAE_OUT_LOGIT, dtype=self.dtype)
        return logit, idx

    def local_begin(self, feed, k):
        """dae_score_topk_begin -> this shard's per-row lower bounds of the k-th largest logit, [B] (reused buffer)."""
        import torch
        rp, col, val, srp, _sc = feed
        rows = rp.numel() - 1
        if self._tau is None or self._tau.numel() != rows:
            self._tau = torch.empty(rows, dtype=torch.float32, device=self.W_enc.device)
        self.ctx.score_topk_begin(rp, col, val, self.W_enc, self.b_enc, self.n_tracks, srp, k, self._tau, dtype=self.dtype)
        return self._tau

    def local_finish(self, feed, k, tau):
        """dae_score_topk_finish with the exchanged threshold: the shard's columns that can still be in the merged list."""
        from ._lib import DAE_OUT_LOGIT
        rp, _col, _val, srp, sc = feed
        logit, idx = self._pair(self._loc, rp.numel() - 1, k)
        self.ctx.score_topk_finish(tau, srp, sc, logit, idx, out_kind=DAE_OUT_LOGIT)
        return logit, idx

    def merge(self, g_logit, g_idx):
        score, idx = self._pair(self._out, g_logit.shape[1], g_logit.shape[2])
        self.ctx.topk_merge(g_logit, g_idx, score, idx, out_kind=self.out_kind)
        return score, idx


# ---- training: vocabulary rows sharded, two all-reduces per step ------------------

class HipTrainStages:
    """The three device stages of a sharded training step + local Adam, through the C ABI
    (include/dae_hip.h: dae_train_shard_encode / _decode / _finish, dae_adam_step)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def encode(self, x, W_enc, lo, hi, ikp, seed, pre):
        from ._lib import _ptr as P
        c = self.ctx
        B, H = pre.shape
        c.check(c.lib.dae_train_shard_encode(c.h, P(x[0]), P(x[1]), P(x[2]), P(W_enc), lo, hi, H, B,
                                             float(ikp), int(seed), P(pre)))

    def decode(self, pre, b_enc, y, W_enc, W_dec, b_dec, lo, hi, n_batch, tied, kp, seed, lam,
               gW_out, gb_dec, dh, cost):
        from ._lib import _ptr as P
        c = self.ctx
        B, H = pre.shape
        c.check(c.lib.dae_train_shard_decode(
            c.h, P(pre), P(b_enc), P(y[0]), P(y[1]), P(y[2]), P(W_enc), P(W_dec), P(b_dec),
            lo, hi, H, B, int(n_batch), 1 if tied else 0, float(kp), int(seed), float(lam),
            P(gW_out), P(gb_dec), P(dh), P(cost)))

    def finish(self, dh, x, W_enc, b_enc, W_dec, b_dec, lo, hi, tied, ikp, kp, seed, lam,
               gW_enc, gb_enc, gW_dec, gb_dec):
        from ._lib import _ptr as P
        c = self.ctx
        B, H = dh.shape
        c.check(c.lib.dae_train_shard_finish(
            c.h, P(dh), P(x[0]), P(x[1]), P(x[2]), P(W_enc), P(b_enc), P(W_dec), P(b_dec),
            lo, hi, H, B, 1 if tied else 0, float(ikp), float(kp), int(seed), float(lam),
            P(gW_enc), P(gb_enc), P(gW_dec), P(gb_dec)))

    def adam(self, p, m, v, g, lr, t):
        from ._lib import _ptr as P
        c = self.ctx
        c.check(c.lib.dae_adam_step(c.h, P(p), P(m), P(v), P(g), p.numel(), float(lr), 0.9, 0.999,
                                    1e-8, int(t)))


class ShardedTrainer:
    """One training step of DAE_tied / DAE (models/DAEs.py:98-102 of the reference) with the [V, H]
    matrices row-sharded over the ranks of `group`.

    Rank g owns rows [lo_g, hi_g) of W_enc, W_dec (untied) and b_dec together with their Adam
    moments; b_enc and its moments are replicated (its gradient is identical on all ranks after the
    dh all-reduce, so the replicas stay in lock-step without a broadcast).  Per step the path
    exchanges exactly two [B, H] fp32 tensors (pre-activation partials, dh partials) plus the
    scalar cost: all-reduce(sum) over RCCL.  `stages` carries the device work (HipTrainStages);
    it is injected so the exchange / bookkeeping logic runs under gloo on CPU in the tests.

    full_params: [W_enc, W_dec, b_enc, b_dec] host arrays (d_params order, DAEs.py:61/:138) that
    every rank slices identically -- a sharded run starts from the same point as an unsharded one.
    """

    def __init__(self, full_params, n_batch, lr, reg_lambda, tied, stages, device="cpu",
                 rank=0, world=1, group=None, seed=0):
        import torch
        W_enc, W_dec, b_enc, b_dec = full_params
        self.V, self.H = W_enc.shape
        self.n_batch, self.lr, self.lam, self.tied = int(n_batch), float(lr), float(reg_lambda), bool(tied)
        self.rank, self.world, self.group, self.stages = rank, world, group, stages
        self.lo, self.hi = shard_bounds(self.V, world, rank)
        self.device = device
        sl = slice(self.lo, self.hi)

        def dev(a):
            return torch.from_numpy(np.array(a, dtype=np.float32, order="C", copy=True)).to(device)   # own copy
        self.W_enc = dev(W_enc[sl])
        self.W_dec = self.W_enc if self.tied else dev(W_dec[sl])
        self.b_dec = dev(b_dec[sl])
        self.b_enc = dev(b_enc)
        names = ["W_enc", "b_enc", "b_dec"] + ([] if self.tied else ["W_dec"])
        self.params = {n: getattr(self, n) for n in names}
        self.grads = {n: torch.zeros_like(p) for n, p in self.params.items()}
        self.moments = {n: (torch.zeros_like(p), torch.zeros_like(p)) for n, p in self.params.items()}
        self.pre = torch.zeros((self.n_batch, self.H), dtype=torch.float32, device=device)
        self.dh = torch.zeros_like(self.pre)
        self.cost = torch.zeros(1, dtype=torch.float32, device=device)
        self.step = 0
        self._rng = np.random.RandomState(seed)        # same stream on every rank: same dropout draws

    def _allreduce(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)

    def train_step(self, x_csr, y_csr, keep_prob, input_keep_prob, fetch_cost=True):
        """x_csr / y_csr: (row_ptr, col, val) device tensors of the WHOLE batch with global column
        ids (models.DAEs.coo_to_csr).  Returns the global cost: a float, or with fetch_cost=False a 0-dim tensor
        on the trainer's device (no host synchronisation in the step: the reader builds the next batch meanwhile)."""
        st, g = self.stages, self.grads
        seed = int(self._rng.randint(0, 2 ** 31 - 1))
        st.encode(x_csr, self.W_enc, self.lo, self.hi, input_keep_prob, seed, self.pre)
        self._allreduce(self.pre)
        gW_out = g["W_enc"] if self.tied else g["W_dec"]
        st.decode(self.pre, self.b_enc, y_csr, self.W_enc, None if self.tied else self.W_dec, self.b_dec,
                  self.lo, self.hi, self.n_batch, self.tied, keep_prob, seed, self.lam,
                  gW_out, g["b_dec"], self.dh, self.cost)
        self._allreduce(self.dh)
        self._allreduce(self.cost)
        st.finish(self.dh, x_csr, self.W_enc, self.b_enc, None if self.tied else self.W_dec, self.b_dec,
                  self.lo, self.hi, self.tied, input_keep_prob, keep_prob, seed, self.lam,
                  g["W_enc"], g["b_enc"], None if self.tied else g["W_dec"], g["b_dec"])
        self.step += 1
        for n, p in self.params.items():
            m, v = self.moments[n]
            st.adam(p, m, v, g[n], self.lr, self.step)
        return float(self.cost.item()) if fetch_cost else self.cost[0].clone()

    def gather_params(self, as_numpy=True):
        """The full d_params list [W_enc, W_dec, b_enc, b_dec] (DAEs.py:61/:138) on every rank:
        all-gather of the row shards (padded to the largest shard, then trimmed).  as_numpy=False keeps the
        gathered tensors on the trainer's device (the model copies them straight into its replica)."""
        import torch
        if self.world == 1:
            full = {n: p for n, p in self.params.items()}
        else:
            import torch.distributed as dist
            bounds = all_shard_bounds(self.V, self.world)
            rows = max(hi - lo for lo, hi in bounds)
            full = {"b_enc": self.b_enc}
            for n in self.params:
                if n == "b_enc":
                    continue
                p = self.params[n]
                pad = torch.zeros((rows,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
                pad[:p.shape[0]] = p
                out = torch.empty((self.world * rows,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)
                dist.all_gather_into_tensor(out, pad, group=self.group)
                out = out.view((self.world, rows) + tuple(p.shape[1:]))
                full[n] = torch.cat([out[r, :hi - lo] for r, (lo, hi) in enumerate(bounds)], dim=0)
        if not as_numpy:
            return [full["W_enc"], full["W_enc"] if self.tied else full["W_dec"], full["b_enc"], full["b_dec"]]
        W_enc = full["W_enc"].cpu().numpy()
        W_dec = W_enc if self.tied else full["W_dec"].cpu().numpy()
        return [W_enc, W_dec, full["b_enc"].cpu().numpy(), full["b_dec"].cpu().numpy()]
