"""Vocabulary (output-column) sharding of the scoring path across the GPUs of one node
(BASELINE.json configs[2]).  One process per GPU, torch.distributed over RCCL.

Rank g owns the decoder rows (vocabulary columns) [lo_g, hi_g) -- tile aligned so every shard's
packed image starts on a 32-column MFMA tile -- and W_enc is replicated, so every rank computes
the same hidden activations with no collective.  The only exchange step of the path is the
all-gather of the per-shard top-k lists ((logit, column) x k per playlist), after which every
rank merges the G lists with the same (logit desc, column asc) rule.  The merge is exact because
each shard's top-k contains that shard's share of the global top-k.
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
    dist.all_gather_into_tensor(g_logit, local_logit.contiguous(), group=group)
    dist.all_gather_into_tensor(g_idx, local_idx.contiguous(), group=group)
    return g_logit.view(world, B, k), g_idx.view(world, B, k)


class ShardedRanker:
    """decode + top-k over this rank's vocabulary shard, exchange, merge.

    `local_topk(h, k) -> (logit [B,k], idx [B,k])` and `merge(g_logit, g_idx) -> (score, idx)` are
    the two device operations (libdae_hip: dae_decode_topk with DAE_OUT_LOGIT and dae_topk_merge);
    they are injected so the exchange logic is testable with gloo on CPU."""

    def __init__(self, local_topk, merge, group=None):
        self.local_topk = local_topk
        self.merge = merge
        self.group = group

    def rank_batch(self, h, k):
        l_logit, l_idx = self.local_topk(h, k)
        g_logit, g_idx = gather_shard_topk(l_logit, l_idx, self.group)
        return self.merge(g_logit, g_idx)
