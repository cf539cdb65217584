"""Worker of tests/test_gpu_rccl.py: ONE rank of an N-rank job (launched by torch.distributed.run).  Every rank holds the
whole synthetic model, so it computes the unsharded answer on its own GPU and compares what the sharded objects of the
product (sharding.ShardedRanker / HipRankStages, ShardedTrainer / HipTrainStages -- what DAE.shard_scoring / shard_training
and bench.py build) return over the process group with it.  backend "nccl" = RCCL over xGMI (one device per rank);
"gloo" = the same flow with every rank on device 0 (the one-GPU rehearsal: device tensors of the collectives go through
the host).  Exit code 0 and a line "RCCL_WORKER_OK <case> rank r" on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from spotify_recsys_challenge_2018_amd import _lib                      # noqa: E402
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr      # noqa: E402
from spotify_recsys_challenge_2018_amd.sharding import (HipRankStages, HipTrainStages, ShardedRanker, ShardedTrainer,     # noqa: E402
                                                        prepack_scoring_shard, row_owner_bounds, scoring_shard)
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights            # noqa: E402


def main():
    case, backend = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # the collectives really span `world` ranks on `world` devices
    probe = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(probe)
    assert int(probe.item()) == world
    if backend == "nccl":
        ids = [None] * world
        dist.all_gather_object(ids, torch.cuda.current_device())
        assert sorted(ids) == list(range(world)), ids

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    if case.startswith("ranker"):
        V, nt, H, k = (170000, 140000, 256, 500) if case == "ranker_full" else (40000, 33000, 128, 500)
        B = 64 * world
        W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
        pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
        rp, col, val = coo_to_csr(pos, ones, B, V)
        srp, sc = seeds_to_csr(seeds, B, nt)
        d_We, d_be, d_Wd, d_bd = up(W_enc), up(b_enc), up(W_dec), up(b_dec)
        feed = tuple(up(a) for a in (rp, col, val, srp, sc if sc.size else np.zeros(1, np.int32)))
        for dtype in (_lib.DAE_DTYPE_F32, _lib.DAE_DTYPE_BF16_EXACT, _lib.DAE_DTYPE_BF16):
            full = _lib.Context(local)
            full.prepack_decoder(d_Wd, d_bd, 0, V, dtype=dtype)
            s0 = torch.empty((B, k), device=dev); i0 = torch.empty((B, k), dtype=torch.int32, device=dev)
            full.score_topk(feed[0], feed[1], feed[2], d_We, d_be, nt, feed[3], feed[4], k, s0, i0, dtype=dtype)
            torch.cuda.synchronize()
            c = _lib.Context(local)
            bound, keep = prepack_scoring_shard(c, d_Wd, d_bd, scoring_shard(nt, V, world, rank), dtype)
            st = HipRankStages(c, d_We, d_be, bound, dtype)
            r0, r1 = row_owner_bounds(B, world, rank)
            for exchange in ("allgather", "alltoall"):
                for tau_x in (False, True):
                    two = dict(local_begin=st.local_begin, local_finish=st.local_finish) if tau_x else {}
                    ranker = ShardedRanker(st.local_topk, st.merge, exchange=exchange, **two)
                    for _ in range(2):                       # twice: buffers reused, no stale state
                        s, i = ranker.rank_batch(feed, k)
                        torch.cuda.synchronize()
                        want_s, want_i = (s0, i0) if exchange == "allgather" else (s0[r0:r1], i0[r0:r1])
                        assert torch.equal(i, want_i), (dtype, exchange, tau_x, "indices")
                        assert torch.equal(s.view(torch.int32), want_s.view(torch.int32)), (dtype, exchange, tau_x, "scores")
            if dtype == _lib.DAE_DTYPE_BF16_EXACT:
                assert c.exact_guard_read() == (0, -1) and full.exact_guard_read() == (0, -1)
            dist.barrier()
            c.close(); full.close()
            del keep
    elif case == "trainer":
        V, nt, H, B = 6000, 5000, 128, 64
        W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt)
        b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
        steps = []
        for s_ in range(3):
            pos, ones, _ = make_playlists(B, nt, V - nt, seed=60 + s_, seed_counts=(3, 9, 20))
            steps.append((tuple(up(a) for a in coo_to_csr(pos, ones, B, V)),
                          tuple(up(a) for a in coo_to_csr(pos, np.ones(len(pos), np.float32), B, V))))
        out = {}
        for name, w_, r_, grp in (("sharded", world, rank, None), ("single", 1, 0, None)):
            ctx = _lib.Context(local)
            tr = ShardedTrainer([W_enc, W_dec, b_enc, b_dec], B, 0.005, 0.0, False, HipTrainStages(ctx), device=dev,
                                rank=r_, world=w_, group=grp, seed=5)
            costs = [tr.train_step(x, y, 0.8, 0.75) for x, y in steps]
            out[name] = (costs, tr.gather_params())
            torch.cuda.synchronize()
            ctx.close()
        (c_s, p_s), (c_1, p_1) = out["sharded"], out["single"]
        assert np.allclose(c_s, c_1, rtol=2e-5), (c_s, c_1)
        for a, b in zip(p_s, p_1):
            assert a.shape == b.shape
            # Adam normalises the update: after 3 steps every parameter moved <= 3 lr; the shards sum dh in another order
            assert np.max(np.abs(a - b)) <= 2e-3 * 0.005 * 3 + 1e-7 or np.mean(np.abs(a - b) > 1e-4) < 1e-4, float(np.max(np.abs(a - b)))
        # every rank holds the same gathered parameters
        chk = torch.tensor([float(np.float64(p_s[0]).sum()), float(np.float64(p_s[1]).sum())], dtype=torch.float64, device=dev)
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        assert torch.equal(lo_, hi_)
    else:
        raise SystemExit("unknown case %r" % case)
    dist.barrier()
    print("RCCL_WORKER_OK %s rank %d/%d backend %s devices %d" % (case, rank, world, backend, torch.cuda.device_count()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
