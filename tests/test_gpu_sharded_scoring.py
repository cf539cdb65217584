"""GPU (-m gpu): vocabulary-sharded scoring through the PRODUCT objects (BASELINE.json configs[2]).

1. sharding.ShardedRanker + HipRankStages (what DAE.shard_scoring and bench.py build) with FOUR shard contexts on
   one device at configs[2]'s shape -- batch 1024, |vocab| = 170 000 -- for both exchanges and every rank's view:
   bit-identical to the unsharded fused path.
2. `main.py --challenge` under a 2-rank process group on ONE device (gloo: RCCL refuses two ranks on a device; the
   device tensors of the exchange go through the host) on the golden challenge file: the result pickle equals the
   1-rank run's.  Real kernels, real driver, real DAE.shard_scoring; only the transport differs from the 8-GPU job."""
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest

from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.sharding import HipRankStages, ShardedRanker, all_shard_bounds, row_owner_bounds
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype,world,B", [(_lib.DAE_DTYPE_F32, 4, 1024), (_lib.DAE_DTYPE_BF16, 4, 1024),
                                           (_lib.DAE_DTYPE_F32, 8, 512), (_lib.DAE_DTYPE_F32, 8, 2048),     # 2048 rows: the wave-per-row threshold kernel
                                           (_lib.DAE_DTYPE_BF16_EXACT, 4, 1024), (_lib.DAE_DTYPE_BF16_EXACT, 8, 512)])
def test_shard_contexts_full_size_equal_unsharded(dtype, world, B):
    import torch
    V, nt, H, k = 170000, 140000, 256, 500
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
    rp, col, val = coo_to_csr(pos, ones, B, V)
    srp, sc = seeds_to_csr(seeds, B, nt)
    d_We, d_be, d_Wd, d_bd = _dev(W_enc), _dev(b_enc), _dev(W_dec), _dev(b_dec)
    feed = tuple(_dev(a) for a in (rp, col, val, srp, sc))
    full = _lib.Context(0)
    full.prepack_decoder(d_Wd, d_bd, 0, V, dtype=dtype)
    s0 = torch.empty((B, k), device="cuda"); i0 = torch.empty((B, k), dtype=torch.int32, device="cuda")
    full.score_topk(feed[0], feed[1], feed[2], d_We, d_be, nt, feed[3], feed[4], k, s0, i0, dtype=dtype)
    if dtype == _lib.DAE_DTYPE_BF16_EXACT:       # the exact mode's lists ARE the fp32 path's lists
        full.prepack_decoder(d_Wd, d_bd, 0, V)
        s32 = torch.empty_like(s0); i32 = torch.empty_like(i0)
        full.score_topk(feed[0], feed[1], feed[2], d_We, d_be, nt, feed[3], feed[4], k, s32, i32)
        assert torch.equal(i0, i32) and torch.equal(s0.view(torch.int32), s32.view(torch.int32))
    ctxs, stages = [], []
    for lo, hi in all_shard_bounds(V, world):
        c = _lib.Context(0)
        c.prepack_decoder(d_Wd, d_bd, lo, hi, dtype=dtype)
        ctxs.append(c)
        stages.append(HipRankStages(c, d_We, d_be, nt, dtype))
    if world == 8:
        assert all_shard_bounds(V, world)[-1][0] > nt      # the last shard holds artist columns only: empty lists

    def gather_for(rank, exchange):
        # stands in for the collective: this process holds every shard, so "receiving" a peer's list = computing it
        def gather(l_logit, l_idx):
            lists = [(l_logit, l_idx) if g == rank else
                     tuple(t.clone() for t in stages[g].local_topk(feed, k)) for g in range(world)]
            gl, gi = torch.stack([a for a, _ in lists]), torch.stack([b for _, b in lists])
            if exchange == "alltoall":
                r0, r1 = row_owner_bounds(B, world, rank)
                gl, gi = gl[:, r0:r1].contiguous(), gi[:, r0:r1].contiguous()
            return gl, gi
        return gather
    for rank in range(world):
        st = stages[rank]
        s, i = ShardedRanker(st.local_topk, st.merge, exchange="allgather",
                             gather=gather_for(rank, "allgather")).rank_batch(feed, k)
        assert torch.equal(i, i0) and torch.equal(s, s0), rank
        s, i = ShardedRanker(st.local_topk, st.merge, exchange="alltoall",
                             gather=gather_for(rank, "alltoall")).rank_batch(feed, k)
        r0, r1 = row_owner_bounds(B, world, rank)
        assert i.shape == (r1 - r0, k) and torch.equal(i, i0[r0:r1]) and torch.equal(s, s0[r0:r1]), rank
    for c in ctxs + [full]:
        c.close()


def _write_run(tmp_path, name, extra=""):
    run = tmp_path / name
    run.mkdir()
    ini = open(os.path.join(G, "config.ini")).read()
    ini = ini.replace("[CHALLENGE]", "[CHALLENGE]\nallow_no_title = True" + extra)
    open(run / "config.ini", "w").write(ini)
    return run


@pytest.mark.parametrize("exchange", ["allgather", "alltoall"])
def test_two_rank_challenge_cli_on_one_device_equals_one_rank(tmp_path, exchange):
    import json
    shutil.copytree(os.path.join(G, "data"), tmp_path / "data")
    tr = json.load(open(tmp_path / "data" / "train"))
    nt = len(tr["track_uri2id"]); V = nt + len(tr["artist_uri2id"])
    W_enc, b_enc, W_dec, b_dec = make_weights(V, 32, seed=8, bias="zipf", n_tracks=nt)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k_, None)
    results = {}
    for name, world, extra in (("one", 1, ""), ("two", 2, "\nshard_exchange = " + exchange)):
        run = _write_run(tmp_path, name, extra)
        with open(run / "w_dae", "wb") as f:
            pickle.dump([W_enc, W_dec, b_enc, b_dec], f)
        cmd = [sys.executable, "-m", "spotify_recsys_challenge_2018_amd.main", "--dir", name, "--challenge"]
        procs = []
        for r in range(world):
            e = dict(env)
            if world > 1:
                e.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                         MASTER_PORT=str(29700 + os.getpid() % 200 + (7 if exchange == "alltoall" else 0)),
                         DAE_DIST_BACKEND="gloo")
            procs.append(subprocess.Popen(cmd, cwd=tmp_path, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for p in procs:
            out, _ = p.communicate(timeout=600)
            assert p.returncode == 0, out.decode()[-3000:]
        res = tmp_path / "challenge_results" / "result_inorder_5to100"
        results[name] = pickle.load(open(res, "rb"))
        os.remove(res)
        log = open(run / "log.txt").read()
        assert ("sharded over 2 ranks (%s exchange" % exchange in log) == (world == 2)
    assert len(results["one"]) == 13 and results["two"] == results["one"]
