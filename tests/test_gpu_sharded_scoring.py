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
from spotify_recsys_challenge_2018_amd.sharding import (HipRankStages, ShardedRanker, prepack_scoring_shard, row_owner_bounds,
                                                        scoring_shard)
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("dtype,world,B", [(_lib.DAE_DTYPE_F32, 4, 1024), (_lib.DAE_DTYPE_BF16, 4, 1024),
                                           (_lib.DAE_DTYPE_F32, 8, 512), (_lib.DAE_DTYPE_F32, 8, 2048),     # 2048 rows: the wave-per-row threshold kernel
                                           (_lib.DAE_DTYPE_BF16_EXACT, 4, 1024), (_lib.DAE_DTYPE_BF16_EXACT, 8, 512),
                                           # BASELINE.json configs[2] literally: batch 1024, the vocabulary in 8 column shards
                                           (_lib.DAE_DTYPE_F32, 8, 1024), (_lib.DAE_DTYPE_BF16, 8, 1024), (_lib.DAE_DTYPE_BF16_EXACT, 8, 1024)])
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
    ctxs, stages, keep = [], [], []
    for g in range(world):
        # every rank: an equal slice of the track columns (ranked) and of the artist columns (decoded, never ranked)
        shard = scoring_shard(nt, V, world, g)
        c = _lib.Context(0)
        bound, rows = prepack_scoring_shard(c, d_Wd, d_bd, shard, dtype)
        assert bound == shard[0][1] and shard[0][1] - shard[0][0] >= nt // world - 32 and shard[1][1] > shard[1][0]
        ctxs.append(c); keep.append(rows)
        stages.append(HipRankStages(c, d_We, d_be, bound, dtype))
    for g in range(world):                                   # no rank returns empty lists, ids are global and in its slice
        lg, ig = stages[g].local_topk(feed, k)
        lo_g, hi_g = scoring_shard(nt, V, world, g)[0]
        ig = ig.cpu().numpy()
        assert (ig[:, 0] >= lo_g).all() and (ig[ig >= 0] >= lo_g).all() and (ig < hi_g).all()

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


@pytest.mark.parametrize("dtype,world,B", [(_lib.DAE_DTYPE_F32, 8, 2048), (_lib.DAE_DTYPE_F32, 4, 1024),
                                           (_lib.DAE_DTYPE_BF16_EXACT, 8, 512), (_lib.DAE_DTYPE_BF16, 4, 1024)])
def test_threshold_exchange_same_lists_from_shorter_shard_lists(dtype, world, B):
    """dae_score_topk_begin / _finish: every shard's own bound of the row's k-th largest logit,
    their element-wise maximum (what one all-gather of 4 bytes per row and rank delivers), the filter launches with
    THAT: the merged lists are the unsharded call's, bit for bit, and the shards together return little more than k
    candidates per row instead of world x k."""
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
    # begin + finish with the call's own threshold == the one-call form (same context)
    tau1 = torch.empty(B, device="cuda")
    s1 = torch.empty_like(s0); i1 = torch.empty_like(i0)
    full.score_topk_begin(feed[0], feed[1], feed[2], d_We, d_be, nt, feed[3], k, tau1, dtype=dtype)
    full.score_topk_finish(tau1, feed[3], feed[4], s1, i1)
    assert torch.equal(i1, i0) and torch.equal(s1, s0)
    with pytest.raises(_lib.DaeError):
        full.score_topk_finish(tau1, feed[3], feed[4], s1, i1)          # no call in progress any more
    ctxs, stages, keep = [], [], []
    for g in range(world):
        c = _lib.Context(0)
        bound, rows = prepack_scoring_shard(c, d_Wd, d_bd, scoring_shard(nt, V, world, g), dtype)
        ctxs.append(c); keep.append(rows)
        stages.append(HipRankStages(c, d_We, d_be, bound, dtype))
    taus = torch.stack([st.local_begin(feed, k).clone() for st in stages])
    assert torch.isfinite(taus).all()
    tau_max = taus.amax(0)
    lists = [tuple(t.clone() for t in st.local_finish(feed, k, tau_max)) for st in stages]
    gl, gi = torch.stack([a for a, _ in lists]), torch.stack([b for _, b in lists])
    s, i = stages[0].merge(gl, gi)
    assert torch.equal(i, i0) and torch.equal(s, s0)
    per_row = (gi >= 0).sum(0).sum(1).float()                            # candidates all shards return for a row
    assert float(per_row.min()) >= k and float(per_row.mean()) < 1.6 * k, (float(per_row.mean()), world * k)
    # ... and through ShardedRanker as rank 2 sees it (the two collectives replaced by the local stand-ins)
    r = 2
    ranker = ShardedRanker(stages[r].local_topk, stages[r].merge, exchange="allgather",
                           local_begin=stages[r].local_begin, local_finish=stages[r].local_finish,
                           gather_tau=lambda t: taus, gather=lambda a, b: (gl, gi))
    s, i = ranker.rank_batch(feed, k)
    assert torch.equal(i, i0) and torch.equal(s, s0)
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
    tau_x = "\nshard_tau_exchange = False" if exchange == "allgather" else ""        # (default: on)
    for name, world, extra in (("one", 1, ""), ("two", 2, "\nshard_exchange = " + exchange + tau_x)):
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
