"""GPU (-m gpu): the vocabulary-row sharded training step (include/dae_hip.h dae_train_shard_*;
SURVEY.md 8e).  The shards of a step are run one after another on the one GPU of the box, the two
all-reduces done by hand, and the concatenated gradients compared with (i) the float64 numpy
restatement (same tolerance as test_gpu_train.py) and (ii) the unsharded C-ABI step."""
import numpy as np
import pytest

import oracle
from oracle import dae_numpy as dn
from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr
from spotify_recsys_challenge_2018_amd.sharding import HipTrainStages, ShardedTrainer, all_shard_bounds
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _uniform(seed, stream, rows, cols):
    l = oracle.lib()
    return np.array([[l.orc_uniform(seed, stream, int(r), int(c)) for c in cols] for r in rows], np.float32)


@pytest.mark.parametrize("V,nt,H,B,tied,lam,ikp,kp,world", [
    (3000, 2400, 128, 37, False, 0.0, 0.75, 0.8, 2),
    (1500, 1200, 64, 64, True, 0.01, 1.0, 0.8, 3),
    (2100, 2000, 256, 250, True, 0.0, 0.75, 1.0, 2),
    (2100, 2000, 256, 250, False, 0.02, 1.0, 1.0, 1),
])
def test_sharded_stages_match_float64_and_unsharded(V, nt, H, B, tied, lam, ikp, kp, world):
    import torch
    ctx = _lib.Context(0)
    st = HipTrainStages(ctx)
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt, tied=tied)
    b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6, seed_counts=(3, 9, 20))
    xcsr = coo_to_csr(pos, ones, B, V)
    ycsr = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)
    x = tuple(_dev(a) for a in xcsr); y = tuple(_dev(a) for a in ycsr)
    seed = 977
    bounds = all_shard_bounds(V, world)
    be = _dev(b_enc)
    sh = []
    for lo, hi in bounds:
        d = dict(lo=lo, hi=hi, We=_dev(W_enc[lo:hi]), bd=_dev(b_dec[lo:hi]),
                 Wd=None if tied else _dev(W_dec[lo:hi]))
        d.update(gWe=torch.zeros((hi - lo, H), device="cuda"), gbd=torch.zeros(hi - lo, device="cuda"),
                 gWd=None if tied else torch.zeros((hi - lo, H), device="cuda"),
                 gbe=torch.zeros(H, device="cuda"), pre=torch.zeros((B, H), device="cuda"),
                 dh=torch.zeros((B, H), device="cuda"), cost=torch.zeros(1, device="cuda"))
        sh.append(d)
    for d in sh:
        st.encode(x, d["We"], d["lo"], d["hi"], ikp, seed, d["pre"])
    pre = sum(d["pre"] for d in sh)                                   # all-reduce #1
    # the decode and finish stages of one shard share ctx scratch: run them back to back per shard,
    # which needs the reduced dh -> first pass computes the partials, second pass finishes
    for d in sh:
        st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                  d["gWe"] if tied else d["gWd"], d["gbd"], d["dh"], d["cost"])
    dh = sum(d["dh"] for d in sh)                                     # all-reduce #2
    cost = float(sum(d["cost"] for d in sh).item())
    for d in sh:
        if world > 1:      # re-establish this shard's h / sigmoid in the ctx scratch (one ctx, many shards)
            keep = (d["gWe"] if tied else d["gWd"]).clone(), d["gbd"].clone()
            st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                      d["gWe"] if tied else d["gWd"], d["gbd"], d["dh"], d["cost"])
            assert torch.equal(keep[0], d["gWe"] if tied else d["gWd"]) and torch.equal(keep[1], d["gbd"])
        st.finish(dh, x, d["We"], be, d["Wd"], d["bd"], d["lo"], d["hi"], tied, ikp, kp, seed, lam,
                  d["gWe"], d["gbe"], d["gWd"], d["gbd"])
    torch.cuda.synchronize()
    gWe = torch.cat([d["gWe"] for d in sh]).cpu().numpy()
    gbd = torch.cat([d["gbd"] for d in sh]).cpu().numpy()
    gWd = None if tied else torch.cat([d["gWd"] for d in sh]).cpu().numpy()
    for d in sh[1:]:
        assert torch.equal(d["gbe"], sh[0]["gbe"])                   # replicated and identical
    gbe = sh[0]["gbe"].cpu().numpy()

    # (i) float64 restatement with the same dropout draws
    xd = dn.sparse_to_dense(pos, ones, B, V)
    yd = dn.sparse_to_dense(pos, np.ones(len(pos), np.float32), B, V)
    im = None
    if ikp < 1.0:
        im = np.ones((B, V))
        for r in range(B):
            cols = xcsr[1][xcsr[0][r]:xcsr[0][r + 1]]
            im[r, cols] = np.floor(np.float32(ikp) + _uniform(seed, 0, [r], cols)[0])
    hm = np.floor(np.float32(kp) + _uniform(seed, 1, range(B), range(H))) if kp < 1.0 else None
    ref = dn.grads(xd, yd, W_enc, b_enc, W_dec, b_dec, n_batch=B, tied=tied, reg_lambda=lam,
                   input_keep_mask=im, ikp=ikp, hidden_keep_mask=hm, kp=kp)
    assert abs(cost - ref["cost"]) <= 1e-5 * abs(ref["cost"])
    tol = dict(rtol=2e-4, atol=2e-7)
    assert np.allclose(gbd, ref["gb_dec"], **tol)
    assert np.allclose(gbe, ref["gb_enc"], **tol)
    assert np.allclose(gWe, ref["gW_enc"], **tol)
    if not tied:
        assert np.allclose(gWd, ref["gW_dec"], **tol)

    # (ii) the unsharded entry point on the same inputs
    P = _lib._ptr
    d0 = dict(We=_dev(W_enc), Wd=_dev(W_dec), bd=_dev(b_dec))
    uWe = torch.zeros((V, H), device="cuda"); ube = torch.zeros(H, device="cuda")
    uWd = torch.zeros((V, H), device="cuda"); ubd = torch.zeros(V, device="cuda")
    ucost = torch.zeros(1, device="cuda")
    ctx.check(ctx.lib.dae_train_forward_backward(
        ctx.h, P(x[0]), P(x[1]), P(x[2]), P(y[0]), P(y[1]), P(y[2]), P(d0["We"]), P(be), P(d0["Wd"]), P(d0["bd"]),
        V, H, B, B, 1 if tied else 0, float(ikp), float(kp), seed, float(lam),
        P(uWe), P(ube), None if tied else P(uWd), P(ubd), P(ucost)))
    torch.cuda.synchronize()
    assert abs(cost - float(ucost.item())) <= 2e-6 * abs(cost)
    assert np.allclose(gWe, uWe.cpu().numpy(), **tol) and np.allclose(gbd, ubd.cpu().numpy(), **tol)
    assert np.allclose(gbe, ube.cpu().numpy(), **tol)
    ctx.close()


def test_sharded_trainer_world1_follows_model_train_step():
    """ShardedTrainer (world 1, HIP stages) and the model's own train_step walk the same costs."""
    import torch
    from spotify_recsys_challenge_2018_amd.models.DAEs import DAE

    class C:
        save = "/tmp/_st_unused"; batch = 48; n_input = 2600; hidden = 64; lr = 0.005; reg_lambda = 0.0
        initval = "NULL"; n_tracks = 2000
    conf = C()
    model = DAE(conf)
    model.fit()
    params = [p.copy() for p in model.get_params()]
    pos, ones, _ = make_playlists(conf.batch, 2000, 600, seed=9)
    xh = coo_to_csr(pos, ones, conf.batch, conf.n_input)
    yh = coo_to_csr(pos, np.ones(len(pos), np.float32), conf.batch, conf.n_input)
    tr = ShardedTrainer(params, conf.batch, conf.lr, 0.0, False, HipTrainStages(model.ctx), device="cuda")
    x = tuple(_dev(a) for a in xh); y = tuple(_dev(a) for a in yh)
    a = [tr.train_step(x, y, 1.0, 1.0) for _ in range(4)]
    b = [model.train_step(pos, ones, pos, np.ones(len(pos), np.float32), 1.0, 1.0) for _ in range(4)]
    assert a[-1] < a[0]
    assert np.allclose(a, b, rtol=1e-4)
    got, want = tr.gather_params(), model.get_params()
    for g, w in zip(got, want):
        assert float(np.mean(np.abs(g - w) > 1e-3)) < 1e-3


def test_model_shard_training_world1_syncs_replica_before_scoring():
    """DAE_tied.shard_training: training goes through the sharded stages, the inference replica is
    refreshed (sync_params) before recommend / get_params see the weights."""
    from spotify_recsys_challenge_2018_amd.models.DAEs import DAE_tied

    class C:
        save = "/tmp/_st_unused2"; batch = 32; n_input = 1800; hidden = 64; lr = 0.01; reg_lambda = 0.0
        n_tracks = 1500
    a, b = DAE_tied(C()), DAE_tied(C())
    a.fit(); b.fit()
    b.shard_training(0, 1)
    pos, ones, seeds = make_playlists(C.batch, 1500, 300, seed=3)
    yo = np.ones(len(pos), np.float32)
    ca = [a.train_step(pos, ones, pos, yo, 1.0, 1.0) for _ in range(3)]
    cb = [b.train_step(pos, ones, pos, yo, 1.0, 1.0) for _ in range(3)]
    assert np.allclose(ca, cb, rtol=1e-4)
    assert b._params_stale
    ia, _ = a.recommend(pos, ones, seeds, k=100)
    ib, _ = b.recommend(pos, ones, seeds, k=100)
    assert not b._params_stale
    # the two runs differ by fp32 re-association only: the rankings agree almost everywhere
    assert np.mean(ia[:, :20] == ib[:, :20]) > 0.9
    wa, wb = a.get_params(), b.get_params()
    assert wb[0] is not None and float(np.mean(np.abs(wa[0] - wb[0]) > 1e-3)) < 1e-3


@pytest.mark.parametrize("tied,world", [(False, 2), (True, 3)])
def test_sharded_stages_with_bf16_gemms_match_the_unsharded_bf16_step(tied, world):
    """dae_set_train_dtype(BF16) on the sharded stages (hidden 256: the row-major bf16 K5, the transposed K6 and the
    bf16 K7 on every shard): concatenated gradients and the summed cost against the unsharded bf16 step on the same
    draws.  The per-element arithmetic is the same; only K7's split over vocabulary chunks differs -> 1e-3 of the norm."""
    import torch
    V, nt, H, B = 4200, 3800, 256, 200
    ctx = _lib.Context(0)
    ctx.set_train_dtype(_lib.DAE_DTYPE_BF16)
    st = HipTrainStages(ctx)
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt, tied=tied)
    b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6, seed_counts=(3, 9, 20))
    x = tuple(_dev(a) for a in coo_to_csr(pos, ones, B, V))
    y = tuple(_dev(a) for a in coo_to_csr(pos, np.ones(len(pos), np.float32), B, V))
    seed, ikp, kp, lam = 977, 0.75, 0.8, 0.0
    be = _dev(b_enc)
    sh = []
    for lo, hi in all_shard_bounds(V, world):
        d = dict(lo=lo, hi=hi, We=_dev(W_enc[lo:hi]), bd=_dev(b_dec[lo:hi]), Wd=None if tied else _dev(W_dec[lo:hi]))
        d.update(gWe=torch.zeros((hi - lo, H), device="cuda"), gbd=torch.zeros(hi - lo, device="cuda"),
                 gWd=None if tied else torch.zeros((hi - lo, H), device="cuda"), gbe=torch.zeros(H, device="cuda"),
                 pre=torch.zeros((B, H), device="cuda"), dh=torch.zeros((B, H), device="cuda"),
                 cost=torch.zeros(1, device="cuda"))
        sh.append(d)
    for d in sh:
        st.encode(x, d["We"], d["lo"], d["hi"], ikp, seed, d["pre"])
    pre = sum(d["pre"] for d in sh)
    for d in sh:
        st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                  d["gWe"] if tied else d["gWd"], d["gbd"], d["dh"], d["cost"])
    dh = sum(d["dh"] for d in sh)
    cost = float(sum(d["cost"] for d in sh).item())
    for d in sh:
        st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                  d["gWe"] if tied else d["gWd"], d["gbd"], d["dh"], d["cost"])      # re-establish this shard's scratch
        st.finish(dh, x, d["We"], be, d["Wd"], d["bd"], d["lo"], d["hi"], tied, ikp, kp, seed, lam,
                  d["gWe"], d["gbe"], d["gWd"], d["gbd"])
    torch.cuda.synchronize()
    got = dict(We=torch.cat([d["gWe"] for d in sh]), bd=torch.cat([d["gbd"] for d in sh]), be=sh[0]["gbe"])
    if not tied:
        got["Wd"] = torch.cat([d["gWd"] for d in sh])
    P = _lib._ptr
    d0 = dict(We=_dev(W_enc), Wd=_dev(W_dec), bd=_dev(b_dec))
    u = dict(We=torch.zeros((V, H), device="cuda"), be=torch.zeros(H, device="cuda"),
             Wd=torch.zeros((V, H), device="cuda"), bd=torch.zeros(V, device="cuda"))
    ucost = torch.zeros(1, device="cuda")
    ctx.check(ctx.lib.dae_train_forward_backward(
        ctx.h, P(x[0]), P(x[1]), P(x[2]), P(y[0]), P(y[1]), P(y[2]), P(d0["We"]), P(be), P(d0["Wd"]), P(d0["bd"]),
        V, H, B, B, 1 if tied else 0, float(ikp), float(kp), seed, float(lam),
        P(u["We"]), P(u["be"]), None if tied else P(u["Wd"]), P(u["bd"]), P(ucost)))
    torch.cuda.synchronize()
    assert abs(cost - float(ucost.item())) <= 1e-4 * abs(cost)
    for k in got:
        err = float((got[k].double() - u[k].double()).norm() / u[k].double().norm())
        assert err <= 1e-3, (k, err)
    ctx.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_size_eight_shards_equal_the_unsharded_step(dtype):
    """BASELINE.json configs[3] "1 and 8 GPUs" at FULL size (V = 170 000, H = 256, B = 256, untied): the three shard
    stages of all eight vocabulary-row shards (~21 250 rows each, tile aligned; shard 7 holds artist rows only) run back to back on
    the one GPU of the box, the two all-reduces done by hand -- cost, gb_enc and every shard's slice of gW_enc /
    gW_dec / gb_dec against the unsharded entry point on the same draws (reference step: main_train.py:193-213).
    fp32: 2e-4 of each tensor's norm (K7's split over vocabulary chunks re-associates); bf16 GEMMs: 1e-3."""
    import torch
    V, nt, H, B, world, tied = 170000, 140000, 256, 256, 8, False
    ctx = _lib.Context(0)
    if dtype == "bf16":
        ctx.set_train_dtype(_lib.DAE_DTYPE_BF16)
    st = HipTrainStages(ctx)
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt, tied=tied)
    b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6)
    x = tuple(_dev(a) for a in coo_to_csr(pos, ones, B, V))
    y = tuple(_dev(a) for a in coo_to_csr(pos, np.ones(len(pos), np.float32), B, V))
    seed, ikp, kp, lam = 977, 0.75, 0.8, 0.0
    be = _dev(b_enc)
    d_We, d_Wd, d_bd = _dev(W_enc), _dev(W_dec), _dev(b_dec)
    bounds = all_shard_bounds(V, world)
    assert bounds[0][0] == 0 and bounds[7][1] == V and bounds[7][0] >= nt    # ~21 250 rows each; shard 7: artist rows only
    sh = []
    for lo, hi in bounds:
        d = dict(lo=lo, hi=hi, We=d_We[lo:hi], bd=d_bd[lo:hi], Wd=d_Wd[lo:hi])        # views: a rank holds just its rows
        d.update(gWe=torch.zeros((hi - lo, H), device="cuda"), gbd=torch.zeros(hi - lo, device="cuda"),
                 gWd=torch.zeros((hi - lo, H), device="cuda"), gbe=torch.zeros(H, device="cuda"),
                 pre=torch.zeros((B, H), device="cuda"), dh=torch.zeros((B, H), device="cuda"),
                 cost=torch.zeros(1, device="cuda"))
        sh.append(d)
    for d in sh:
        st.encode(x, d["We"], d["lo"], d["hi"], ikp, seed, d["pre"])
    pre = sum(d["pre"] for d in sh)                                   # all-reduce #1
    for d in sh:
        st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                  d["gWd"], d["gbd"], d["dh"], d["cost"])
    dh = sum(d["dh"] for d in sh)                                     # all-reduce #2
    cost = float(sum(d["cost"] for d in sh).item())
    for d in sh:
        st.decode(pre, be, y, d["We"], d["Wd"], d["bd"], d["lo"], d["hi"], B, tied, kp, seed, lam,
                  d["gWd"], d["gbd"], d["dh"], d["cost"])              # re-establish this shard's scratch (one ctx, many shards)
        st.finish(dh, x, d["We"], be, d["Wd"], d["bd"], d["lo"], d["hi"], tied, ikp, kp, seed, lam,
                  d["gWe"], d["gbe"], d["gWd"], d["gbd"])
    torch.cuda.synchronize()
    P = _lib._ptr
    u = dict(We=torch.zeros((V, H), device="cuda"), be=torch.zeros(H, device="cuda"),
             Wd=torch.zeros((V, H), device="cuda"), bd=torch.zeros(V, device="cuda"))
    ucost = torch.zeros(1, device="cuda")
    ctx.check(ctx.lib.dae_train_forward_backward(
        ctx.h, P(x[0]), P(x[1]), P(x[2]), P(y[0]), P(y[1]), P(y[2]), P(d_We), P(be), P(d_Wd), P(d_bd),
        V, H, B, B, 0, float(ikp), float(kp), seed, float(lam), P(u["We"]), P(u["be"]), P(u["Wd"]), P(u["bd"]), P(ucost)))
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == "f32" else 1e-3
    assert abs(cost - float(ucost.item())) <= (2e-6 if dtype == "f32" else 1e-4) * abs(cost)
    for d in sh[1:]:
        assert torch.equal(d["gbe"], sh[0]["gbe"])                   # replicated and identical on every rank
    err = float((sh[0]["gbe"].double() - u["be"].double()).norm() / u["be"].double().norm())
    assert err <= tol, ("gb_enc", err)
    for g, d in enumerate(sh):                                       # every rank's slices, shard 0 and shard 7 included
        for name, key in (("gWe", "We"), ("gWd", "Wd"), ("gbd", "bd")):
            ref = u[key][d["lo"]:d["hi"]].double()
            nrm = float(ref.norm())
            diff = float((d[name].double() - ref).norm())
            assert diff <= tol * max(nrm, 1e-30) + 1e-12, (g, name, diff, nrm)
    ctx.close()
