"""GPU (-m gpu): the training step (forward with dropout + weighted BCE + backward, TF-Adam)
through the C ABI against the float64 numpy restatement of DAEs.py:98-102 (oracle/dae_numpy.py).
The backward GEMMs sum in a different order than the reference would, so parity is by tolerance:
rtol 2e-4 / atol 2e-7 on gradients (fp32 sums over up to B*V terms), 1e-5 relative on the cost."""
import ctypes
import os
import pickle
import shutil

import numpy as np
import pytest

import oracle
from oracle import dae_numpy as dn
from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _uniform(seed, stream, rows, cols):
    l = oracle.lib()
    out = np.empty((len(rows), len(cols)), np.float32)
    for i, r in enumerate(rows):
        for j, c in enumerate(cols):
            out[i, j] = l.orc_uniform(seed, stream, int(r), int(c))
    return out


@pytest.mark.parametrize("V,nt,H,B,tied,lam,ikp,kp", [
    (3000, 2400, 128, 37, False, 0.0, 0.75, 0.8),
    (1500, 1200, 64, 64, True, 0.0, 1.0, 0.8),
    (900, 700, 32, 10, False, 0.01, 0.6, 1.0),
    (2100, 2000, 256, 250, True, 0.02, 0.75, 0.8),
    # round 6, fp32 K5 through LDS (hidden 256): waves without a playlist, a full batch with more tiles than workgroups
    (3000, 2500, 256, 37, False, 0.0, 0.75, 0.8),
    (20000, 16000, 256, 256, False, 0.0, 1.0, 1.0),
    (33, 20, 256, 256, False, 0.0, 0.75, 0.8),                   # two tiles, the second with one decoder row
    (31, 20, 256, 3, False, 0.0, 0.75, 0.8),                     # less than a tile, less than a wave's playlists
])
def test_train_step_gradients(V, nt, H, B, tied, lam, ikp, kp):
    import torch
    ctx = _lib.Context(0)
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt, tied=tied)
    b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6, seed_counts=(3, 9, 20))
    xr, xc, xv = coo_to_csr(pos[pos[:, 1] < nt], ones[pos[:, 1] < nt], B, V)     # tracks-only input
    yr, yc, yv = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)            # tracks+artists target
    seed = 4242
    d = dict(We=_dev(W_enc), be=_dev(b_enc), Wd=_dev(W_dec), bd=_dev(b_dec))
    gWe = torch.zeros((V, H), device="cuda"); gbe = torch.zeros(H, device="cuda")
    gWd = torch.zeros((V, H), device="cuda"); gbd = torch.zeros(V, device="cuda")
    cost = torch.zeros(1, device="cuda")
    P = _lib._ptr
    csr = [_dev(a) for a in (xr, xc, xv, yr, yc, yv)]        # keep the device copies alive
    ctx.check(ctx.lib.dae_train_forward_backward(
        ctx.h, P(csr[0]), P(csr[1]), P(csr[2]), P(csr[3]), P(csr[4]), P(csr[5]),
        P(d["We"]), P(d["be"]), P(d["Wd"]), P(d["bd"]), V, H, B, B, 1 if tied else 0,
        float(ikp), float(kp), seed, float(lam), P(gWe), P(gbe), None if tied else P(gWd), P(gbd), P(cost)))
    torch.cuda.synchronize()

    # reference: dense float64 with the SAME dropout draws
    x = dn.sparse_to_dense(pos[pos[:, 1] < nt], ones[pos[:, 1] < nt], B, V)
    y = dn.sparse_to_dense(pos, np.ones(len(pos), np.float32), B, V)
    im = None
    if ikp < 1.0:
        im = np.ones((B, V))
        for r in range(B):
            cols = xc[xr[r]:xr[r + 1]]
            u = _uniform(seed, 0, [r], cols)[0]
            im[r, cols] = np.floor(np.float32(ikp) + u)
    hm = np.floor(np.float32(kp) + _uniform(seed, 1, range(B), range(H))) if kp < 1.0 else None
    ref = dn.grads(x, y, W_enc, b_enc, W_dec, b_dec, n_batch=B, tied=tied, reg_lambda=lam,
                   input_keep_mask=im, ikp=ikp, hidden_keep_mask=hm, kp=kp)
    assert abs(float(cost.item()) - ref["cost"]) <= 1e-5 * abs(ref["cost"])
    tol = dict(rtol=2e-4, atol=2e-7)
    assert np.allclose(gbd.cpu().numpy(), ref["gb_dec"], **tol)
    assert np.allclose(gbe.cpu().numpy(), ref["gb_enc"], **tol)
    assert np.allclose(gWe.cpu().numpy(), ref["gW_enc"], **tol)
    if not tied:
        assert np.allclose(gWd.cpu().numpy(), ref["gW_dec"], **tol)
    ctx.close()


def test_adam_matches_tf_formulation():
    import torch
    ctx = _lib.Context(0)
    rng = np.random.default_rng(0)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    dp, dm, dv = _dev(p), _dev(m), _dev(v)
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32) * (rng.random(n) < 0.3)      # sparse gradient
        dg = _dev(g)
        ctx.check(ctx.lib.dae_adam_step(ctx.h, _lib._ptr(dp), _lib._ptr(dm), _lib._ptr(dv),
                                        _lib._ptr(dg), n, 0.005, 0.9, 0.999, 1e-8, t))
        p, m, v = dn.adam_tf(p, m, v, g, 0.005, t)
    # same fp32 operation sequence as the oracle: moments bit-equal, parameters within 1 ulp of the
    # divide/sqrt pair
    assert np.array_equal(dm.cpu().numpy(), m)
    assert np.array_equal(dv.cpu().numpy(), v)
    assert np.allclose(dp.cpu().numpy(), p, rtol=3e-7, atol=1e-9)
    ctx.close()


def test_pretrain_dae_challenge_drivers_end_to_end(tmp_path, capsys):
    """main.py --pretrain -> --dae -> --dae --testmode -> --challenge -> --title -> --challenge on the golden mini dataset
    (BASELINE.json configs[0] shape: tied pretrain, then untied DAE from the pretrain pickle)."""
    import random
    from spotify_recsys_challenge_2018_amd import main as cli
    work = tmp_path / "run"
    work.mkdir()
    shutil.copy(os.path.join(G, "config.ini"), work / "config.ini")
    shutil.copytree(os.path.join(G, "data"), tmp_path / "data")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        random.seed(0); np.random.seed(0)
        assert cli.main(["--dir", "run", "--pretrain"]) == 0
        w = pickle.load(open(work / "w_pretrain", "rb"))
        assert len(w) == 4 and w[0].dtype == np.float32 and np.array_equal(w[0], w[1])      # tied
        assert cli.main(["--dir", "run", "--dae"]) == 0
        w2 = pickle.load(open(work / "w_dae", "rb"))
        assert w2[0].shape == w[0].shape and not np.array_equal(w2[0], w2[1])              # untied
        log = open(work / "log.txt").read()
        losses = [float(l.split(":")[1]) for l in log.splitlines() if l.startswith("training loss")]
        assert len(losses) == 4 and losses[1] < losses[0] and losses[3] < losses[2]
        assert "rprecision" in log and "Parameters are saved" in log
        for split in ("test-0", "test-1", "test-5", "test-25r"):        # readme.md:69 seed patterns, every epoch
            assert log.count("seed num: %s rprecision" % split) == 4
        assert cli.main(["--dir", "run", "--dae", "--testmode"]) == 0
        with pytest.raises(FileNotFoundError):                          # no title variables yet: the reference fails
            cli.main(["--dir", "run", "--challenge"])                    # too (saver.restore); opt in explicitly
        ini = open(work / "config.ini").read()
        open(work / "config.ini", "w").write(ini.replace("[CHALLENGE]", "[CHALLENGE]\nallow_no_title = True"))
        assert cli.main(["--dir", "run", "--challenge"]) == 0           # plain DAE, titles_use = 0

        def check(res):
            assert len(res) == 13
            for row in res:
                assert isinstance(row[0], int) and 1 <= len(row) - 1 <= 500
                assert all(u.startswith("spotify:track:") for u in row[1:])
                assert len(set(row[1:])) == len(row) - 1
        res_plain = pickle.load(open(tmp_path / "challenge_results" / "result_inorder_5to100", "rb"))
        check(res_plain)
        assert "plain DAE" in open(work / "log.txt").read()
        # --title: the character CNN on top of the frozen w_dae, then --challenge mixes its scores in
        assert cli.main(["--dir", "run", "--title"]) == 0
        tv = pickle.load(open(work / "graph" / "model.ckpt.pkl", "rb"))
        assert tv["Output_W"].shape == (400, w2[0].shape[0]) and tv["char_embedding"].shape == (41, 50)
        assert np.array_equal(pickle.load(open(work / "w_dae", "rb"))[1], w2[1])          # the DAE stayed frozen
        log = open(work / "log.txt").read()
        tl = [float(l.split(":")[1]) for l in log.splitlines() if l.startswith("training loss")][4:]
        assert len(tl) == 2 and tl[1] < tl[0]
        assert cli.main(["--dir", "run", "--title", "--testmode"]) == 0
        assert cli.main(["--dir", "run", "--challenge"]) == 0
        res_mixed = pickle.load(open(tmp_path / "challenge_results" / "result_inorder_5to100", "rb"))
        check(res_mixed)
        assert "title scorer" in open(work / "log.txt").read()
        named = [i for i in range(13) if (900000 + i) % 4]                # synth_challenge gives these a name
        assert any(res_mixed[i] != res_plain[i] for i in named)
        assert all(res_mixed[i] == res_plain[i] for i in range(13) if i not in named)      # titles_use = 0 rows
        # [BASE] decode_dtype = exact_bf16 (north_star: the bf16 GEMM, the fp32 lists): the same result file, row for
        # row -- the plain rows through the bf16 filter + fp32 refine, the title-mixed launches on the fp32 kernels
        ini = open(work / "config.ini").read()
        open(work / "config.ini", "w").write(ini.replace("[BASE]", "[BASE]\ndecode_dtype = exact_bf16"))
        assert cli.main(["--dir", "run", "--challenge"]) == 0
        res_exact = pickle.load(open(tmp_path / "challenge_results" / "result_inorder_5to100", "rb"))
        assert res_exact == res_mixed
    finally:
        os.chdir(cwd)


def test_challenge_driver_with_titles_under_exact_bf16_at_hidden_256(tmp_path, monkeypatch):
    """The same drivers with [DAE] hidden = 256 (the shipped size): under `decode_dtype = exact_bf16` the title-mixed launches
    of `--challenge` take dae_mix_topk_exact (both GEMMs in one bf16 launch per pass) and the result file equals the fp32
    run's, row for row."""
    import random
    from spotify_recsys_challenge_2018_amd import main as cli
    from spotify_recsys_challenge_2018_amd import _lib as L
    work = tmp_path / "run"
    work.mkdir()
    ini = open(os.path.join(G, "config.ini")).read().replace("hidden = 32", "hidden = 256")
    ini = ini.replace("[CHALLENGE]", "[CHALLENGE]\nallow_no_title = True")
    open(work / "config.ini", "w").write(ini)
    shutil.copytree(os.path.join(G, "data"), tmp_path / "data")
    calls = []
    real, real1 = L.Context.mix_topk_exact, L.Context.title_score_exact        # (the latter: the whole launch in one call)
    monkeypatch.setattr(L.Context, "mix_topk_exact", lambda self, *a, **kw: (calls.append(1), real(self, *a, **kw))[1])
    monkeypatch.setattr(L.Context, "title_score_exact", lambda self, *a, **kw: (calls.append(1), real1(self, *a, **kw))[1])
    # ... or, since round 5, inside the library's titled pipeline (dae_pipeline_create_titled with DAE_DTYPE_BF16_EXACT,
    # feeds through dae_pipeline_submit_titled -> dae_title_score): recorded at the ctypes wrapper
    real_init, real_submit = L.Pipeline.__init__, L.Pipeline.submit

    def init(self, *a, **kw):
        real_init(self, *a, **kw)
        self._t_exact = kw.get("title") is not None and int(kw.get("dtype", 0)) == L.DAE_DTYPE_BF16_EXACT

    def submit(self, positions, values, n_rows, titles=None, titles_use=None):
        ok = real_submit(self, positions, values, n_rows, titles, titles_use)
        if ok and titles is not None and getattr(self, "_t_exact", False):
            calls.append(1)
        return ok
    monkeypatch.setattr(L.Pipeline, "__init__", init)
    monkeypatch.setattr(L.Pipeline, "submit", submit)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        random.seed(0); np.random.seed(0)
        for mode in ("--pretrain", "--dae", "--title"):
            assert cli.main(["--dir", "run", mode]) == 0
        assert cli.main(["--dir", "run", "--challenge"]) == 0
        res32 = pickle.load(open(tmp_path / "challenge_results" / "result_inorder_5to100", "rb"))
        assert not calls and len(res32) == 13
        open(work / "config.ini", "w").write(ini.replace("[BASE]", "[BASE]\ndecode_dtype = exact_bf16"))
        assert cli.main(["--dir", "run", "--challenge"]) == 0
        res_exact = pickle.load(open(tmp_path / "challenge_results" / "result_inorder_5to100", "rb"))
        assert calls, "the title-mixed launches did not take the exact path"
        assert res_exact == res32
    finally:
        os.chdir(cwd)


def _step(ctx, csr, d, V, H, B, tied, ikp, kp, seed, lam):
    import torch
    P = _lib._ptr
    out = dict(gWe=torch.zeros((V, H), device="cuda"), gbe=torch.zeros(H, device="cuda"),
               gWd=torch.zeros((V, H), device="cuda"), gbd=torch.zeros(V, device="cuda"),
               cost=torch.zeros(1, device="cuda"))
    ctx.check(ctx.lib.dae_train_forward_backward(
        ctx.h, P(csr[0]), P(csr[1]), P(csr[2]), P(csr[3]), P(csr[4]), P(csr[5]),
        P(d["We"]), P(d["be"]), P(d["Wd"]), P(d["bd"]), V, H, B, B, 1 if tied else 0,
        float(ikp), float(kp), seed, float(lam), P(out["gWe"]), P(out["gbe"]),
        None if tied else P(out["gWd"]), P(out["gbd"]), P(out["cost"])))
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("V,nt,H,B,tied", [(2100, 2000, 256, 250, False), (1500, 1200, 64, 64, True),
                                           (5000, 4000, 256, 130, False),
                                           # round 6, K5 through LDS: waves without a playlist, a full batch, more tiles
                                           # than workgroups
                                           (3000, 2500, 256, 37, False), (2100, 2000, 256, 256, False),
                                           (20000, 16000, 256, 200, False),
                                           # ... and vocabularies of one or two tiles (the fused K5 + K7 launch's clamps)
                                           (40, 30, 256, 5, False), (33, 20, 256, 256, False)])
def test_train_step_bf16_gemms(V, nt, H, B, tied):
    """dae_set_train_dtype(BF16) (BASELINE.json configs[3]): the three GEMMs of the step (forward, gW_dec, dh) run on
    bf16 operands with fp32 accumulate (hidden = 256 / 128: the 4-tile kernels; other sizes keep fp32 backward
    GEMMs); loss, dL/dz, parameters and Adam stay fp32.  Stated tolerance against the fp32 step on the same draws:
    cost within 3e-3 relative, every gradient within 2e-2 of its Frobenius norm (bf16 keeps 8 significant bits;
    the error of an operand is ~2^-9 relative, averaged over the sums)."""
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt, tied=tied)
    b_enc = (np.random.default_rng(2).standard_normal(H) * 0.1).astype(np.float32)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6, seed_counts=(3, 9, 20))
    xr, xc, xv = coo_to_csr(pos[pos[:, 1] < nt], ones[pos[:, 1] < nt], B, V)
    yr, yc, yv = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)
    csr = [_dev(a) for a in (xr, xc, xv, yr, yc, yv)]
    d = dict(We=_dev(W_enc), be=_dev(b_enc), Wd=_dev(W_dec), bd=_dev(b_dec))
    ctx = _lib.Context(0)
    f32 = _step(ctx, csr, d, V, H, B, tied, 0.75, 0.8, 31337, 0.0)
    ctx.set_train_dtype(_lib.DAE_DTYPE_BF16)
    b16 = _step(ctx, csr, d, V, H, B, tied, 0.75, 0.8, 31337, 0.0)
    again = _step(ctx, csr, d, V, H, B, tied, 0.75, 0.8, 31337, 0.0)
    ctx.set_train_dtype(_lib.DAE_DTYPE_F32)
    back = _step(ctx, csr, d, V, H, B, tied, 0.75, 0.8, 31337, 0.0)
    assert b16["cost"][0] != f32["cost"][0]                       # the bf16 path really ran
    assert abs(b16["cost"][0] - f32["cost"][0]) <= 3e-3 * abs(f32["cost"][0])
    for k in ("gWe", "gbe", "gbd") + (() if tied else ("gWd",)):
        err = np.linalg.norm(b16[k].astype(np.float64) - f32[k]) / np.linalg.norm(f32[k].astype(np.float64))
        assert err <= 2e-2, (k, err)
        assert np.allclose(again[k], b16[k], rtol=2e-4, atol=2e-7)
        assert np.allclose(back[k], f32[k], rtol=2e-4, atol=2e-7)
    with pytest.raises(_lib.DaeError):
        ctx.set_train_dtype(7)
    ctx.close()


def test_rows_adam_is_bit_identical_to_dense_adam():
    """dae_adam_rows_begin / _apply / _flush on a row-sparse gradient walk every element through exactly the
    updates dae_adam_step applies every step: parameters and both moments bit-equal, the dense gradient buffer
    left all-zero, whatever the pattern of touched rows (duplicates in the list, rows never touched, empty steps)."""
    import torch
    ctx = _lib.Context(0)
    P = _lib._ptr
    rng = np.random.default_rng(11)
    n_rows, row_len, steps, cap = 300, 96, 40, 64
    p0 = rng.standard_normal((n_rows, row_len)).astype(np.float32)
    dense = [_dev(p0), _dev(np.zeros_like(p0)), _dev(np.zeros_like(p0))]
    lazy = [_dev(p0), _dev(np.zeros_like(p0)), _dev(np.zeros_like(p0))]
    state = torch.zeros(2 * n_rows, dtype=torch.int32, device="cuda")
    tab = torch.zeros(cap, dtype=torch.float32, device="cuda")
    g_lazy = torch.zeros((n_rows, row_len), device="cuda")
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 0.005
    for t in range(1, steps + 1):
        k = 0 if t % 7 == 0 else int(rng.integers(1, 40))
        rows = rng.choice(n_rows // 2, size=k, replace=False) if k else np.zeros(0, np.int64)   # upper half: never touched
        listed = np.concatenate([rows, rows[: k // 2]]).astype(np.int32)                       # duplicates
        g = np.zeros((n_rows, row_len), np.float32)
        g[rows] = rng.standard_normal((k, row_len)).astype(np.float32)
        if k:
            g[rows[0], ::3] = 0.0                                                              # zeros inside a listed row
        d_list = _dev(listed if len(listed) else np.zeros(1, np.int32))
        n_dev = _dev(np.array([len(listed)], np.int32))
        ctx.check(ctx.lib.dae_adam_step(ctx.h, P(dense[0]), P(dense[1]), P(dense[2]), P(_dev(g)), n_rows * row_len,
                                        lr, b1, b2, eps, t))
        ctx.check(ctx.lib.dae_adam_rows_begin(ctx.h, P(lazy[0]), P(lazy[1]), P(lazy[2]), P(state), P(tab), cap,
                                              n_rows, row_len, P(d_list), P(n_dev), len(listed) + 5, b1, b2, eps, t))
        g_lazy[torch.from_numpy(rows).cuda().long()] = torch.from_numpy(g[rows]).cuda()
        ctx.check(ctx.lib.dae_adam_rows_apply(ctx.h, P(lazy[0]), P(lazy[1]), P(lazy[2]), P(g_lazy), P(state), P(tab),
                                              cap, n_rows, row_len, P(d_list), P(n_dev), len(listed) + 5,
                                              lr, b1, b2, eps, t))
        assert not g_lazy.any()                                     # re-zeroed by apply
        if t in (13, steps):                                        # a sync point in the middle, and the end
            ctx.check(ctx.lib.dae_adam_rows_flush(ctx.h, P(lazy[0]), P(lazy[1]), P(lazy[2]), P(state), P(tab), cap,
                                                  n_rows, row_len, b1, b2, eps, t))
            for a, b in zip(dense, lazy):
                assert torch.equal(a, b)
    with pytest.raises(_lib.DaeError):                               # the alpha table is full
        ctx.lib.dae_adam_rows_begin.restype = ctypes.c_int
        ctx.check(ctx.lib.dae_adam_rows_begin(ctx.h, P(lazy[0]), P(lazy[1]), P(lazy[2]), P(state), P(tab), cap,
                                              n_rows, row_len, P(state), None, 1, b1, b2, eps, cap))
    ctx.close()


def test_model_rows_adam_follows_dense_adam():
    """The untied model's default (encoder through dae_adam_rows_*) against encoder_adam = "dense": same costs and,
    after the flush that get_params() triggers, the same parameters -- to the rounding of the float atomics in the
    sparse encoder gradient, which differ between any two runs."""
    from spotify_recsys_challenge_2018_amd.models.DAEs import DAE

    class C:
        save = "/tmp/_ra_unused"; batch = 32; n_input = 1500; hidden = 64; lr = 0.01; reg_lambda = 0.0
        initval = "NULL"; n_tracks = 1200
    rng = np.random.default_rng(5)
    batches = []
    for s in range(12):
        pos, ones, _ = make_playlists(C.batch, 1200, 300, seed=100 + s, seed_counts=(3, 9, 20))
        batches.append((pos[pos[:, 1] < 1200], ones[pos[:, 1] < 1200], pos, np.ones(len(pos), np.float32)))
    ca_ = C(); ca_.rows_adam_table = 16            # the per-step alpha table has to grow twice in 36 steps
    a = DAE(ca_); a.fit()
    cd = C(); cd.encoder_adam = "dense"
    b = DAE(cd); b.fit()
    assert a.encoder_adam == "rows" and b.encoder_adam == "dense"
    ca, cb = [], []
    for i in range(36):
        x, xv, y, yv = batches[i % len(batches)]
        ca.append(a.train_step(x, xv, y, yv, 0.8, 0.75))
        cb.append(b.train_step(x, xv, y, yv, 0.8, 0.75))
        if i == 17:                                   # a scoring call in the middle: flush, then keep training
            ia, _ = a.recommend(x, xv, [[] for _ in range(C.batch)], k=50)
            ib, _ = b.recommend(x, xv, [[] for _ in range(C.batch)], k=50)
            assert (ia == ib).mean() > 0.99
    assert a._lazy is not None and b._lazy is None and a._lazy["tab"].numel() >= 64
    assert np.allclose(ca, cb, rtol=2e-4)
    for pa, pb in zip(a.get_params(), b.get_params()):
        assert np.allclose(pa, pb, rtol=1e-3, atol=2e-6)


@pytest.mark.parametrize("bf16", [False, True])
def test_armed_decoder_adam_is_bit_identical(bf16):
    """dae_arm_decoder_adam: the step applies the dense Adam update of W_dec inside the decoder-gradient kernel.
    Same inputs, same draws: W_dec / m / v must be the bits of `write gW_dec, then dae_adam_step`, and every other
    output of the step (cost, gW_enc, gb_enc, gb_dec) must be unchanged -- K7 still multiplies by the old W_dec."""
    import torch
    V, nt, H, B = 3000, 2400, 128, 100
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt)
    pos, ones, _ = make_playlists(B, nt, V - nt, seed=6, seed_counts=(3, 9, 20))
    xr, xc, xv = coo_to_csr(pos[pos[:, 1] < nt], ones[pos[:, 1] < nt], B, V)
    yr, yc, yv = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)
    csr = [_dev(a) for a in (xr, xc, xv, yr, yc, yv)]
    ctx = _lib.Context(0)
    if bf16:
        ctx.set_train_dtype(_lib.DAE_DTYPE_BF16)
    P = _lib._ptr
    rng = np.random.default_rng(1)
    m0 = (rng.standard_normal((V, H)) * 1e-3).astype(np.float32)
    v0 = (rng.random((V, H)) * 1e-6).astype(np.float32)
    lr, t_step = 0.005, 7

    def run(armed):
        d = dict(We=_dev(W_enc), be=_dev(b_enc), Wd=_dev(W_dec), bd=_dev(b_dec))
        m, v = _dev(m0), _dev(v0)
        out = dict(gWe=torch.zeros((V, H), device="cuda"), gbe=torch.zeros(H, device="cuda"),
                   gWd=torch.zeros((V, H), device="cuda"), gbd=torch.zeros(V, device="cuda"),
                   cost=torch.zeros(1, device="cuda"))
        if armed:
            ctx.check(ctx.lib.dae_arm_decoder_adam(ctx.h, P(m), P(v), lr, 0.9, 0.999, 1e-8, t_step))
        ctx.check(ctx.lib.dae_train_forward_backward(
            ctx.h, P(csr[0]), P(csr[1]), P(csr[2]), P(csr[3]), P(csr[4]), P(csr[5]),
            P(d["We"]), P(d["be"]), P(d["Wd"]), P(d["bd"]), V, H, B, B, 0, 0.75, 0.8, 99, 0.0,
            P(out["gWe"]), P(out["gbe"]), None if armed else P(out["gWd"]), P(out["gbd"]), P(out["cost"])))
        if not armed:
            ctx.check(ctx.lib.dae_adam_step(ctx.h, P(d["Wd"]), P(m), P(v), P(out["gWd"]), V * H, lr, 0.9, 0.999, 1e-8, t_step))
        torch.cuda.synchronize()
        return d["Wd"], m, v, out

    Wa, ma, va, oa = run(True)
    Wb, mb, vb, ob = run(False)
    assert torch.equal(Wa, Wb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert not torch.equal(Wa, _dev(W_dec))
    assert torch.equal(oa["cost"], ob["cost"]) and torch.equal(oa["gbe"], ob["gbe"]) and torch.equal(oa["gbd"], ob["gbd"])
    assert np.allclose(oa["gWe"].cpu().numpy(), ob["gWe"].cpu().numpy(), rtol=2e-4, atol=2e-7)   # float atomics
    # one step only: the next call needs gW_dec again
    with pytest.raises(_lib.DaeError):
        ctx.check(ctx.lib.dae_train_forward_backward(
            ctx.h, P(csr[0]), P(csr[1]), P(csr[2]), P(csr[3]), P(csr[4]), P(csr[5]),
            P(Wa), P(_dev(b_enc)), P(Wa), P(_dev(b_dec)), V, H, B, B, 0, 0.75, 0.8, 99, 0.0,
            P(oa["gWe"]), P(oa["gbe"]), None, P(oa["gbd"]), P(oa["cost"])))
    ctx.close()


def test_dae_driver_with_bf16_keys_in_config(tmp_path):
    """main.py --pretrain / --dae with `[BASE] train_dtype = bf16, decode_dtype = bf16` in config.ini (the two keys this
    build adds): trains, evaluates and saves through the bf16 GEMMs; the loss still falls."""
    import configparser
    import random
    from spotify_recsys_challenge_2018_amd import main as cli
    work = tmp_path / "run"
    work.mkdir()
    ini = configparser.ConfigParser()
    ini.read(os.path.join(G, "config.ini"))
    ini["BASE"]["train_dtype"] = "bf16"
    ini["BASE"]["decode_dtype"] = "bf16"
    with open(work / "config.ini", "w") as f:
        ini.write(f)
    shutil.copytree(os.path.join(G, "data"), tmp_path / "data")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        random.seed(0); np.random.seed(0)
        assert cli.main(["--dir", "run", "--pretrain"]) == 0
        assert cli.main(["--dir", "run", "--dae"]) == 0
        w2 = pickle.load(open(work / "w_dae", "rb"))
        assert w2[0].dtype == np.float32 and np.isfinite(w2[0]).all() and np.isfinite(w2[1]).all()
        log = open(work / "log.txt").read()
        losses = [float(l.split(":")[1]) for l in log.splitlines() if l.startswith("training loss")]
        assert len(losses) == 4 and losses[1] < losses[0] and losses[3] < losses[2]
        assert "rprecision" in log
    finally:
        os.chdir(cwd)
