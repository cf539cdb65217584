"""GPU (-m gpu): the bf16 decode path (BASELINE.json configs[4]: bf16 MFMA decode + fp32 encode and
top-k).  The MFMA sums 16 exact bf16 products per instruction in an unspecified order, so parity
with the bf16 oracle (operands rounded to bf16, fp32 accumulation) is by tolerance: |dz| <= 3e-5
absolute on logits of magnitude <= 10.  Rankings are compared with the fp32 path through
r-precision (tolerance 0.02) as configs[4] asks."""
import os
import pickle
import shutil

import numpy as np
import pytest

import oracle
from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE, coo_to_csr, seeds_to_csr
from spotify_recsys_challenge_2018_amd.utils import metrics as met
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF = _lib.DAE_DTYPE_BF16


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("V,H,B", [(3000, 256, 300), (2000, 32, 8), (5000, 128, 130), (1111, 96, 70), (4096, 256, 64)])
def test_bf16_decode_dense_vs_bf16_oracle(ctx, V, H, B):
    import torch
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=3, bias="zipf")
    h = np.random.default_rng(0).random((B, H)).astype(np.float32)
    dW, db, dh = _dev(W_dec), _dev(b_dec), _dev(h)
    ctx.prepack_decoder(dW, db, dtype=BF)
    out = torch.empty((B, V), dtype=torch.float32, device="cuda")
    ctx.decode_dense(dh, out, apply_sigmoid=False, dtype=BF)
    z_ref = oracle.decode(h, W_dec, b_dec, bf16=True)
    z = out.cpu().numpy()
    assert np.max(np.abs(z - z_ref)) <= 3e-5
    z32 = oracle.decode(h, W_dec, b_dec)
    assert np.max(np.abs(z - z32)) < 0.05 and np.max(np.abs(z - z32)) > 1e-6      # it really is bf16


def test_bf16_fused_topk_equals_unfused_and_tracks_fp32(ctx):
    import torch
    V, nt, H, B, k = 60000, 50000, 256, 300, 500
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=1, bias="zipf", n_tracks=nt)
    pos, ones, seeds = make_playlists(B, nt, V - nt, seed=2)
    rp, col, val = coo_to_csr(pos, ones, B, V)
    srp, sc = seeds_to_csr(seeds, B, nt)
    d = [_dev(a) for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
    ctx.prepack_decoder(d[5], d[6], dtype=BF)
    ctx.prepack_decoder(d[5], d[6], dtype=_lib.DAE_DTYPE_F32)
    s16 = torch.empty((B, k), device="cuda"); i16 = torch.empty((B, k), dtype=torch.int32, device="cuda")
    s32 = torch.empty_like(s16); i32 = torch.empty_like(i16)
    ctx.score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, s16, i16, dtype=BF)
    assert ctx.last_plan()["fused"] == 1 and ctx.last_plan()["R_TILE"] == 128
    ctx.score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, s32, i32)
    # unfused bf16: same logits, same ranking rule -> identical
    h = torch.empty((B, H), device="cuda")
    ctx.encode(d[0], d[1], d[2], d[3], d[4], h)
    z = torch.empty((B, V), device="cuda")
    ctx.decode_dense(h, z, apply_sigmoid=False, dtype=BF)
    s_u = torch.empty_like(s16); i_u = torch.empty_like(i16)
    ctx.topk_dense(z, nt, 0, d[7], d[8], k, s_u, i_u)
    assert torch.equal(i16, i_u) and torch.equal(s16, s_u)
    # against fp32: r-precision of the bf16 list w.r.t. the fp32 top-R as "answers"
    a16, a32 = i16.cpu().numpy(), i32.cpu().numpy()
    for R in (10, 100, 500):
        rp_ = np.mean([met.get_r_precision(a32[r, :R].tolist(), a16[r].tolist()) for r in range(B)])
        assert rp_ >= 0.97, (R, rp_)
    assert np.max(np.abs(s16.cpu().numpy()[:, 0] - s32.cpu().numpy()[:, 0])) < 5e-3


SPLITS = ("test-0", "test-1", "test-5", "test-25r")        # readme.md:69 seed patterns BASELINE.json configs[4] names
RPREC_TOL = 0.02                                             # stated tolerance of configs[4]


def test_bf16_vs_fp32_r_precision_on_trained_model_per_split(tmp_path, capsys):
    """configs[4]: r-precision@500 of the bf16 decode within 0.02 of fp32 on EACH of the seed-0/1/5/25r test splits
    (golden mini dataset: splits written by the repaired generator from a held-out slice; 3 epochs of --pretrain on
    the GPU).  The 0-seed split feeds all-zero rows: h = sigmoid(b_enc), a pure popularity ranking."""
    import json
    import random
    from spotify_recsys_challenge_2018_amd import main as cli
    from spotify_recsys_challenge_2018_amd.main_runner import main_train
    from spotify_recsys_challenge_2018_amd.utils.data_reader import data_reader_test
    work = tmp_path / "run"; work.mkdir()
    cfg = open(os.path.join(G, "config.ini")).read().replace("epochs = 2", "epochs = 3")
    open(work / "config.ini", "w").write(cfg)
    shutil.copytree(os.path.join(G, "data"), tmp_path / "data")
    cwd = os.getcwd(); os.chdir(tmp_path)
    try:
        random.seed(1); np.random.seed(1)
        assert cli.main(["--dir", "run", "--pretrain"]) == 0
        conf = cli.load_conf("./run"); conf.set_dae_conf(); conf.initval = conf.save = str(work / "w_pretrain")
        tr = json.load(open("./data/train"))
        conf.n_tracks = len(tr["track_uri2id"]); conf.n_input = conf.n_tracks + len(tr["artist_uri2id"])
        models = {}
        for name in ("f32", "bf16"):
            conf.decode_dtype = name
            models[name] = DAE(conf); models[name].fit()
        res = {}
        for split in SPLITS:
            rd = data_reader_test("./data", split, conf.batch, 1000)
            assert len(rd.playlists) >= 8
            if split == "test-0":
                assert all(len(p[0]) == 0 for p in rd.playlists)
            res[split] = {name: main_train.eval(rd, conf, m) for name, m in models.items()}
        assert res["test-5"]["f32"] > 0.05 and res["test-25r"]["f32"] > 0.05          # the model learnt something
        assert res["test-0"]["f32"] > 0.0                                              # popularity alone hits
        for split in SPLITS:
            assert abs(res[split]["f32"] - res[split]["bf16"]) <= RPREC_TOL, (split, res)
    finally:
        os.chdir(cwd)


def test_bf16_vs_fp32_r_precision_full_vocabulary_per_seed_pattern(ctx):
    """The same tolerance at BASELINE's full size (|vocab| = 170 000, hidden 256, 256 rows per split) on SYNTHETIC
    splits of the four seed patterns -- real splits need the MPD, which is licence-gated.  Inputs are what
    data_reader_test feeds: the seed TRACKS with weight 1 (none for seed-0: every row is the popularity ranking).
    Answers per row: 40 tracks drawn from the fp32 path's top-500 plus 20 tracks it does not rank (misses), so that
    fp32 r-precision sits where a trained model's does; bf16 must stay within 0.02 on every split."""
    import torch
    V, nt, H, B, k = 170000, 140000, 256, 256, 500
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    d_We, d_be, d_Wd, d_bd = _dev(W_enc), _dev(b_enc), _dev(W_dec), _dev(b_dec)
    ctx.prepack_decoder(d_Wd, d_bd, dtype=BF)
    ctx.prepack_decoder(d_Wd, d_bd, dtype=_lib.DAE_DTYPE_F32)
    rng = np.random.default_rng(2018)
    out = {}
    for n_seed, label in ((0, "seed-0"), (1, "seed-1"), (5, "seed-5"), (25, "seed-25r")):
        seeds = []
        for r in range(B):
            ids = np.minimum(nt - 1, np.floor(np.exp(rng.random(n_seed * 2) * np.log(nt))).astype(np.int64) - 1).clip(0)
            ids = list(dict.fromkeys(ids.tolist()))[:n_seed]
            if label.endswith("r"):
                rng.shuffle(ids)
            seeds.append(ids)
        rows = np.repeat(np.arange(B), [len(s_) for s_ in seeds])
        pos = np.stack([rows, np.array([t for s_ in seeds for t in s_], np.int64)], 1) if n_seed else np.zeros((0, 2), np.int64)
        rp, col, val = coo_to_csr(pos, np.ones(len(pos), np.float32), B, V)
        srp, sc = seeds_to_csr(seeds, B, nt)
        d = [_dev(a) for a in (rp, col if col.size else np.zeros(1, np.int32), val if val.size else np.zeros(1, np.float32),
                               srp, sc if sc.size else np.zeros(1, np.int32))]
        lists = {}
        for name, dt in (("f32", _lib.DAE_DTYPE_F32), ("bf16", BF)):
            s_ = torch.empty((B, k), device="cuda"); i_ = torch.empty((B, k), dtype=torch.int32, device="cuda")
            ctx.score_topk(d[0], d[1], d[2], d_We, d_be, nt, d[3], d[4], k, s_, i_, dtype=dt)
            assert ctx.last_plan()["fused"] == 1
            lists[name] = i_.cpu().numpy()
        if n_seed == 0:
            assert np.all(lists["f32"] == lists["f32"][0])                    # identical rows: pure popularity
        rp_ = {"f32": 0.0, "bf16": 0.0}
        for r in range(B):
            top = lists["f32"][r]
            hits = rng.choice(top, size=40, replace=False).tolist()
            misses = [int(t) for t in rng.integers(0, nt, 40) if t not in set(top.tolist()) and t not in seeds[r]][:20]
            answers = hits + misses
            rng.shuffle(answers)
            for name in rp_:
                rp_[name] += met.eval_topk(lists[name][r], answers) / B
        out[label] = rp_
        assert 0.02 < rp_["f32"] < 0.6, (label, rp_)
        assert abs(rp_["f32"] - rp_["bf16"]) <= RPREC_TOL, (label, rp_)
    print("full-size r-precision fp32 vs bf16 per seed pattern:", out)


@pytest.mark.parametrize("V,nt,B,k", [(40000, 33000, 1, 500), (40037, 39990, 37, 500), (52000, 52000, 129, 100),
                                      (33000, 20011, 257, 500), (70000, 64000, 1024, 500), (36000, 30000, 128, 1)])
def test_bf16_fused_equals_unfused_across_shapes(ctx, V, nt, B, k):
    """The dedicated bf16 phase-B kernel (tile pairs, ragged last pair, bias through the matrix pipe)
    against the dense bf16 path + the same ranking: identical indices and scores for every batch shape."""
    import torch
    H = 256
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=5, bias="zipf", n_tracks=nt)
    pos, ones, seeds = make_playlists(B, nt, V - nt, seed=6)
    rp, col, val = coo_to_csr(pos, ones, B, V)
    srp, sc = seeds_to_csr(seeds, B, nt)
    d = [_dev(a) for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc if sc.size else np.zeros(1, np.int32))]
    ctx.prepack_decoder(d[5], d[6], dtype=BF)
    s16 = torch.empty((B, k), device="cuda"); i16 = torch.empty((B, k), dtype=torch.int32, device="cuda")
    ctx.score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, s16, i16, dtype=BF)
    assert ctx.last_plan()["fused"] == 1
    h = torch.empty((B, H), device="cuda")
    ctx.encode(d[0], d[1], d[2], d[3], d[4], h)
    z = torch.empty((B, V), device="cuda")
    ctx.decode_dense(h, z, apply_sigmoid=False, dtype=BF)
    s_u = torch.empty_like(s16); i_u = torch.empty_like(i16)
    ctx.topk_dense(z, nt, 0, d[7], d[8], k, s_u, i_u)
    assert torch.equal(i16, i_u) and torch.equal(s16, s_u)
    z_ref = oracle.decode(h.cpu().numpy(), W_dec, b_dec, bf16=True)
    assert np.max(np.abs(z.cpu().numpy() - z_ref)) <= 3e-5


@pytest.mark.parametrize("B", [1024, 2048])
def test_bf16_fused_equals_unfused_at_the_loops_launch_sizes(B):
    """Full vocabulary, the launches the drivers' loop issues (1 024 / 2 048 rows): phase A takes per-WAVE group maxima there
    (decode_bf16_h256_wavemax_kernel; waves paired at 1 024 rows).  A threshold that was too high would lose winners: the fused
    lists must equal the dense bf16 logits ranked by the same rule, indices and scores."""
    import torch
    c = _lib.Context(0)
    try:
        V, nt, H, k = 170000, 140000, 256, 500
        W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
        pos, ones, seeds = make_playlists(B, nt, V - nt, seed=21)
        rp, col, val = coo_to_csr(pos, ones, B, V)
        srp, sc = seeds_to_csr(seeds, B, nt)
        d = [_dev(a) for a in (rp, col, val, W_enc, b_enc, W_dec, b_dec, srp, sc)]
        c.prepack_decoder(d[5], d[6], dtype=BF)
        s16 = torch.empty((B, k), device="cuda"); i16 = torch.empty((B, k), dtype=torch.int32, device="cuda")
        c.score_topk(d[0], d[1], d[2], d[3], d[4], nt, d[7], d[8], k, s16, i16, dtype=BF)
        assert c.last_plan()["fused"] == 1
        h = torch.empty((B, H), device="cuda")
        c.encode(d[0], d[1], d[2], d[3], d[4], h)
        z = torch.empty((B, V), device="cuda")
        c.decode_dense(h, z, apply_sigmoid=False, dtype=BF)
        s_u = torch.empty_like(s16); i_u = torch.empty_like(i16)
        c.topk_dense(z, nt, 0, d[7], d[8], k, s_u, i_u)
        assert torch.equal(i16, i_u) and torch.equal(s16, s_u)
    finally:
        c.close()
