"""GPU (-m gpu): the drivers' streamed loop (`DAE.recommend_iter`: main_challenge.py:72-93 / main_train.py:62-96 with the
host and the device overlapped -- the library's dae_pipeline_*, three contexts taking the launches in turn) returns, batch
for batch, what `recommend` returns for the same feed -- fp32 bit for bit, bf16 likewise (same kernels, same order of
operations).  (Round 6 retired the interpreter loop these tests used to run beside it.)"""
import pickle

import numpy as np
import pytest

from spotify_recsys_challenge_2018_amd import _lib
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE, SEEDS_FROM_INPUT
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,B", [("f32", 256), ("bf16", 256), ("f32", 150), ("f32", 250), ("exact_bf16", 256), ("exact_bf16", 150)])
def test_recommend_iter_equals_recommend(tmp_path, dtype, B):
    """7 feeds (short ones among them) through the streamed loop: coalesced 4 or 5 to a launch (256 / 250 -> 4, the
    reference's challenge batch of 150 -> 5; 8 in the bf16 modes), launches dealt to the pipeline's lanes in turn."""
    nt, na, H, k = 20000, 4000, 256, 500
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=2, bias="zipf", n_tracks=nt)
    path = str(tmp_path / "init.pkl")
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = str(tmp_path / "unused"); batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    batches = [make_playlists(B, nt, na, seed=10 + s) for s in range(7)]           # odd count: the lanes end unevenly
    rows = [B, B, 100, B, 1, B, 37]                                                # short last batches of a file
    assert m._coalesce_count() == {256: 4, 250: 4, 150: 5}[B]
    want = [m.recommend(p, o, s, k=k, n_rows=n, dtype=dtype) for (p, o, s), n in zip(batches, rows)]
    if dtype == "exact_bf16":                     # north_star: the fp32 path's lists, indices and scores, bit for bit
        for (p, o, s), n, (wi, ws) in zip(batches, rows, want):
            fi, fs = m.recommend(p, o, s, k=k, n_rows=n, dtype="f32")
            assert np.array_equal(fi, wi) and np.array_equal(fs.view(np.uint32), ws.view(np.uint32))
    feeds = [(p, o, SEEDS_FROM_INPUT, n) for (p, o, _s), n in zip(batches, rows)]
    got = list(m.recommend_iter(feeds, k=k, dtype=dtype))
    assert len(got) == len(want)
    for (gi, gs), (wi, ws) in zip(got, want):
        assert np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
    # the pipeline follows a weight change (its packed image is rebuilt), and other lane counts agree
    m.weights["decoder_h"].mul_(-1.0); m._mark_dirty()
    want2 = m.recommend(*batches[0], k=k, dtype=dtype)
    got2 = list(m.recommend_iter(feeds[:3], k=k, dtype=dtype))
    assert np.array_equal(got2[0][0], want2[0]) and not np.array_equal(got2[0][0], want[0][0])
    for nl in (1, 2):
        m.n_lanes = nl
        m.__dict__.pop("_pipes", None)
        got1 = list(m.recommend_iter(feeds[:3], k=k, dtype=dtype))
        assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got1, got2))
    m.n_lanes = None
    # explicit seed lists (feeds the pipeline does not take go through `recommend`, in order) mixed with the drivers' kind
    mixed = [(p, o, (s if i % 2 else SEEDS_FROM_INPUT), n) for i, ((p, o, s), n) in enumerate(zip(batches, rows))]
    got3 = list(m.recommend_iter(mixed, k=k, dtype=dtype))
    want3 = [m.recommend(p, o, s, k=k, n_rows=n, dtype=dtype) for (p, o, s), n in zip(batches, rows)]
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(got3, want3))
    # with and without the scores, twice over the launch slots; a consumer that stops early leaves the model usable
    for ws_ in (True, False):
        got4 = list(m.recommend_iter(mixed * 2, k=k, dtype=dtype, want_scores=ws_))
        assert len(got4) == 2 * len(want3)
        for a, b in zip(got4, want3 * 2):
            assert np.array_equal(a[0], b[0]) and (a[1] is None if not ws_ else np.array_equal(a[1], b[1]))
    it = m.recommend_iter(mixed * 2, k=k, dtype=dtype)
    first = next(it); it.close()
    assert np.array_equal(first[0], want3[0][0])
    got5 = list(m.recommend_iter(feeds[:2], k=k, dtype=dtype))
    assert np.array_equal(got5[1][0], want2[0]) is False and len(got5) == 2
    bad = [(np.array([[0, 1], [1, 2]], np.int64), np.ones(3, np.float32), SEEDS_FROM_INPUT, 2)]
    with pytest.raises(ValueError):
        list(m.recommend_iter(bad * 4, k=k, dtype=dtype))
    # a model that builds its CSRs on the host (device_csr = False) is served feed by feed: the same lists
    m.device_csr = False
    got6 = list(m.recommend_iter(feeds[:2], k=k, dtype=dtype))
    m.device_csr = True
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(got6, got5))


def test_native_pipeline_orders_lends_and_recovers(tmp_path):
    """dae_pipeline_* directly (include/dae_hip.h): feeds of different sizes come back in submission order with the lists
    dae_score_topk gives each alone; results are views of pinned blocks that go back when the arrays die and are copied when
    the caller hoards them; a feed with an index out of range is an error and not a silent skip; the exact mode's bound guard
    (forged margin) makes the pipeline re-score the launch in fp32 -- the lists stay the fp32 lists."""
    import gc
    import torch
    nt, na, H, k = 9000, 1500, 128, 200
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=6, bias="zipf", n_tracks=nt)
    W_dec = (W_dec * 40).astype(np.float32)
    dev = [torch.from_numpy(a).cuda() for a in (W_enc, b_enc, W_dec, b_dec)]
    path = str(tmp_path / "init.pkl")
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = str(tmp_path / "unused"); batch = 64; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    sizes = [64, 17, 64, 1, 40, 64, 64, 5, 64, 33, 64, 64]
    batches = [make_playlists(n, nt, na, seed=50 + i) for i, n in enumerate(sizes)]
    want = [m.recommend(p, o, SEEDS_FROM_INPUT, k=k, n_rows=n, dtype="f32") for (p, o, _s), n in zip(batches, sizes)]
    for dtype in (_lib.DAE_DTYPE_F32, _lib.DAE_DTYPE_BF16_EXACT):
        pipe = _lib.Pipeline(*dev, nt, dtype=dtype, k=k, group_rows=160, max_nnz=1 << 16, lanes=2, want_scores=True, result_blocks=7)
        got = []
        for (p, o, _s), n in zip(batches, sizes):
            while not pipe.submit(p, o, n):
                got.append(pipe.poll(True))
        pipe.flush()
        while pipe.pending:
            got.append(pipe.poll(True))
        assert len(got) == len(want) and pipe.poll(True) is None
        for (gi, gs), (wi, ws) in zip(got, want):
            assert gi.shape == wi.shape and np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        st = pipe.stats()
        assert st["feeds"] == len(sizes) and 0 < st["launches"] < len(sizes) and st["guard_fallbacks"] == 0
        assert len(pipe._held) <= pipe.max_held          # hoarded results were copied, not lent
        del got, gi, gs
        gc.collect()
        assert not pipe._held
        if dtype == _lib.DAE_DTYPE_BF16_EXACT:
            pipe.exact_margin(1e-3)                       # forged bound: the guard fires, the launch is re-scored in fp32
            for (p, o, _s), n in zip(batches[:4], sizes[:4]):
                assert pipe.submit(p, o, n)
            pipe.flush()
            again = [pipe.poll(True, copy=True) for _ in range(4)]
            assert pipe.stats()["guard_fallbacks"] >= 1
            for (gi, gs), (wi, ws) in zip(again, want):
                assert np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        # a column out of range: the launch reports it
        bad = np.array([[0, 1], [1, V + 5]], np.int64)
        assert pipe.submit(bad, np.ones(2, np.float32), 2)
        with pytest.raises(_lib.DaeError):
            pipe.poll(True)
        pipe.close()
        # a row outside the feed is refused at submit, and the pipeline lives on
        pipe = _lib.Pipeline(*dev, nt, dtype=dtype, k=k, group_rows=160, max_nnz=1 << 16, lanes=2)
        with pytest.raises(_lib.DaeError):
            pipe.submit(np.array([[7, 1]], np.int64), np.ones(1, np.float32), 2)
        p0, o0, _ = batches[0]
        assert pipe.submit(p0, o0, 64)
        gi, gs = pipe.poll(True)
        assert np.array_equal(gi, want[0][0])
        pipe.close()


def test_streamed_loop_re_scores_when_the_guard_fires(tmp_path):
    """A forged bound (dae_set_exact_margin < 1 on the model's context, handed on to the pipeline's own images) makes the
    streamed loop warn and return the fp32 kernels' lists (dae_pipeline_poll re-scores the launch) -- and an honest bound stays
    silent, with the counters of earlier hits NOT charged to later launches."""
    import warnings
    nt, na, H, k, B = 20000, 3000, 128, 300, 64
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=9, bias="zipf", n_tracks=nt)
    W_dec = (W_dec * 40).astype(np.float32)
    path = str(tmp_path / "init.pkl")
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = str(tmp_path / "unused"); batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    batches = [make_playlists(B, nt, na, seed=70 + i)[:2] for i in range(6)]
    feeds = [(p, o, SEEDS_FROM_INPUT, B) for p, o in batches]
    want = [m.recommend(p, o, SEEDS_FROM_INPUT, k=k, dtype="f32") for p, o in batches]

    def same(got):
        assert len(got) == len(want)
        for (gi, gs), (wi, ws) in zip(got, want):
            assert np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        same(list(m.recommend_iter(feeds, k=k, dtype="exact_bf16")))
    assert m.__dict__.get("_guard_fallbacks", 0) == 0
    m.ctx.set_exact_margin(1e-3)
    m._mark_dirty()
    with pytest.warns(UserWarning, match="bound guard"):
        same(list(m.recommend_iter(feeds, k=k, dtype="exact_bf16")))
    assert m._guard_fallbacks >= 1
    m.ctx.set_exact_margin(1.0)
    m._mark_dirty()
    n0 = m._guard_fallbacks
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        same(list(m.recommend_iter(feeds, k=k, dtype="exact_bf16")))
    assert m._guard_fallbacks == n0


def test_clock_probe_reads_a_plausible_engine_clock():
    """dae_clock_probe (bench.py's `roofline.sustained_clock`): shader cycles over wall-clock ticks of one wave."""
    import torch
    ctx = _lib.Context(0)
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    khz = ctx.clock_probe(out, window_us=200)
    torch.cuda.synchronize()
    cyc, ticks = (int(v) for v in out.cpu())
    assert khz > 0 and ticks >= 200 * khz // 1000 and cyc > 0
    ghz = cyc / ticks * khz / 1e6
    assert 0.1 < ghz < 3.0, ghz
    ctx.close()


def test_native_pipeline_feed_semantics_duplicates_long_rows_and_bad_indices(tmp_path):
    """Round 6: dae_pipeline stages a feed as 32-bit pairs in ONE pass (row check included) and builds CSR + seed lists with the
    light per-row kernel (one wave, 512 entries in LDS).  The feed's semantics must be what dae_coo_to_csr gives `recommend`:
    the LAST duplicate wins (DAEs.py:33-35), zeros drop out, a row longer than the kernel's LDS takes its global-memory path,
    a column outside the vocabulary raises ValueError, a row index outside the feed is refused at submit and the pipeline lives on."""
    nt, na, H, k, B = 6000, 1000, 64, 100, 8
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=4, bias="zipf", n_tracks=nt)
    path = str(tmp_path / "init.pkl")
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = str(tmp_path / "unused"); batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    rng = np.random.default_rng(5)
    feeds = []
    for f in range(5):
        rows, cols, vals = [], [], []
        for r in range(B):
            n = 700 if (r == 3 and f == 1) else int(rng.integers(1, 60))        # one row past the 512-entry LDS buffer
            c = rng.integers(0, V, size=n)
            c[n // 2:] = rng.choice(c[:max(n // 2, 1)], size=n - n // 2)        # duplicates, in feed order ...
            v = rng.choice(np.array([1.0, 0.5, 0.15, 0.0], np.float32), size=n)  # ... with different weights, some of them zero
            rows += [r] * n; cols += c.tolist(); vals += v.tolist()
        order = rng.permutation(len(rows)) if f == 2 else np.arange(len(rows))    # one feed that is not row-ordered at all
        pos = np.stack([np.asarray(rows, np.int64)[order], np.asarray(cols, np.int64)[order]], 1)
        feeds.append((pos, np.asarray(vals, np.float32)[order]))
    want = [m.recommend(p, v, SEEDS_FROM_INPUT, k=k, dtype="f32") for p, v in feeds]
    got = list(m.recommend_iter([(p, v, SEEDS_FROM_INPUT, B) for p, v in feeds], k=k, dtype="f32"))
    for (gi, gs), (wi, ws) in zip(got, want):
        assert np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
    # a column outside [0, V): the device flags it, the loop raises like the host CSR builder
    bad = feeds[0][0].copy(); bad[5, 1] = V + 3
    with pytest.raises(ValueError):
        list(m.recommend_iter([(bad, feeds[0][1], SEEDS_FROM_INPUT, B)], k=k, dtype="f32"))
    big = feeds[0][0].copy(); big[7, 1] = 2 ** 40          # does not fit 32 bits: the same error, not a wrapped column
    with pytest.raises(ValueError):
        list(m.recommend_iter([(big, feeds[0][1], SEEDS_FROM_INPUT, B)], k=k, dtype="f32"))
    # a row index outside the feed is refused by submit; the same pipeline object then scores the next feeds
    pipe = _lib.Pipeline(m.weights["encoder_h"], m.biases["encoder_b"], m.weights["decoder_h"], m.biases["decoder_b"], nt,
                         k=k, group_rows=4 * B, lanes=2)
    try:
        assert pipe.submit(feeds[0][0], feeds[0][1], B)
        wrong = feeds[1][0].copy(); wrong[0, 0] = B
        with pytest.raises(_lib.DaeError, match="row index"):
            pipe.submit(wrong, feeds[1][1], B)
        neg = feeds[1][0].copy(); neg[3, 0] = -1
        with pytest.raises(_lib.DaeError, match="row index"):
            pipe.submit(neg, feeds[1][1], B)
        assert pipe.submit(feeds[1][0], feeds[1][1], B)
        pipe.flush()
        out = [pipe.poll(True, copy=True) for _ in range(2)]
        for (gi, gs), (wi, ws) in zip(out, want[:2]):
            assert np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        # a refused FIRST feed of a launch leaves no half-open launch behind
        with pytest.raises(_lib.DaeError, match="row index"):
            pipe.submit(wrong, feeds[1][1], B)
        pipe.flush()
        assert pipe.poll(False) is None
        assert pipe.submit(feeds[3][0], feeds[3][1], B)
        pipe.flush()
        gi, gs = pipe.poll(True, copy=True)
        assert np.array_equal(gi, want[3][0])
    finally:
        pipe.close()
