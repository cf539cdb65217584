#!/usr/bin/env python
"""The titled drivers' loop alone, one mode per call (for rocprofv3): python scripts/time_title.py <f32|bf16|exact_bf16> [n_feeds]
Shapes of scripts/bench_title.py ([TITLE] batch = 150, filters 3/5/7/9 x 100, 170 000 columns)."""
import os
import pickle
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib as _l   # noqa: E402
if os.environ.get("DAE_LIB_AB"):       # A/B against another build of the library
    _l.LIB_PATH = os.environ["DAE_LIB_AB"]
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE_title, SEEDS_FROM_INPUT   # noqa: E402
from spotify_recsys_challenge_2018_amd.models.title_models import get_model   # noqa: E402
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights   # noqa: E402


def build(B=150, nt=140000, na=30000, H=256):
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias=os.environ.get("BIAS", "zipf"), n_tracks=nt)
    W_dec = (W_dec * np.float32(float(os.environ.get("SCALE", "1")))).astype(np.float32)
    path = "/tmp/_title_dae.pkl"
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        batch = B; n_input = V; n_output = V; n_tracks = nt; hidden = H; lr = 0.001; reg_lambda = 0.0
        char_emb = 50; strmaxlen = 25; charsize = 41; char_model = 'Char_CNN'; filter_num = 100
        filter_size = [3, 5, 7, 9]; save = "/tmp/_t_unused"; initval = "NULL"; DAEval = path; title_lr = 0.001
    mt = get_model(C()); mt.fit()
    m = DAE_title(C(), mt); m.fit()
    pos, ones, _ = make_playlists(B, nt, na, seed=1)
    rng = np.random.default_rng(0)
    titles = rng.integers(0, 41, (B, 25)); titles[:, 18:] = -1
    use = np.ones(B, np.float32)
    return m, (pos, ones, SEEDS_FROM_INPUT, B, titles, use)


def main():
    import torch
    mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    m, feed = build()
    if os.environ.get('TITLE_COALESCE'):
        m.coalesce = int(os.environ['TITLE_COALESCE'])
    if os.environ.get('TITLE_LANES'):
        m.n_lanes = int(os.environ['TITLE_LANES'])
    for _ in m.recommend_iter([feed] * 10, k=500, dtype=mode, want_scores=False):
        pass
    if os.environ.get('TITLE_GC', 'freeze') == 'freeze':      # as main.py --challenge does before its loop (a full collection of
        import gc                                                # the interpreter's heap otherwise lands in the timed loop: 35 ms)
        gc.collect(); gc.freeze()
    elif os.environ.get('TITLE_GC') == 'off':
        import gc
        gc.disable()
    spin_us = float(os.environ.get('TITLE_CONSUMER_US', '0'))      # a consumer that works this long on every feed's lists
    torch.cuda.synchronize(); t0 = time.perf_counter()
    first = None
    stamps = []
    for _ in m.recommend_iter([feed] * n, k=500, dtype=mode, want_scores=False):
        if first is None:
            first = time.perf_counter() - t0
        stamps.append(time.perf_counter())
        if spin_us:
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < spin_us:
                pass
    if os.environ.get('TITLE_GAPS'):
        d = np.diff(np.asarray(stamps)) * 1e3
        big = np.argsort(d)[-12:]
        print("  gaps between yields (ms): median %.3f, sum %.1f; the 12 largest at feed#:" % (np.median(d), d.sum()),
              [(int(i), round(float(d[i]), 2)) for i in sorted(big)])
    t_end = time.perf_counter()               # (every list is on the host when it is handed out: nothing to wait for)
    torch.cuda.synchronize()
    print("  device-wide synchronize after the loop: %.2f ms (not part of the rate)" % ((time.perf_counter() - t_end) * 1e3))
    ds = (t_end - t0) / n
    print("titled recommend_iter %s: %.3f ms per batch of 150 = %.0f playlists/s (first lists after %.2f ms)" % (mode, ds * 1e3, 150 / ds, first * 1e3))
    for _g, pipe in m.__dict__.get("_pipes", {}).values():
        print("  pipeline:", pipe.stats(), pipe.times())
    if mode == "exact_bf16":
        print("  last launch:", m.title_model.ctx.exact_stats_read(), "guard", m.title_model.ctx.exact_guard_read(),
              "fallbacks", getattr(m, "_guard_fallbacks", 0))


if __name__ == "__main__":
    main()
