#!/usr/bin/env python
"""The drivers' loop (DAE.recommend_iter on the library's dae_pipeline: host feeds in, host index lists out) per decode mode and
lane count.  usage: bench_loop.py [batch] native [modes f32,exact_bf16,bf16] [lanes 2,3,4]   (the second argument is kept for the
command lines of rounds 4 - 5: "native" is the only engine since round 6)"""
import os
import pickle
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib                                        # noqa: E402
if os.environ.get("DAE_LIB_AB"):       # A/B against another build of the library (scripts/build_exp.py)
    _lib.LIB_PATH = os.environ["DAE_LIB_AB"]
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE, SEEDS_FROM_INPUT          # noqa: E402
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights   # noqa: E402


def main():
    import torch
    nt, na, H = 140000, 30000, 256
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    engines = ["native"]
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "exact_bf16", "bf16"]
    lanes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    path = "/tmp/_loop_init.pkl"
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = "/tmp/_loop_unused"; batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    if os.environ.get("COALESCE"):       # A/B: feeds per launch (model.coalesce)
        m.coalesce = int(os.environ["COALESCE"])
    batches = [make_playlists(B, nt, na, seed=200 + s)[:2] for s in range(8)]

    def feeds(reps):
        for _ in range(reps):
            for p_, o_ in batches:
                yield p_, o_, SEEDS_FROM_INPUT, B
    for eng in engines:
        for mode in modes:
            for nl in lanes:
                m.n_lanes = nl or None
                m.__dict__.pop("_pipes", None)
                for _ in m.recommend_iter(feeds(40 if mode == "f32" else 120), k=500, want_scores=False, dtype=mode):
                    pass                                          # (~0.3 s: the device at its sustained state, as bench.py's prime phase)
                torch.cuda.synchronize()
                if os.environ.get("FREEZE"):          # as bench.py's host-loop rows: one full collection, survivors frozen
                    import gc
                    gc.collect(); gc.freeze()
                hold = None
                if os.environ.get("HOLD"):            # ... and one view of a result block kept alive across the timed loop
                    for hold in m.recommend_iter(feeds(1), k=500, want_scores=False, dtype=mode):
                        pass
                reps = (100 if mode == "f32" else 300) * int(os.environ.get("REPS_X", "1"))      # REPS_X=4: a loop long enough for the sustained clock
                t0 = time.perf_counter()
                n = 0
                for _i, _s in m.recommend_iter(feeds(reps), k=500, want_scores=False, dtype=mode):
                    n += B
                dt = time.perf_counter() - t0
                extra = ""
                if eng == "native":
                    for pp in m.__dict__.get("_pipes", {}).values():
                        extra = "  %s %s total_ms=%.1f" % (pp[1].times(), pp[1].stats(), dt * 1e3)
                print("%-6s %-10s lanes=%s batch=%d: %9.0f playlists/s (%.4f ms per feed)%s" % (eng, mode, nl or "default", B, n / dt, dt / (n / B) * 1e3, extra), flush=True)


if __name__ == "__main__":
    main()
