#!/usr/bin/env python
"""End-to-end rate THROUGH the Python shim (host feed -> H2D -> fused scoring -> D2H of idx/score),
i.e. what main_challenge.py-style driver code sees.  Never bench.py's `value` (that one starts with the
inputs resident in HBM); quoted in DESIGN.md section 5 as the PCIe-inclusive figure."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE          # noqa: E402
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights   # noqa: E402


def main():
    import pickle
    import torch
    nt, na, H = 140000, 30000, 256
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256      # 150 = the reference's [CHALLENGE] / [TITLE] batch
    V = nt + na
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    path = "/tmp/_shim_init.pkl"
    with open(path, "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)

    class C:
        save = "/tmp/_shim_unused"; batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
        n_tracks = nt; initval = path
    m = DAE(C()); m.fit()
    batches = [make_playlists(B, nt, na, seed=s) for s in range(8)]
    for dev_csr in (True, False):
        m.device_csr = dev_csr
        for p, o, s in batches[:2]:
            m.recommend(p, o, s, k=500)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for rep in range(5):
            for p, o, s in batches:
                m.recommend(p, o, s, k=500)
                n += B
        dt = time.perf_counter() - t0
        print("model.recommend, feed -> top-500 on the host, device_csr=%s: %.0f playlists/s (%.2f ms per batch of %d)"
              % (dev_csr, n / dt, dt / (n / B) * 1e3, B))
    # the drivers' loop: batches streamed through recommend_iter (upload / launch of batch n + 1 before the fetch of
    # batch n), seeds cut out of the input on the device, indices only -- what main.py --challenge runs
    from spotify_recsys_challenge_2018_amd.models.DAEs import SEEDS_FROM_INPUT
    m.device_csr = True
    it_dtype = "bf16" if "--bf16" in sys.argv else ("exact_bf16" if "--exact" in sys.argv else None)   # decode mode of the streamed loop
    if "--alone" in sys.argv:                 # one feed per launch, one context: the loop before coalescing / two lanes
        m.coalesce = 1
        m.n_lanes = 1
    for label, seeds_of, scores in (("seed lists from the host, idx + score", lambda s_: s_, True),
                                    ("seeds = input tracks (device), idx only", lambda s_: SEEDS_FROM_INPUT, False)):
        def feeds(reps):
            for _ in range(reps):
                for p_, o_, s_ in batches:
                    yield p_, o_, seeds_of(s_), B
        for _ in m.recommend_iter(feeds(1), k=500, want_scores=scores, dtype=it_dtype):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _idx, _sc in m.recommend_iter(feeds(25 if scores else 100), k=500, want_scores=scores, dtype=it_dtype):
            n += B
        dt = time.perf_counter() - t0
        print("model.recommend_iter%s (%s): %.0f playlists/s (%.3f ms per batch of %d)" % (" " + it_dtype if it_dtype else "", label, n / dt, dt / (n / B) * 1e3, B))
    # same answers either way
    a = m.recommend(batches[0][0], batches[0][1], batches[0][2], k=500)
    b = next(iter(m.recommend_iter([(batches[0][0], batches[0][1], SEEDS_FROM_INPUT, B)], k=500)))
    print("recommend_iter(SEEDS_FROM_INPUT) == recommend(seed lists):", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])))


if __name__ == "__main__":
    main()
