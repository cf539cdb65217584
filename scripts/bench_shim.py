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
    nt, na, H, B = 140000, 30000, 256, 256
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


if __name__ == "__main__":
    main()
