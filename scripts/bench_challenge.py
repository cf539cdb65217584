#!/usr/bin/env python
"""What a `main.py --challenge` user sees at full size: the driver end to end (challenge file of 10 000 playlists read
from disk, batches built by the reader, scoring through recommend_iter, ranked ids -> 'spotify:track:<uri>' rows, result
pickle written) on a synthetic data directory with the MPD's shape (140 000 tracks + 30 000 artists, hidden 256).
    python scripts/bench_challenge.py [--playlists 10000] [--batch 256] [--profile]"""
import argparse
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--playlists", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    nt, na, H = 140000, 30000, 256
    V = nt + na
    root = tempfile.mkdtemp(prefix="dae_ch_")
    data = os.path.join(root, "data"); os.makedirs(data)
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    with open(os.path.join(root, "dae.pkl"), "wb") as f:
        pickle.dump([W_enc, W_dec, b_enc, b_dec], f)
    pos, _ones, seeds = make_playlists(args.playlists, nt, na, seed=3)
    arts = [[] for _ in range(args.playlists)]
    for r, c in pos[pos[:, 1] >= nt]:
        arts[int(r)].append(int(c))
    rng = np.random.default_rng(0)
    pls = [[seeds[i], arts[i], [int(x) for x in rng.integers(0, 41, 12)] + [-1] * 13, [1], 1000000 + i]
           for i in range(args.playlists)]
    id2uri = {str(i): "%022d" % i for i in range(nt)}
    with open(os.path.join(data, "challenge"), "w") as f:
        json.dump({"playlists": pls, "id2uri": id2uri, "num_tracks": nt, "num_items": V, "in_order": True,
                   "max_title_len": 25, "num_char": 41}, f)

    class Conf:
        dir = root; data_dir = data; challenge_data = "challenge"; batch = args.batch; hidden = H; lr = 0.005
        reg_lambda = 0.0; initval = os.path.join(root, "dae.pkl"); DAEval = initval; save = os.path.join(root, "unused")
        result = os.path.join(root, "result.pkl"); verbose = False; allow_no_title = True; char_model = "none"
        device_index = 0
    from spotify_recsys_challenge_2018_amd.main_runner import main_challenge
    import torch
    torch.zeros(1).cuda()                                  # context creation is not the driver's time
    t0 = time.perf_counter()
    if args.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    out = main_challenge.run(Conf())
    if args.profile:
        pr.disable()
    dt = time.perf_counter() - t0
    assert len(out) == args.playlists and all(len(r) == 501 for r in out[:50])
    print("main.py --challenge end to end: %d playlists in %.2f s = %.0f playlists/s (batch %d; file read, model load and "
          "prepack, batches, scoring, URI rows, result pickle)" % (args.playlists, dt, args.playlists / dt, args.batch))
    if args.profile:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
