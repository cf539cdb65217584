#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by RUNNING THE REAL REFERENCE where it can be
imported (pure-Python parts only: TensorFlow is unavailable).  Run in the build
container (needs /root/reference); the fixtures it writes are data (inputs + expected outputs) and
are what travels to the GPU box -- nothing under tests/ reads /root/reference at test time.

    python tests/golden/make_golden.py

Writes
  data/train, data/challenge_inorder_5to100     <- reference Spotify_train / Spotify_challenge
  data/test-reader5f                            <- hand-built in the 5-field layout the reference's READER unpacks
                                                   (the reference's Spotify_test raises)
  data/test-0, test-1, test-5, test-25r         <- THIS repo's repaired Spotify_test on a held-out slice
                                                   (mpd/mpd.slice.100-159.json.gz); readme.md:69 seed patterns
  expected_readers.json.gz                      <- reference utils/data_reader.py outputs
  expected_metrics.json                         <- reference utils/metrics.py outputs
  expected_ranking.npz                          <- reference main_runner/main_challenge.py cand_generate outputs
                                                   (numpy-only function; `tensorflow` is an inert module object
                                                   while the file is imported)
  forward_small.npz                             <- forward/top-k vectors of THIS repo's restatement
                                                   (oracle/: labelled as such, parity unpinned)
"""
import io
import json
import os
import random
import sys
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
DATA = os.path.join(HERE, "data")


def synth_mpd(path, n_playlists, rng, first_pid, n_tracks=150, n_artists=45):
    """A tiny MPD-shaped slice with Zipf track popularity and repeated artists."""
    playlists = []
    for p in range(n_playlists):
        n = int(rng.integers(3, 70))
        tr = np.minimum(n_tracks - 1, np.floor(np.exp(rng.random(n) * np.log(n_tracks))).astype(int) - 1)
        tracks = []
        for pos, t in enumerate(tr):
            t = int(max(t, 0))
            tracks.append({"pos": pos, "track_uri": "spotify:track:T%04d" % t,
                           "artist_uri": "spotify:artist:A%03d" % (t % n_artists),
                           "track_name": "t%d" % t, "artist_name": "a%d" % (t % n_artists)})
        if p == 1:      # one track / artist seen exactly once: spotify_reader.py:140 needs a count
            tracks.append({"pos": len(tracks), "track_uri": "spotify:track:T9999%d" % first_pid,
                           "artist_uri": "spotify:artist:A999%d" % first_pid,
                           "track_name": "rare", "artist_name": "rare"})   # == min_count - 1
        playlists.append({"name": "Chill #%d vibes!" % p if p % 3 else "ROCK & roll_%d" % p,
                          "pid": first_pid + p, "tracks": tracks, "num_tracks": n})
    with open(path, "w") as f:
        json.dump({"info": {}, "playlists": playlists}, f)


def synth_heldout(path, rng, first_pid, n_tracks=150, n_artists=45):
    """Held-out playlists long enough for every seed pattern the golden splits use (readme.md:69; accepted
    answer counts spotify_reader.py:231-242): lengths 12..160, a few tracks below the training min-count
    (id -1: out-of-vocabulary answers) and a few never seen in training (dropped, spotify_reader.py:221)."""
    playlists = []
    lengths = [12, 20, 35, 48, 60, 75, 90, 110, 130, 160]
    for p in range(60):
        n = lengths[p % len(lengths)] + int(rng.integers(0, 4))
        tr = np.minimum(n_tracks - 1, np.floor(np.exp(rng.random(n) * np.log(n_tracks))).astype(int) - 1)
        tracks = []
        for pos, t in enumerate(tr):
            t = int(max(t, 0))
            uri, art = "T%04d" % t, "A%03d" % (t % n_artists)
            if pos % 17 == 5:
                uri, art = "T99990", "A9990"            # seen once in training: below min_count -> id -1
            if pos % 23 == 7:
                uri, art = "T7777%d" % p, "A7777"       # never seen in training -> dropped
            tracks.append({"pos": pos, "track_uri": "spotify:track:" + uri, "artist_uri": "spotify:artist:" + art,
                           "track_name": "t", "artist_name": "a"})
        playlists.append({"name": ["Road Trip!! %d" % p, "late night <3 #%d" % p, "GYM/workout_%d" % p][p % 3],
                          "pid": first_pid + p, "tracks": tracks, "num_tracks": n})
    with open(path, "w") as f:
        json.dump({"info": {}, "playlists": playlists}, f)


def synth_challenge(path, rng, n_tracks=150, n_artists=45):
    playlists = []
    pid = 900000
    for num_samples in [5, 10, 25, 100, 100, 5, 25, 10, 100, 5, 10, 25, 100]:
        tr = np.minimum(n_tracks - 1, np.floor(np.exp(rng.random(num_samples) * np.log(n_tracks))).astype(int) - 1)
        tracks = [{"pos": pos, "track_uri": "spotify:track:T%04d" % int(max(t, 0)),
                   "artist_uri": "spotify:artist:A%03d" % (int(max(t, 0)) % n_artists)}
                  for pos, t in enumerate(tr)]
        pl = {"pid": pid, "num_samples": num_samples, "tracks": tracks, "num_holdouts": 10,
              "num_tracks": num_samples + 10}
        if pid % 4:
            pl["name"] = "Summer Mix %d" % pid
        playlists.append(pl)
        pid += 1
    with open(path, "w") as f:
        json.dump({"playlists": playlists}, f)


def tolist(a):
    return np.asarray(a).astype(np.int64).tolist()


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs %s (build container only)" % REF)
    sys.path.insert(0, REF)
    if not hasattr(np, "int"):
        np.int = int                       # numpy 2.x removed the alias the reference uses
    os.makedirs(DATA, exist_ok=True)
    tmp = "/tmp/_golden_mpd"
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(20180630)

    # ---- reference preprocessing (spotify_reader.py) -------------------------------------------
    from utils import spotify_reader as sr
    synth_mpd(os.path.join(tmp, "mpd.slice.0-59.json"), 60, rng, 0)
    synth_mpd(os.path.join(tmp, "mpd.slice.60-99.json"), 40, rng, 60)
    synth_challenge(os.path.join(tmp, "challenge_set.json"), rng)
    with redirect_stdout(io.StringIO()):
        sr.Spotify_train([os.path.join(tmp, "mpd.slice.0-59.json"), os.path.join(tmp, "mpd.slice.60-99.json")],
                         2, 2, True, DATA)
        sr.Spotify_challenge([os.path.join(tmp, "challenge_set.json")], os.path.join(DATA, "train"),
                             DATA, [5, 10, 25, 100], True)
    train = json.load(open(os.path.join(DATA, "train")))
    n_tracks = len(train["track_uri2id"])

    # the inputs of that preprocessing run travel too (gzip), so that this repo's own spotify_reader can
    # be checked byte for byte against the files the reference wrote (tests/test_preprocess_cpu.py)
    import gzip
    os.makedirs(os.path.join(HERE, "mpd"), exist_ok=True)
    synth_heldout(os.path.join(tmp, "mpd.slice.100-159.json"), np.random.default_rng(20180701), 100)
    for name in ("mpd.slice.0-59.json", "mpd.slice.60-99.json", "mpd.slice.100-159.json", "challenge_set.json"):
        with open(os.path.join(tmp, name), "rb") as fi, gzip.GzipFile(os.path.join(HERE, "mpd", name + ".gz"), "wb",
                                                                       mtime=0) as fo:
            fo.write(fi.read())
    titles = ["Chill #1 vibes!", "ROCK & roll_2", "  a..b,,c  ", "Summer/Fun <3 + more - 2018", "ÀÉÎõü ok",
              "x" * 40, "", "1234567890/<>+-", "Tabs\tand\nnewlines", "UPPER lower MiXeD (feat. Someone) [live]"]
    with open(os.path.join(HERE, "expected_titles.json"), "w") as f:
        json.dump([{"title": t, "normalized": sr.normalize_name(t),
                    "ixs_raw": sr.change_title2ixs(t), "ixs_norm": sr.change_title2ixs(sr.normalize_name(t))}
                   for t in titles], f)

    # test split in the layout the reference's READER unpacks (data_reader.py:158):
    # [seed, seed_art, answer, seed_cls, answer_cls]
    test_pl = []
    for p in train["playlists"][:23]:
        trk = p[0]
        if len(trk) < 8:
            continue
        seed, answer = trk[:5], [t for t in dict.fromkeys(trk[5:]) if t not in trk[:5]]
        answer = answer + [-1] if len(answer) % 2 else answer      # out-of-vocabulary answers
        test_pl.append([seed, p[1][:5], answer, [0] * len(seed), [0] * len(answer)])
    with open(os.path.join(DATA, "test-reader5f"), "w") as f:
        json.dump({"playlists": test_pl}, f)

    # the seed patterns of readme.md:69 that BASELINE.json configs[4] names (0 / 1 / 5 / 25r), written by this
    # repo's repaired generator (the reference's raises) from the held-out slice, one RNG stream for all splits
    sys.path.insert(0, ROOT)
    from spotify_recsys_challenge_2018_amd.utils import spotify_reader as my_sr
    split_rng = random.Random(180610)
    with redirect_stdout(io.StringIO()):
        for n_seeds, shuffled in ((0, False), (1, False), (5, False), (25, True)):
            t = my_sr.Spotify_test([os.path.join(tmp, "mpd.slice.100-159.json")], os.path.join(DATA, "train"),
                                   n_seeds, DATA, shuffled, rng=split_rng)
            assert t.num_playlists >= 8, (n_seeds, t.num_playlists)

    # ---- reference readers (utils/data_reader.py) ----------------------------------------------
    from utils import data_reader as dr
    exp = {"n_tracks": n_tracks}
    with redirect_stdout(io.StringIO()):
        random.seed(1234)
        r = dr.data_reader(DATA, "train", 16)
        out = []
        for _ in range(9):                                           # crosses the epoch wrap
            tp, ap, yp, titles, tv, av = r.next_batch()
            out.append({"trk": tolist(tp), "art": tolist(ap), "y": tolist(yp), "titles": titles,
                        "trk_val": list(map(float, tv)), "art_val": list(map(float, av)),
                        "train_idx": r.train_idx})
        exp["data_reader"] = out

        for key, ft in (("firstN_frac", [0.0, 0.3]), ("firstN_count", [1.0, 5.0])):
            random.seed(4321)
            r = dr.data_reader_firstN(DATA, "train", 16, ft)
            out = []
            for _ in range(8):
                tp, ap, yp, titles, tv, av = r.next_batch()
                out.append({"trk": tolist(tp), "art": tolist(ap), "y": tolist(yp),
                            "trk_val": list(map(float, tv)), "art_val": list(map(float, av)),
                            "train_idx": r.train_idx})
            exp[key] = out

        r = dr.data_reader_challenge(DATA, "challenge_inorder_5to100", 5)
        out = []
        while True:
            xp, seed, titles, texist, pid, xo = r.next_batch()
            out.append({"x": tolist(xp), "seed": seed, "titles": titles, "titles_exist": texist,
                        "pid": pid, "x_ones": list(map(float, xo))})
            if r.ch_idx == 0:
                break
        exp["challenge"] = out
        exp["challenge_meta"] = {"num_tracks": r.num_tracks, "num_items": r.num_items,
                                 "in_order": r.is_in_order}

        r = dr.data_reader_test(DATA, "test-reader5f", 6, 1000)
        out = []
        while True:
            tp, seed, answer, _cls, _afg = r.next_batch_test()
            out.append({"x": tolist(tp), "seed": seed, "answer": answer})
            if r.test_idx == 0:
                break
        exp["test"] = out
    import gzip
    with gzip.GzipFile(os.path.join(HERE, "expected_readers.json.gz"), "wb", mtime=0) as f:   # reproducible bytes
        f.write(json.dumps(exp).encode())

    # ---- reference metrics (utils/metrics.py) ---------------------------------------------------
    from utils import metrics as rm
    cases = []
    mrng = np.random.default_rng(7)
    for i in range(40):
        n_ans = int(mrng.integers(1, 60))
        answer = [int(x) for x in mrng.choice(400, size=n_ans, replace=False)]
        if i % 5 == 0:
            answer += [-1, -1]
        cand = [int(x) for x in mrng.permutation(400)[:int(mrng.integers(max(n_ans, 5), 300))]]
        cases.append({"answer": answer, "cand": cand,
                      "r_precision": rm.get_r_precision(answer, cand, [0] * len(answer), [10, 20, 30]),
                      "ndcg": rm.get_ndcg(answer, cand), "rsc": rm.get_rsc(answer, cand)})
    with open(os.path.join(HERE, "expected_metrics.json"), "w") as f:
        json.dump(cases, f)

    # ---- reference ranking (main_runner/main_challenge.py:26-41 cand_generate) --------------------
    # The module imports tensorflow at the top (and models/DAEs.py does) but cand_generate itself is numpy + list
    # operations only: an inert module object under the name lets the import go through; nothing of it is called.
    import types
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    from main_runner import main_challenge as rmc
    rrng = np.random.default_rng(500)
    n_cols, k_ref = 3000, 500
    id2uri = {str(i): "U%06d" % i for i in range(n_cols)}
    rows, seeds_l, kinds = [], [], []

    def sig(z):
        return (1.0 / (1.0 + np.exp(-z.astype(np.float32)))).astype(np.float32)
    for r in range(26):
        z = (rrng.standard_normal(n_cols) * 3.0 - np.log1p(np.arange(n_cols)) * 0.8 + 4.0).astype(np.float32)
        kind = "plain"
        if r % 4 == 1:                                 # saturated head: a plateau of exactly 1.0f (fp32 sigmoid)
            z[: 40 + 30 * (r // 4)] += 30.0
            kind = "plateau_top"
        if r % 4 == 2:                                 # quantised scores: ties everywhere, incl. across rank 500
            z = np.round(z * 2.0) / 2.0
            kind = "ties_everywhere"
        y = sig(z)
        order = np.argsort(-y, kind="stable")
        if r % 4 == 3:                                 # distinct scores: the answer is unique
            y = (np.float32(1.0) - np.arange(n_cols, dtype=np.float32)[np.argsort(order)] / np.float32(4096.0))
            kind = "distinct"
        n_seed = [0, 1, 5, 10, 25, 100][r % 6]
        sd = [int(x) for x in rrng.choice(order[:800], size=n_seed, replace=False)]
        if r % 5 == 0 and sd:
            sd = sd + sd[:3]                           # duplicated seeds: the second remove() raises -> ignored
        if r % 7 == 0:
            sd = sd + [n_cols + 5, n_cols + 77]        # ids beyond the track range (artist ids never are seeds,
            #                                            but remove() of an absent value is silently skipped)
        rows.append(y); seeds_l.append(sd); kinds.append(kind)
    short = sig((rrng.standard_normal(420) * 2.0).astype(np.float32))        # fewer than 500 rankable columns
    got = [rmc.cand_generate(y, sd, id2uri) for y, sd in zip(rows, seeds_l)]
    got_short = rmc.cand_generate(short, [3, 7, 7, 419], id2uri)
    exp_ids = np.full((len(rows), k_ref), -1, np.int32)
    for r, uris in enumerate(got):
        assert all(u.startswith("spotify:track:U") for u in uris)
        exp_ids[r, :len(uris)] = [int(u[len("spotify:track:U"):]) for u in uris]
    seed_ptr = np.zeros(len(rows) + 1, np.int64)
    np.cumsum([len(sd) for sd in seeds_l], out=seed_ptr[1:])
    np.savez_compressed(os.path.join(HERE, "expected_ranking.npz"),
                        note="outputs of /root/reference/main_runner/main_challenge.py cand_generate (numpy %s)"
                             % np.__version__,
                        scores=np.stack(rows), seed_ptr=seed_ptr,
                        seed_flat=np.array([x for sd in seeds_l for x in sd], np.int64), kinds=np.array(kinds),
                        expected_ids=exp_ids, uris_row0=np.array(got[0]),
                        short_scores=short, short_seeds=np.array([3, 7, 7, 419], np.int64),
                        short_expected=np.array([int(u[len("spotify:track:U"):]) for u in got_short], np.int32))

    # ---- forward / top-k vectors of this repo's own restatement (NOT reference outputs) ----------
    import oracle
    from oracle import dae_numpy as dn
    from spotify_recsys_challenge_2018_amd.models.DAEs import coo_to_csr, seeds_to_csr
    from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights
    V, nt, H, B, k = 2000, 1500, 32, 8, 500
    W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
    pos, ones, seeds = make_playlists(B, nt, V - nt, seed=1)
    rp, col, val = coo_to_csr(pos, ones, B, V)
    srp, sc = seeds_to_csr(seeds, B, nt)
    h = oracle.encode(rp, col, val, W_enc, b_enc)
    z = oracle.decode(h, W_dec, b_dec, 0, nt)
    s, i = oracle.topk(z, k, srp, sc)
    x = dn.sparse_to_dense(pos, ones, B, V)
    _, h_np, z_np = dn.forward(x, W_enc, b_enc, W_dec, b_dec)
    np.savez_compressed(os.path.join(HERE, "forward_small.npz"),
                        note="generated by oracle/ (this repo's CPU restatement), not by TensorFlow",
                        shape=np.array([V, nt, H, B, k]), pos=pos, ones=ones,
                        seeds_row_ptr=srp, seeds_col=sc,
                        h=h, logits_tracks_row0=z[0], topk_idx=i, topk_score=s,
                        h_numpy_dense=h_np, w_checksum=np.array([W_enc.sum(dtype=np.float64),
                                                                 W_dec.sum(dtype=np.float64)]))
    print("golden fixtures written under", HERE)
    for f in sorted(os.listdir(HERE)) + ["data/" + x for x in sorted(os.listdir(DATA))]:
        p = os.path.join(HERE, f)
        if os.path.isfile(p):
            print("  %-40s %8d B" % (f, os.path.getsize(p)))


if __name__ == "__main__":
    main()
