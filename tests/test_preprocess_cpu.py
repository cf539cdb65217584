"""CPU: this repo's MPD preprocessing (utils/spotify_reader.py, data_generator.py) against files and
vectors the REAL reference produced (tests/golden/make_golden.py imports /root/reference/utils/
spotify_reader.py in the build container; the inputs of that run travel gzip'd under golden/mpd)."""
import gzip
import json
import os
import shutil

import pytest

from spotify_recsys_challenge_2018_amd.utils import spotify_reader as sr

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def mpd(tmp_path):
    out = {}
    for name in ("mpd.slice.0-59.json", "mpd.slice.60-99.json", "mpd.slice.100-159.json", "challenge_set.json"):
        p = tmp_path / name
        with gzip.open(os.path.join(G, "mpd", name + ".gz"), "rb") as fi, open(p, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        out[name] = str(p)
    return out


def test_title_functions_match_reference_vectors():
    for case in json.load(open(os.path.join(G, "expected_titles.json"))):
        assert sr.normalize_name(case["title"]) == case["normalized"]
        assert sr.change_title2ixs(case["title"]) == case["ixs_raw"]
        assert sr.change_title2ixs(sr.normalize_name(case["title"])) == case["ixs_norm"]
    assert sr.NUM_CHAR == 41 and sr.MAX_TITLE_LEN == 25


def test_train_and_challenge_files_are_byte_identical_to_the_reference(mpd, tmp_path, capsys):
    out = tmp_path / "data"
    sr.Spotify_train([mpd["mpd.slice.0-59.json"], mpd["mpd.slice.60-99.json"]], 2, 2, True, str(out))
    assert (out / "train").read_bytes() == open(os.path.join(G, "data", "train"), "rb").read()
    sr.Spotify_challenge([mpd["challenge_set.json"]], str(out / "train"), str(out), [5, 10, 25, 100], True)
    assert (out / "challenge_inorder_5to100").read_bytes() == \
        open(os.path.join(G, "data", "challenge_inorder_5to100"), "rb").read()
    # ids are popularity ranks: counts never increase with the id
    tr = json.load(open(out / "train"))
    counts = tr["track_count"]
    assert all(a >= b for a, b in zip(counts, counts[1:])) and len(counts) == len(tr["track_uri2id"])
    assert sorted(tr["artist_uri2id"].values())[0] == len(tr["track_uri2id"])


def test_min_count_cut_keeps_intent_where_the_snapshot_raises():
    ranked = [("a", 9), ("b", 5), ("c", 4), ("d", 2), ("e", 1)]      # nothing seen exactly 2 times... (min 3)
    uris, counts, m = sr.create_uri2id(ranked, 3, 10)
    assert uris == list("abcde") and counts == [9, 5, 4] and m == {"a": 10, "b": 11, "c": 12}
    assert sr.create_uri2id(ranked, 1, 0)[2] == {u: i for i, u in enumerate("abcde")}


def test_test_split_generator_repaired_layout_feeds_the_reader(mpd, tmp_path, capsys):
    out = tmp_path / "data"
    sr.Spotify_train([mpd["mpd.slice.0-59.json"]], 2, 2, True, str(out))
    t = sr.Spotify_test([mpd["mpd.slice.60-99.json"]], str(out / "train"), 5, str(out), True)
    assert t.num_playlists > 0 and os.path.exists(out / "test-5r")
    d = json.load(open(out / "test-5r"))
    assert d["class_divpnt"] == json.load(open(out / "train"))["class_divpnt"]
    for seed, seed_art, ixs, answers in d["playlists"]:        # the generator's layout, spotify_reader.py:286
        assert len(seed) <= 5 and len(ixs) == sr.MAX_TITLE_LEN and any(i != -1 for i in ixs)
        assert not (set(seed) & {a for a in answers if a != -1})          # answers exclude the seeds
        known = [a for a in answers if a != -1]
        assert len(known) == len(set(known))                              # known answers are unique
    from spotify_recsys_challenge_2018_amd.utils.data_reader import data_reader_test
    r = data_reader_test(str(out), "test-5r", 4, 1000)
    x, seeds, answers, titles, ones = r.next_batch_test()
    assert len(seeds) == min(4, t.num_playlists) and x.shape[1] == 2 and len(ones) == len(x)
    # the titles survive the round trip: --title evaluation feeds them (main_train.py:69-79)
    assert r.has_titles and titles == [p[2] for p in d["playlists"][:len(seeds)]]
    assert all(len(tt) == sr.MAX_TITLE_LEN and any(i != -1 for i in tt) for tt in titles)
    # same RNG seed -> same split (the snapshot seeds the module RNG with 180610)
    t2 = sr.Spotify_test([mpd["mpd.slice.60-99.json"]], str(out / "train"), 5, str(tmp_path / "again"), True)
    assert t2.playlists == t.playlists


def test_committed_golden_splits_are_what_the_generator_writes(mpd, tmp_path, capsys):
    """tests/golden/data/test-{0,1,5,25r}: readme.md:69 seed patterns from the held-out slice, one RNG stream."""
    import random
    out = tmp_path / "data"
    rng = random.Random(180610)
    for n, shuffled, name in ((0, False, "test-0"), (1, False, "test-1"), (5, False, "test-5"), (25, True, "test-25r")):
        sr.Spotify_test([mpd["mpd.slice.100-159.json"]], os.path.join(G, "data", "train"), n, str(out), shuffled, rng=rng)
        assert (out / name).read_bytes() == open(os.path.join(G, "data", name), "rb").read()
        pl = json.load(open(out / name))["playlists"]
        lo, hi = sr._ANSWER_RANGE[n]
        assert len(pl) >= 8
        for seed, seed_art, ixs, answers in pl:
            assert len(seed) <= n and (n == 0) == (len(seed) == 0 and len(seed_art) == 0)
            assert any(a == -1 for p in pl for a in p[3])                  # out-of-vocabulary answers occur
    # in-order splits keep the playlist order: the seeds of test-1 are the first seeds of test-5
    p1 = {tuple(p[2]): p[0] for p in json.load(open(out / "test-1"))["playlists"]}
    p5 = {tuple(p[2]): p[0] for p in json.load(open(out / "test-5"))["playlists"]}
    common = [k for k in p1 if k in p5 and p1[k] and p5[k]]
    assert common and all(p5[k][0] == p1[k][0] for k in common)


def test_data_generator_cli(mpd, tmp_path, capsys):
    from spotify_recsys_challenge_2018_amd import data_generator as dg
    tr, te = tmp_path / "mpd_train", tmp_path / "mpd_test"
    tr.mkdir(); te.mkdir()
    shutil.copy(mpd["mpd.slice.0-59.json"], tr); shutil.copy(mpd["mpd.slice.60-99.json"], te)
    out = tmp_path / "data"
    assert dg.main(["--datadir", str(out), "--mpd_tr", str(tr), "--mpd_te", str(te), "--mincount_trk", "2",
                    "--mincount_art", "2", "--challenge", mpd["challenge_set.json"]]) == 0
    names = set(os.listdir(out))
    # readme.md:69: seed 0, 1, 5, 10, 25, 100 in playlist order + 25r, 100r shuffled (files with no qualifying
    # playlist are still written, empty)
    assert {"train", "test-0", "test-1", "test-5", "test-10", "test-25", "test-100", "test-25r", "test-100r",
            "challenge_inorder_5", "challenge_inorder_10to100"} <= names
    assert not any(n.endswith("r") and n[5:-1] in ("0", "1", "5", "10") for n in names if n.startswith("test-"))
