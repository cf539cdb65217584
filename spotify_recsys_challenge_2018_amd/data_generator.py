"""Reference data_generator.py (citations relative to /root/reference): MPD slice directories -> the
train / test-N / challenge files under --datadir.

    python -m spotify_recsys_challenge_2018_amd.data_generator --datadir ./data --mpd_tr ./mpd_train \\
        --mpd_te ./mpd_test [--challenge ./challenge_set.json] [--mincount_trk 5 --mincount_art 3]

The snapshot's script calls Spotify_train with one argument too few (data_generator.py:30: the
is_title_normalize flag is missing) and so cannot run; the flag is passed here (default: normalise,
as the shipped data was built, readme.md).  Slice files are taken in sorted order (the reference uses
os.listdir order, which is file-system dependent; ids of equally frequent tracks depend on it).

Test splits: the snapshot loops over seed counts [1, 5, 10, 25, 50, 100], all shuffled
(data_generator.py:35-37), which produces none of the files its own config.ini files name
(test_seed = 1,5,10,25,100,25r,100r / 0,1,5,10,25,100).  REPAIR: the seed pattern
readme.md:69 documents -- "seed 0, 1, 5, 10, 25, 100, 25r, 100r": first-N seeds in playlist order,
plus the two random-order splits.  All splits draw from ONE random stream seeded like the snapshot's
module RNG (spotify_reader.py:13), in the order below."""
import argparse
import os
import random

from .utils.spotify_reader import Spotify_challenge, Spotify_test, Spotify_train

TEST_SPLITS = ((0, False), (1, False), (5, False), (10, False), (25, False), (100, False),
               (25, True), (100, True))                                 # readme.md:69
# the challenge categories as the shipped run directories group them (0to1_inorder, 5_inorder,
# 10to100_inorder, 25to100_random)
CHALLENGE_GROUPS = ((True, (0, 1)), (True, (5,)), (True, (10, 25, 100)), (False, (25, 100)))
_OPTIONS = (  # flag, type, default, what it is
    ("--datadir", str, "./data", "where train / test-N / challenge_* are written"),
    ("--mpd_tr", str, "./mpd_train", "directory of MPD slices to train on"),
    ("--mpd_te", str, "./mpd_test", "directory of MPD slices held out for the test splits, or NULL"),
    ("--mincount_trk", int, 5, "a track needs this many playlists to enter the vocabulary"),
    ("--mincount_art", int, 3, "same for artists"),
    ("--challenge", str, "NULL", "challenge_set.json, or NULL"),
)


def slice_paths(directory):
    return [os.path.join(directory, name) for name in sorted(os.listdir(directory))]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, typ, default, text in _OPTIONS:
        ap.add_argument(flag, type=typ, default=default, help=text)
    ap.add_argument("--no_title_normalize", action="store_true")
    opt = ap.parse_args(argv)

    Spotify_train(slice_paths(opt.mpd_tr), opt.mincount_trk, opt.mincount_art, not opt.no_title_normalize,
                  opt.datadir)
    train_json = os.path.join(opt.datadir, "train")
    if opt.mpd_te != "NULL":
        held_out = slice_paths(opt.mpd_te)
        rng = random.Random(180610)                                     # spotify_reader.py:13
        for n_seeds, shuffled in TEST_SPLITS:
            Spotify_test(held_out, train_json, n_seeds, opt.datadir, shuffled, rng=rng)
    if opt.challenge != "NULL":
        for in_order, counts in CHALLENGE_GROUPS:
            Spotify_challenge([opt.challenge], train_json, opt.datadir, list(counts), in_order)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
