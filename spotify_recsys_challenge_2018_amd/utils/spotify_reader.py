"""MPD slices -> the `train` / `test-N[r]` / `challenge_*` JSON files the readers consume (reference
utils/spotify_reader.py; citations relative to /root/reference).  Host-side preprocessing, SURVEY.md
8f row 4: outside the GPU hot path, here so that `main.py` can run on real MPD slices when supplied.

Byte-level contract: for the same input slices `Spotify_train` and `Spotify_challenge` write files
IDENTICAL to the reference's (tests/test_preprocess_cpu.py compares against files the real reference
wrote; json.dump(indent="\\t"), key order and id assignment are part of the format).

Id assignment (spotify_reader.py:63-72, :134-144): tracks are numbered 0.. in order of decreasing
playlist count (ties in first-seen order), artists continue after the last track id -- so an id IS a
popularity rank, which the scoring path exploits (threshold sample).

Repairs of snapshot defects, each marked REPAIR below:
  * create_uri2id: the snapshot finds the cut with list.index(min_count - 1), which raises when no
    item has exactly that count; the intent -- keep items seen >= min_count times -- is implemented.
  * Spotify_test: the snapshot reads self.class_divpnt / self.get_class, which do not exist (the
    popularity-class bookkeeping is a research leftover).  The class divide points
    come from the train file here; rows are written in the layout the snapshot's generator writes
    (spotify_reader.py:286): [seed_tracks, seed_artists, title_ixs, answers] -- the title indices
    are what `--title` evaluation feeds (main_train.py:69-79).  This repo's data_reader_test accepts
    that layout and the 5-field one the snapshot's reader unpacks (data_reader.py:158).
  * the module RNG: the snapshot seeds `random` once at import (spotify_reader.py:13) and
    data_generator builds every split in one process, so the splits draw from ONE stream; here the
    caller passes one `random.Random(180610)` to every Spotify_test (data_generator.main does).
"""
import json
import os
import random
import re

VARIOUS_ARTISTS_URI = '0LyfQWJT6nXafLPZqxe9Of'                 # spotify_reader.py:15
MAX_TITLE_LEN = 25
CHARS = 'abcdefghijklmnopqrstuvwxyz/<>+-1234567890'            # :17
CHAR2IX = {ch: i for i, ch in enumerate(CHARS)}
NUM_CHAR = len(CHAR2IX)

_PUNCT = re.compile(r"[.,#!$%\^\*;:{}=\_`~()@]")
_SPACE = re.compile(r'\s+')


def normalize_name(name):
    """spotify_reader.py:21-25: lower-case, listed punctuation -> space, runs of whitespace -> one."""
    return _SPACE.sub(' ', _PUNCT.sub(' ', name.lower())).strip()


def change_title2ixs(title):
    """spotify_reader.py:28-37: the first MAX_TITLE_LEN characters that are in the alphabet, as
    indices, right-padded with -1."""
    ixs = [CHAR2IX[ch] for ch in title if ch in CHAR2IX][:MAX_TITLE_LEN]
    return ixs + [-1] * (MAX_TITLE_LEN - len(ixs))


def _uri(s):
    return s.split(':')[2]


def _load_playlists(fullpaths):
    for path in fullpaths:
        with open(path) as f:
            for playlist in json.load(f)['playlists']:
                yield playlist


def _dump(obj, path):
    with open(path, 'w') as f:
        json.dump(obj, f, indent="\t")


def _by_count(hist):
    """Counter.most_common() order: decreasing count, ties in first-seen order (stable sort)."""
    return sorted(hist.items(), key=lambda kv: kv[1], reverse=True)


def create_uri2id(ranked, min_count, start_from):
    """spotify_reader.py:134-144.  ranked: [(uri, count)] by decreasing count.  Returns (all uris, the
    counts of the kept ones, uri -> id for the kept ones)."""
    uris = [u for u, _ in ranked]
    counts = [c for _, c in ranked]
    keep = len(counts)
    if min_count > 1:
        keep = next((i for i, c in enumerate(counts) if c < min_count), len(counts))    # REPAIR (intent)
    return uris, counts[:keep], {u: start_from + i for i, u in enumerate(uris[:keep])}


def get_cdf(count_list):
    """spotify_reader.py:156-164."""
    total = sum(count_list)
    out, cum = [], 0
    for c in count_list:
        cum += c
        out.append(cum / total)
    return out


def get_class_divpnt(cdf, points):
    """spotify_reader.py:166-174: for each point, (index of the first cdf value above it) - 1, searching
    on from the previous answer."""
    idx = [0]
    for p in points:
        for i in range(idx[-1], len(cdf)):
            if cdf[i] > p:
                idx.append(i - 1)
                break
    return idx[1:]


def get_class(class_divpnt, track_id):
    """REPAIR: popularity class of a track id (0 = most popular band) from the train file's divide
    points; the snapshot calls this function without defining it."""
    for cls, pnt in enumerate(class_divpnt):
        if track_id <= pnt:
            return cls
    return len(class_divpnt)


class Spotify_train:
    """A set of MPD slices -> `<save_dir>/train` (spotify_reader.py:41-119)."""

    def __init__(self, train_fullpaths, trk_min_count, art_min_count, is_title_normalize, save_dir):
        self.is_title_normalize = is_title_normalize
        track_hist, artist_hist = {}, {}
        pl_tracks, pl_artists, pl_titles = [], [], []
        for playlist in _load_playlists(train_fullpaths):
            name = playlist['name']
            pl_titles.append(normalize_name(name) if is_title_normalize else name)
            tracks = [_uri(t['track_uri']) for t in playlist['tracks']]
            artists = [_uri(t['artist_uri']) for t in playlist['tracks']]
            for u in tracks:
                track_hist[u] = track_hist.get(u, 0) + 1
            for u in artists:
                artist_hist[u] = artist_hist.get(u, 0) + 1
            pl_tracks.append(tracks)
            pl_artists.append(artists)

        total_trk, trk_counts, track_uri2id = create_uri2id(_by_count(track_hist), trk_min_count, 0)
        artist_hist.pop(VARIOUS_ARTISTS_URI, None)                      # :67
        _total_art, _art_counts, artist_uri2id = create_uri2id(_by_count(artist_hist), art_min_count,
                                                               len(track_uri2id))
        class_divpnt = get_class_divpnt(get_cdf(trk_counts), [0.3, 0.8, 0.9])    # :74-75

        playlists = []
        print("len %d %d %d" % (len(pl_tracks), len(pl_artists), len(pl_titles)))
        for tracks, artists, title in zip(pl_tracks, pl_artists, pl_titles):
            tracks_id = [track_uri2id[u] for u in tracks if u in track_uri2id]
            artists_id = [artist_uri2id[u] for u in artists if u in artist_uri2id]
            if not tracks_id and not artists_id:
                continue
            if len(tracks_id) > 250 or len(artists_id) > 250:           # :85-86
                continue
            playlists.append([tracks_id, artists_id, change_title2ixs(title)])
        self.num_playlists = len(playlists)

        os.makedirs(save_dir, exist_ok=True)
        file_data = {'is_title_normalize': is_title_normalize, 'max_title_len': MAX_TITLE_LEN,
                     'num_char': NUM_CHAR, 'track_total': total_trk, 'track_count': trk_counts,
                     'track_uri2id': track_uri2id, 'artist_uri2id': artist_uri2id,
                     'playlists': playlists, 'class_divpnt': class_divpnt}      # key order = file format
        print('train')
        _dump(file_data, os.path.join(save_dir, 'train'))
        print("num playlists: %d, tracks_total: %d, tracks>=min_count: %d, artists>=min_count: %d" %
              (self.num_playlists, len(total_trk), len(track_uri2id), len(artist_uri2id)))


# accepted numbers of held-out tracks per seed count (spotify_reader.py:231-242)
_ANSWER_RANGE = {0: (10, 50), 1: (9, 77), 5: (5, 95), 10: (30, 90), 25: (76, None), 100: (50, None)}


class Spotify_test:
    """MPD slices + the train file -> `<save_dir>/test-<N>[r]` (spotify_reader.py:177-286, repaired)."""

    def __init__(self, test_fullpaths, train_json, test_seeds_num, save_dir, is_shuffle, rng=None):
        with open(train_json) as f:
            train = json.load(f)
        track_uri2id, artist_uri2id = train['track_uri2id'], train['artist_uri2id']
        track_total = set(train['track_total'])
        class_divpnt = train['class_divpnt']                            # REPAIR: was never loaded
        normalize = bool(train['is_title_normalize'])                   # :186
        rng = rng if rng is not None else random.Random(180610)         # :13 seeds the module RNG
        n = test_seeds_num
        self.playlists = []
        for playlist in _load_playlists(test_fullpaths):
            tracks, artists = [], []
            for t in playlist['tracks']:
                tu = _uri(t['track_uri'])
                if tu not in track_total:                               # :221-222 unseen tracks are ignored
                    continue
                tracks.append(track_uri2id.get(tu, -1))
                artists.append(artist_uri2id.get(_uri(t['artist_uri']), -1))
            if len(tracks) <= n:
                continue
            l_answers = len(tracks) - n
            lo, hi = _ANSWER_RANGE.get(n, (None, None))
            if (lo is not None and l_answers < lo) or (hi is not None and l_answers > hi):
                continue
            if is_shuffle:
                order = list(range(len(tracks)))
                rng.shuffle(order)
                tracks = [tracks[i] for i in order]
                artists = [artists[i] for i in order]
            seed_trk = [t for t in tracks[:n] if t != -1]
            seed_art = [a for a in artists[:n] if a != -1]
            answers = []
            for t in tracks[n:]:                                        # :267-273: -1 (OOV) answers repeat
                if t not in seed_trk and (t == -1 or t not in answers):
                    answers.append(t)
            name_ = playlist['name']                                    # :279-282
            ixs = change_title2ixs(normalize_name(name_) if normalize else name_)
            self.playlists.append([seed_trk, seed_art, ixs, answers])   # :286
        self.num_playlists = len(self.playlists)
        name = 'test-' + str(n) + ('r' if is_shuffle else '')
        print(name)
        os.makedirs(save_dir, exist_ok=True)
        _dump({'playlists': self.playlists, 'class_divpnt': class_divpnt}, os.path.join(save_dir, name))
        print("num_playlists:%d" % self.num_playlists)


class Spotify_challenge:
    """challenge_set.json + the train file -> `<save_dir>/challenge_{inorder|random}_<N>[to<M>]`
    (spotify_reader.py:289-369)."""

    def __init__(self, challenge_fullpaths, train_json, save_dir, num_trk_lst, in_order):
        with open(train_json) as f:
            train = json.load(f)
        track_uri2id, artist_uri2id = train['track_uri2id'], train['artist_uri2id']
        normalize = bool(train['is_title_normalize'])
        self.playlists = []
        for playlist in _load_playlists(challenge_fullpaths):
            last_pos = playlist['tracks'][-1]['pos'] if playlist['tracks'] else -1
            num_samples = playlist['num_samples']
            if ((last_pos + 1 == num_samples) != in_order) or (num_samples not in num_trk_lst):   # :341-343
                continue
            tracks = [track_uri2id[u] for u in (_uri(t['track_uri']) for t in playlist['tracks'])
                      if u in track_uri2id]
            artists = [artist_uri2id[u] for u in (_uri(t['artist_uri']) for t in playlist['tracks'])
                       if u in artist_uri2id]
            is_name, ixs = 0, [-1] * MAX_TITLE_LEN
            if 'name' in playlist:
                is_name = 1
                name = playlist['name']
                ixs = change_title2ixs(normalize_name(name) if normalize else name)
            self.playlists.append([tracks, artists, ixs, [is_name], playlist['pid']])
        self.num_playlists = len(self.playlists)

        os.makedirs(save_dir, exist_ok=True)
        file_data = {'max_title_len': MAX_TITLE_LEN, 'num_char': NUM_CHAR, 'in_order': in_order,
                     'num_tracks': len(track_uri2id), 'num_items': len(track_uri2id) + len(artist_uri2id),
                     'id2uri': {v: k for k, v in track_uri2id.items()}, 'playlists': self.playlists}
        name = 'challenge' + ('_inorder' if in_order else '_random')
        name += ('_%d' % num_trk_lst[0]) if len(num_trk_lst) == 1 else ('_%dto%d' % (num_trk_lst[0], num_trk_lst[-1]))
        print(name)
        _dump(file_data, os.path.join(save_dir, name))
        print("num_playlists:%d" % self.num_playlists)
