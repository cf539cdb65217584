"""Batch readers with the reference's class names, constructor arguments and return layouts
(utils/data_reader.py; citations relative to /root/reference), rebuilt around flat numpy arrays.

The reference builds every batch with a Python loop that makes four small numpy arrays per
playlist (np.array([..]).T, np.full_like, np.concatenate, ...).  Here each file is flattened once
at load time into CSR-like arrays (ids + offsets), and a batch is two fancy-index gathers.
The `random` module is consulted at exactly the same points and with the same arguments as the
reference (shuffle at the epoch wrap, randrange per playlist for firstN), so with the same RNG
state the outputs are value-identical -- tests/golden/ pins that against the real reference.

Returned `positions` are int64 [nnz, 2] (row-in-batch, item id), values are float32 arrays
(the reference returns Python lists of ints/floats; TF converts both to the same tensors).
"""
import json
import random

import numpy as np


def _flatten(list_of_lists):
    lens = np.fromiter((len(x) for x in list_of_lists), dtype=np.int64, count=len(list_of_lists))
    off = np.zeros(len(list_of_lists) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.fromiter((v for x in list_of_lists for v in x), dtype=np.int64, count=int(off[-1]))
    return flat, off


def _gather(flat, off, order):
    """Concatenate segments `order` of a flattened ragged array -> (values, row-in-batch, lens)."""
    starts, lens = off[order], off[order + 1] - off[order]
    total = int(lens.sum())
    rows = np.repeat(np.arange(len(order), dtype=np.int64), lens)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    return flat[np.repeat(starts, lens) + within], rows, lens


def _positions(rows, ids):
    return np.stack([rows, ids], axis=1).astype(np.int64) if rows.size else np.zeros((0, 2), np.int64)


class _TrainFile:
    """data/train (spotify_reader.py:93-108): playlists[i] = [track_ids, artist_ids, title_ixs]."""

    def __init__(self, data_dir, filename):
        with open(data_dir + '/' + filename) as f:
            d = json.load(f)
        self.num_tracks = len(d['track_uri2id'])                       # data_reader.py:12
        self.num_items = self.num_tracks + len(d['artist_uri2id'])     # :13
        self.max_title_len = d['max_title_len']
        self.num_char = d['num_char']
        self.class_divpnt = d.get('class_divpnt')
        self.playlists = d['playlists']
        self._index()

    def _index(self):
        self._trk, self._trk_off = _flatten([p[0] for p in self.playlists])
        self._art, self._art_off = _flatten([p[1] for p in self.playlists])
        self._order = np.arange(len(self.playlists), dtype=np.int64)


class data_reader(_TrainFile):
    """Whole-playlist training batches (data_reader.py:7-54)."""

    def __init__(self, data_dir, filename, batch_size):
        _TrainFile.__init__(self, data_dir, filename)
        self.batch_size = batch_size
        self.train_idx = 0

    def _take(self):
        """Indices of the next batch_size playlists; shuffles exactly where the reference does
        (data_reader.py:44-46: when the cursor hits the end, reset and random.shuffle)."""
        n = len(self.playlists)
        picked = []
        for _ in range(self.batch_size):
            picked.append(self._order[self.train_idx])
            self.train_idx += 1
            if self.train_idx == n:
                self.train_idx = 0
                perm = list(range(n))
                random.shuffle(perm)             # same RNG draw sequence as shuffling the list
                self._order = self._order[np.asarray(perm, dtype=np.int64)]
                self.playlists = [self.playlists[i] for i in perm]
        return np.asarray(picked, dtype=np.int64)

    def next_batch(self):
        order = self._take()
        trk, trk_rows, _ = _gather(self._trk, self._trk_off, order)
        art, art_rows, _ = _gather(self._art, self._art_off, order)
        trk_positions = _positions(trk_rows, trk)
        art_positions = _positions(art_rows, art)
        y_positions = np.concatenate((trk_positions, art_positions), 0)
        titles = [self._title(i) for i in order]
        return (trk_positions, art_positions, y_positions, titles,
                np.ones(len(trk_positions), np.float32), np.ones(len(art_positions), np.float32))

    def _title(self, orig_index):
        return self._titles[orig_index]

    def _index(self):
        _TrainFile._index(self)
        self._titles = [p[2] for p in self.playlists]


class data_reader_firstN(data_reader):
    """Training batches whose input keeps only the first N items of each playlist
    (data_reader.py:57-128): all positions are fed, the values are 1 for the first `given_num`
    items and 0 for the rest; given_num ~ randrange per playlist, separately for tracks/artists."""

    def __init__(self, data_dir, filename, batch_size, from_to):
        data_reader.__init__(self, data_dir, filename, batch_size)
        self.from_to = from_to

    def _given(self, length):
        lo_f, hi_f = self.from_to[0], self.from_to[1]
        if lo_f >= 1:                                            # data_reader.py:85-87
            n, m = int(lo_f), int(min(length, hi_f))
        else:                                                    # :89-90
            n, m = int(max(length * lo_f, 1)), int(max(length * hi_f, 1))
        return random.randrange(n, m + 1)                        # :91 / :109

    def next_batch(self):
        # the RNG is consulted playlist by playlist, tracks then artists, BEFORE the wrap
        # shuffle of that playlist's slot -- replay that order exactly
        n_pl = len(self.playlists)
        order, g_trk, g_art = [], [], []
        for _ in range(self.batch_size):
            oi = self._order[self.train_idx]
            lt = int(self._trk_off[oi + 1] - self._trk_off[oi])
            la = int(self._art_off[oi + 1] - self._art_off[oi])
            g_trk.append(self._given(lt) if lt != 0 else 0)
            g_art.append(self._given(la) if la != 0 else 0)
            order.append(oi)
            self.train_idx += 1
            if self.train_idx == n_pl:
                self.train_idx = 0
                perm = list(range(n_pl))
                random.shuffle(perm)
                self._order = self._order[np.asarray(perm, dtype=np.int64)]
                self.playlists = [self.playlists[i] for i in perm]
        order = np.asarray(order, dtype=np.int64)
        trk, trk_rows, trk_lens = _gather(self._trk, self._trk_off, order)
        art, art_rows, art_lens = _gather(self._art, self._art_off, order)

        def first_n(lens, given):
            within = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
            return (within < np.repeat(np.asarray(given, dtype=np.int64), lens)).astype(np.float32)

        trk_positions = _positions(trk_rows, trk)
        art_positions = _positions(art_rows, art)
        y_positions = np.concatenate((trk_positions, art_positions), 0)
        titles = [self._titles[i] for i in order]
        return (trk_positions, art_positions, y_positions, titles,
                first_n(trk_lens, g_trk), first_n(art_lens, g_art))


class data_reader_test:
    """Evaluation batches (data_reader.py:131-196).  File rows are either the layout the
    reference's generator writes, [seed_trk, seed_art, title_ixs, answers]
    (spotify_reader.py:286; what this repo's Spotify_test writes), or the 5-field layout its reader
    unpacks, [seed, seed_art, answer, seed_cls, answer_cls] (data_reader.py:158), which carries no
    title; both are accepted.
    Returns what main_train.py:64 unpacks: (x_positions, test_seed, test_answer, titles, x_ones):
    seed TRACKS only, weight 1 (main_train.py:66-68); `titles[i]` is the row's title indices, or None
    for a 5-field row (`has_titles` says whether the file carries them)."""

    def __init__(self, data_dir, filename, batch_size, test_num):
        print("now processing: " + filename)
        with open(data_dir + '/' + filename) as f:
            d = json.load(f)
        self.playlists = d['playlists'][:test_num]
        if test_num > len(self.playlists):
            print("the number of test will be changed to %d" % len(self.playlists))
        self.batch_size = batch_size
        self.test_idx = 0
        seeds, self._answers, self._titles = [], [], []
        for p in self.playlists:
            if len(p) == 4:
                seed, _art, title, answer = p
            else:
                seed, _art, answer = p[0], p[1], p[2]
                title = None
            seeds.append(seed)
            self._answers.append(answer)
            self._titles.append(title)
        self._seeds = seeds
        self.has_titles = bool(self._titles) and all(t is not None for t in self._titles)
        self._seed_flat, self._seed_off = _flatten(seeds)

    def next_batch_test(self):
        n = len(self.playlists)
        stop = min(self.test_idx + self.batch_size, n)
        order = np.arange(self.test_idx, stop, dtype=np.int64)
        self.test_idx = 0 if stop == n else stop                 # data_reader.py:185-188
        ids, rows, _ = _gather(self._seed_flat, self._seed_off, order)
        x_positions = _positions(rows, ids)
        return (x_positions, [self._seeds[i] for i in order], [self._answers[i] for i in order],
                [self._titles[i] for i in order], np.ones(len(x_positions), np.float32))


class data_reader_challenge:
    """Challenge batches (data_reader.py:257-319): playlists[i] = [trk, art, title_ixs,
    [has_name], pid].  Track weights: in-order files with more than 50 seeds weigh the last 15
    tracks 1.0 and the earlier ones 0.15, otherwise 1.0; artists always 0.5 (:288-291, :317)."""

    def __init__(self, data_dir, filename, batch_size):
        print("now processing: " + filename)
        with open(data_dir + '/' + filename) as f:
            d = json.load(f)
        self.playlists = d['playlists']
        self.id2uri = d['id2uri']
        self.num_tracks = d['num_tracks']
        self.num_items = d['num_items']
        self.is_in_order = d['in_order']
        self.max_title_len = d['max_title_len']
        self.num_char = d['num_char']
        self.batch_size = batch_size
        self.ch_idx = 0
        self._trk, self._trk_off = _flatten([p[0] for p in self.playlists])
        self._art, self._art_off = _flatten([p[1] for p in self.playlists])
        # the titles once more as ONE int32 array (rows padded with -1) and the has-name flags as a float array: the scoring
        # loop takes `last_titles` / `last_titles_use` of a batch instead of converting 150 Python lists per batch (0.16 ms
        # of a 0.6 ms launch); `next_batch` itself returns what the reference's returns
        L = int(self.max_title_len)
        self._titles_arr = np.full((len(self.playlists), L), -1, np.int32)
        for i, p in enumerate(self.playlists):
            if p[2]:
                self._titles_arr[i, :min(len(p[2]), L)] = p[2][:L]
        self._texist_arr = np.asarray([float(p[3][0]) for p in self.playlists], np.float32)
        self.last_titles = self.last_titles_use = None

    def next_batch(self):
        n = len(self.playlists)
        stop = min(self.ch_idx + self.batch_size, n)
        order = np.arange(self.ch_idx, stop, dtype=np.int64)
        self.last_titles = self._titles_arr[self.ch_idx:stop]
        self.last_titles_use = self._texist_arr[self.ch_idx:stop]
        self.ch_idx = 0 if stop == n else stop                   # data_reader.py:308-311
        trk, trk_rows, trk_lens = _gather(self._trk, self._trk_off, order)
        art, art_rows, _ = _gather(self._art, self._art_off, order)
        within = np.arange(trk.size, dtype=np.int64) - np.repeat(np.cumsum(trk_lens) - trk_lens, trk_lens)
        len_rep = np.repeat(trk_lens, trk_lens)
        trk_ones = np.ones(trk.size, np.float32)
        if self.is_in_order:
            early = (len_rep > 50) & (within < len_rep - 15)
            trk_ones[early] = np.float32(0.15)
        x_positions = np.concatenate((_positions(trk_rows, trk), _positions(art_rows, art)), 0)
        x_ones = np.concatenate((trk_ones, np.full(art.size, 0.5, np.float32)))
        pl = [self.playlists[i] for i in order]
        return (x_positions, [p[0] for p in pl], [p[2] for p in pl], [p[3] for p in pl],
                [p[4] for p in pl], x_ones)
