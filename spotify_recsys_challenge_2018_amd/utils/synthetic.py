"""Synthetic weights and playlists of the shape BASELINE.md section 4 / SURVEY.md 8(d) prescribe
(no MPD data or trained weights exist on the box).  Shared by bench.py, smoke() and the tests."""
import numpy as np


def make_weights(V, H, seed=0, bias="zeros", n_tracks=None, tied=False):
    """Xavier-uniform W_enc/W_dec ~ U(+-sqrt(6/(V+H))) (reference DAEs.py:54-55), b_enc = 0,
    b_dec zeros (init) or log-odds of a Zipf popularity (trained-like: ids ARE popularity ranks,
    spotify_reader.py:63-64,143)."""
    rng = np.random.default_rng(seed)
    lim = np.sqrt(6.0 / (V + H))
    W_enc = rng.uniform(-lim, lim, size=(V, H)).astype(np.float32)
    W_dec = W_enc if tied else rng.uniform(-lim, lim, size=(V, H)).astype(np.float32)
    b_enc = np.zeros(H, np.float32)
    if bias == "zeros":
        b_dec = np.zeros(V, np.float32)
    else:
        nt = V if n_tracks is None else n_tracks
        rank = np.concatenate([np.arange(nt), np.arange(V - nt)]).astype(np.float64) + 1.0
        p = np.minimum(0.5, 60.0 / rank / np.log(V))
        b_dec = np.log(p / (1.0 - p)).astype(np.float32)
    return W_enc, b_enc, W_dec, b_dec


def _zipf_ids(rng, n, count):
    u = rng.random(count)
    return np.minimum(n - 1, np.floor(np.exp(u * np.log(n))).astype(np.int64) - 1).clip(0)


def make_playlists(B, n_tracks, n_artists, seed=1, dist="zipf", seed_counts=(1, 5, 10, 25, 100)):
    """Challenge-shaped batch (data_reader.py:275-319): per row `c` seed tracks + `c` seed artists,
    c cycling over the challenge categories (readme.md:69); track weight 1.0, artist 0.5
    (data_reader.py:317).  Duplicate ids are LEFT IN the COO feed, as in the real data.
    Returns (x_positions int64 [nnz,2], x_ones float32 [nnz], seeds list[list[int]])."""
    rng = np.random.default_rng(seed)
    trk_pos, art_pos, seeds = [], [], []
    for r in range(B):
        c = seed_counts[r % len(seed_counts)]
        if dist == "zipf":
            t = _zipf_ids(rng, n_tracks, c)
            a = _zipf_ids(rng, max(n_artists, 1), c) if n_artists > 0 else np.zeros(0, np.int64)
        else:
            t = rng.integers(0, n_tracks, size=c)
            a = rng.integers(0, max(n_artists, 1), size=c) if n_artists > 0 else np.zeros(0, np.int64)
        seeds.append([int(x) for x in t])
        trk_pos.append(np.stack([np.full(t.size, r, np.int64), t], axis=1))
        if n_artists > 0:
            art_pos.append(np.stack([np.full(a.size, r, np.int64), a + n_tracks], axis=1))
    trk = np.concatenate(trk_pos) if trk_pos else np.zeros((0, 2), np.int64)
    art = np.concatenate(art_pos) if art_pos else np.zeros((0, 2), np.int64)
    x_positions = np.concatenate([trk, art], axis=0)
    x_ones = np.concatenate([np.ones(len(trk), np.float32), np.full(len(art), 0.5, np.float32)])
    return x_positions, x_ones, seeds
