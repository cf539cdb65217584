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


# ---- a TRAINED model without MPD data: clustered playlists ------------------------------------------------------------
# The Xavier + Zipf-bias model above is popularity-dominated: every playlist of a batch ranks nearly the same tracks first.
# A model trained on playlists with structure ranks them differently per playlist, its decoder rows have very
# different norms, and its bias is the learnt popularity prior -- the case the exact-bf16 mode has to survive
# (VERDICT r3 item 1c).  Tracks belong to `n_clusters` "genres" (a fixed pseudo-random assignment); a playlist draws most
# of its tracks from one or two clusters, Zipf by popularity inside the cluster (ids ARE popularity ranks,
# spotify_reader.py:63-64,143), the rest from the global Zipf; the artist of track t is t's block of the artist range.
class ClusteredPlaylists:
    def __init__(self, n_tracks, n_artists, n_clusters=256, seed=0, global_frac=0.15):
        self.n_tracks, self.n_artists, self.C, self.global_frac = int(n_tracks), int(n_artists), int(n_clusters), global_frac
        t = np.arange(self.n_tracks, dtype=np.uint64)
        h = (t * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed * 1315423911 + 12345)) >> np.uint64(29)
        self.cluster_of = (h % np.uint64(self.C)).astype(np.int64)
        order = np.argsort(self.cluster_of, kind="stable")                 # tracks grouped by cluster, popularity order kept
        sizes = np.bincount(self.cluster_of, minlength=self.C)
        self.sizes = sizes.astype(np.int64)
        self.start = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.members = order.astype(np.int64)

    def artist_of(self, tracks):
        return (tracks * self.n_artists) // max(self.n_tracks, 1)

    def _draw_matrix(self, B, rng, min_len=12, max_len=100):
        """-> (trk [B, Lm] int64, L [B]): row r's playlist is trk[r, :L[r]] (duplicates possible, as in real feeds)."""
        L = rng.integers(min_len, max_len + 1, size=B)
        Lm = int(L.max())
        c1 = rng.integers(0, self.C, size=B)
        c2 = np.where(rng.random(B) < 0.3, rng.integers(0, self.C, size=B), c1)
        cl = np.where(rng.random((B, Lm)) < 0.5, c1[:, None], c2[:, None])
        n_c = self.sizes[cl]
        rank_in = np.minimum(n_c - 1, np.floor(np.exp(rng.random((B, Lm)) * np.log(np.maximum(n_c, 2)))).astype(np.int64) - 1).clip(0)
        trk = self.members[self.start[cl] + rank_in]
        glob = rng.random((B, Lm)) < self.global_frac
        g_ids = np.minimum(self.n_tracks - 1,
                           np.floor(np.exp(rng.random((B, Lm)) * np.log(self.n_tracks))).astype(np.int64) - 1).clip(0)
        return np.where(glob, g_ids, trk), L

    def draw(self, B, rng, min_len=12, max_len=100):
        """-> list of B int64 arrays: the tracks of each playlist, in playlist order, duplicates removed."""
        trk, L = self._draw_matrix(B, rng, min_len, max_len)
        out = []
        for r in range(B):
            row = trk[r, : L[r]]
            _, first = np.unique(row, return_index=True)
            out.append(row[np.sort(first)])
        return out

    def training_feed(self, B, rng, seed_counts=(1, 5, 10, 25, 100)):
        """(x_positions, x_ones, y_positions, y_ones): input = the first n tracks (+ their artists, 0.5) of each playlist
        (firstN, data_reader.py:73-128 shape), target = every track and artist of the playlist (ones).  Duplicate COO
        entries are left in, as in the real feeds (the CSR build keeps the last)."""
        trk, L = self._draw_matrix(B, rng)
        n = np.minimum(np.asarray(seed_counts)[rng.integers(0, len(seed_counts), size=B)], np.maximum(1, L - 1))
        ar = np.arange(trk.shape[1])[None, :]
        rows = np.broadcast_to(np.arange(B)[:, None], trk.shape)

        def coo(mask, w_art):
            r_, t_ = rows[mask], trk[mask]
            if self.n_artists <= 0:
                return np.stack([r_, t_], 1), np.ones(t_.size, np.float32)
            a_ = self.artist_of(t_) + self.n_tracks
            return (np.concatenate([np.stack([r_, t_], 1), np.stack([r_, a_], 1)]),
                    np.concatenate([np.ones(t_.size, np.float32), np.full(a_.size, w_art, np.float32)]))
        xp, xo = coo(ar < n[:, None], 0.5)
        yp, yo = coo(ar < L[:, None], 1.0)
        return xp, xo, yp, yo

    def scoring_feed(self, B, rng, seed_counts=(1, 5, 10, 25, 100)):
        """Challenge-shaped batch as make_playlists: (x_positions, x_ones, seeds)."""
        lists = self.draw(B, rng)
        xp, xo, seeds = [], [], []
        for r, row in enumerate(lists):
            n = min(int(seed_counts[r % len(seed_counts)]), len(row))
            s = row[:n]
            a_in = (np.unique(self.artist_of(s)) + self.n_tracks) if self.n_artists > 0 else np.zeros(0, np.int64)
            xc = np.concatenate([s, a_in])
            xp.append(np.stack([np.full(xc.size, r, np.int64), xc], 1))
            xo.append(np.concatenate([np.ones(s.size, np.float32), np.full(a_in.size, 0.5, np.float32)]))
            seeds.append([int(x) for x in s])
        return np.concatenate(xp), np.concatenate(xo), seeds


def train_clustered_model(n_tracks, n_artists, H, steps=2000, batch=256, seed=0, n_clusters=256, lr=0.005,
                          keep_prob=0.8, input_keep_prob=0.75, train_dtype="bf16", device_index=0, log=None):
    """Trains the untied DAE (models/DAEs.py DAE.train_step: the library's own training step, DAEs.py:98-102 of the
    reference) on clustered synthetic playlists, on the GPU.  Returns (W_enc, b_enc, W_dec, b_dec, ClusteredPlaylists,
    info).  Needs the HIP library and a device: there is no CPU fallback."""
    import time
    from ..models.DAEs import DAE

    class C:
        pass
    conf = C()
    V = n_tracks + n_artists
    conf.save = "/tmp/_unused_trained"; conf.batch = batch; conf.n_input = V; conf.hidden = H; conf.lr = lr
    conf.reg_lambda = 0.0; conf.initval = "NULL"; conf.n_tracks = n_tracks; conf.train_dtype = train_dtype
    conf.device_index = device_index; conf.init_seed = seed
    gen = ClusteredPlaylists(n_tracks, n_artists, n_clusters=n_clusters, seed=seed)
    rng = np.random.default_rng(1000 + seed)
    m = DAE(conf)
    m.fit()
    t0 = time.perf_counter()
    costs = []
    for s in range(steps):
        xp, xo, yp, yo = gen.training_feed(batch, rng)
        c = m.train_step(xp, xo, yp, yo, keep_prob, input_keep_prob, fetch_cost=(s % 200 == 0 or s == steps - 1))
        if isinstance(c, float):
            costs.append(round(c, 3))
            if log:
                log("step %d cost %.3f" % (s, c))
    W_enc, W_dec, b_enc, b_dec = m.get_params()
    info = {"steps": steps, "batch": batch, "train_s": round(time.perf_counter() - t0, 2), "costs": costs,
            "n_clusters": n_clusters, "train_dtype": train_dtype}
    m.ctx.close()
    return W_enc, b_enc, W_dec, b_dec, gen, info
