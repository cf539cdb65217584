"""Diagnosis: bench.py's drivers_loop row against scripts/bench_loop.py (same loop, different rates).  usage: row_diag.py <modes csv> [settle 0/1] [copy_first 0/1]"""
import os, sys, time, pickle, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from spotify_recsys_challenge_2018_amd import _lib as _l
if os.environ.get('DAE_LIB_AB'):
    _l.LIB_PATH = os.environ['DAE_LIB_AB']
from spotify_recsys_challenge_2018_amd.models.DAEs import DAE, SEEDS_FROM_INPUT
from spotify_recsys_challenge_2018_amd.utils.synthetic import make_playlists, make_weights
nt, na, H, k, B = 140000, 30000, 256, 500, 256
V = nt + na
W_enc, b_enc, W_dec, b_dec = make_weights(V, H, seed=0, bias="zipf", n_tracks=nt)
modes = sys.argv[1].split(",")
settle = len(sys.argv) < 3 or sys.argv[2] == "1"
copy_first = len(sys.argv) < 4 or sys.argv[3] == "1"
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "init.pkl")
with open(path, "wb") as f:
    pickle.dump([W_enc, W_dec, b_enc, b_dec], f)
class C:
    save = os.path.join(tmp, "unused"); batch = B; n_input = V; hidden = H; lr = 0.005; reg_lambda = 0.0
    initval = path
C.n_tracks = nt
m = DAE(C()); m.fit()
if os.environ.get('KEEP_PIPES'):
    m.keep_pipelines = True
batches = [make_playlists(B, nt, na, seed=200 + s_, dist="zipf")[:2] for s_ in range(8)]
def feeds(reps):
    for _ in range(reps):
        for p_, o_ in batches:
            yield p_, o_, SEEDS_FROM_INPUT, B
first = {}
for name in modes:
    reps, warm = (150, 60) if name == "f32" else (500, 200)
    for i_, (idx_, _s) in enumerate(m.recommend_iter(feeds(warm), k=k, want_scores=False, dtype=name)):
        if i_ == 0 and copy_first:
            first[name] = idx_.copy()
    torch.cuda.synchronize()
    if settle:
        bench._settle_interpreter()
    t0 = time.perf_counter(); n = 0
    for _idx, _s in m.recommend_iter(feeds(reps), k=k, want_scores=False, dtype=name):
        n += B
    el = time.perf_counter() - t0
    if os.environ.get("RELEASE_VIEWS"):       # nothing views the pipeline's pinned blocks any more: the next mode's pipeline reuses its streams
        _idx = _s = idx_ = None
        import gc; gc.collect()
    extra = ""
    for pp in m.__dict__.get("_pipes", {}).values():
        extra = "  %s %s total_ms=%.1f" % (pp[1].times(), pp[1].stats(), el * 1e3)
    print("%s settle=%d copy=%d modes=%s: %.0f playlists/s%s" % (name, settle, copy_first, sys.argv[1], n / el, extra), flush=True)
