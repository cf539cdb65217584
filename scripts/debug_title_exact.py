import os, sys, pickle, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spotify_recsys_challenge_2018_amd import _lib as _l
if os.environ.get('DAE_LIB_AB'):
    _l.LIB_PATH = os.environ['DAE_LIB_AB']
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pathlib, tempfile
import test_gpu_title_exact as T
conf = T._conf()
tmp = pathlib.Path(tempfile.mkdtemp())
m = T._model(tmp, conf)
pos, ones, seeds = T._feed(conf, 5, empty_rows=(2, 11))
titles = T._titles(conf.batch, seed=6)
use = (np.arange(conf.batch) % 3 != 0).astype(np.float32); use[2] = 1.0
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
wi, ws = m.recommend(pos, ones, seeds, k=k, titles=titles, titles_use=use, dtype="f32")
gi, gs = m.recommend(pos, ones, seeds, k=k, titles=titles, titles_use=use, dtype="exact_bf16")
print("stats", m.title_model.ctx.exact_stats_read(), "fallbacks", getattr(m, "_guard_fallbacks", 0))
import torch
tc = m.title_model.ctx
for r in range(conf.batch):
    same = np.array_equal(wi[r], gi[r])
    nvalid = int((gi[r] >= 0).sum())
    miss = sorted(set(wi[r].tolist()) - set(gi[r].tolist()))[:8]
    print(r, "ok" if same else "DIFF", "valid", nvalid, "nseeds", len(seeds[r]), "use", use[r], "missing", miss, "kth score", ws[r][-1])
# only the rows with 100 seeds, replicated
rows100 = [r for r in range(conf.batch) if len(seeds[r]) == 100]
r0 = rows100[0]
sel = pos[pos[:, 0] == r0]
pos2 = np.concatenate([np.stack([np.full(len(sel), r), sel[:, 1]], 1) for r in range(conf.batch)])
seeds2 = [seeds[r0]] * conf.batch
titles2 = np.stack([titles[r0]] * conf.batch)
use2 = np.full(conf.batch, use[r0], np.float32)
wi, ws = m.recommend(pos2, 1.0 if np.ndim(ones) == 0 else np.ones(len(pos2), np.float32), seeds2, k=k, titles=titles2, titles_use=use2, dtype="f32")
gi, gs = m.recommend(pos2, 1.0 if np.ndim(ones) == 0 else np.ones(len(pos2), np.float32), seeds2, k=k, titles=titles2, titles_use=use2, dtype="exact_bf16")
print("row", r0, "replicated: stats", m.title_model.ctx.exact_stats_read(), "valid", int((gi[0] >= 0).sum()), "same", np.array_equal(wi, gi))
