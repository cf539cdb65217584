"""Host-side profile of the titled drivers' loop (cProfile): python scripts/prof_title_host.py <mode>"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import time_title as T
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "exact_bf16"
m, feed = T.build()
for _ in m.recommend_iter([feed] * 10, k=500, dtype=mode, want_scores=False):
    pass
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in m.recommend_iter([feed] * 200, k=500, dtype=mode, want_scores=False):
    pass
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(40)
