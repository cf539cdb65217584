#!/usr/bin/env python
"""How selective the exact title mix's filter is at the bench's titled shapes: candidates / recomputed per row of one 150-row
`recommend` call (DAE_LIB_AB: another build of the library).  usage: title_candidates.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import time_title as T     # noqa: E402

m, feed = T.build()
pos, ones, seeds, B, titles, use = feed
for k in (500, 100):
    m.recommend(pos, ones, seeds, k=k, titles=titles, titles_use=use, dtype="exact_bf16")
    print("k=%d" % k, m.title_model.ctx.exact_stats_read(), "guard", m.title_model.ctx.exact_guard_read(), "fallbacks", m.__dict__.get("_guard_fallbacks", 0))
