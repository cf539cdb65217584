#!/usr/bin/env python
"""Randomised comparison of the exact title mix (dae_mix_topk_exact) with the fp32 title path: models, scales, seed counts,
title usage, k.  usage: fuzz_title_exact.py [n_cases] [seed]"""
import os
import pathlib
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_title_exact as T          # noqa: E402


def run(n_cases=24, seed=0, log=print):
    """-> number of cases whose exact lists differ from the fp32 lists (tests/test_gpu_title_exact.py collects two seeds)."""
    rng = np.random.default_rng(seed)
    tmp = pathlib.Path(tempfile.mkdtemp())
    bad = 0
    for case in range(n_cases):
        nt = int(rng.choice([1500, 4000, 20000, 60000]))
        conf = T._conf(n_tracks=nt, n_input=nt + int(rng.integers(100, 3000)), batch=int(rng.choice([7, 24, 96, 150])))
        bias = str(rng.choice(["zipf", "zeros"]))
        w_scale = float(rng.choice([1.0, 1.0, 8.0, 40.0]))
        feat_scale = float(rng.choice([1.0, 1.0, 6.0, 25.0]))
        out_scale = float(rng.choice([1.0, 1.0, 10.0, 60.0]))
        m = T._model(tmp / ("c%d" % case), conf, bias, w_scale, title_seed=int(rng.integers(1, 1000)), feat_scale=feat_scale,
                     out_scale=out_scale) if (tmp / ("c%d" % case)).mkdir() is None else None
        if os.environ.get("FUZZ_AUDIT"):          # every launch audited (64 sampled tiles): an honest image must stay silent
            m.title_model.ctx.set_exact_audit(1, 64)
            m.ctx.set_exact_audit(1, 64)
        k = int(rng.choice([1, 10, 100, 500, 777]))
        k = min(k, 1024)
        pos, ones, seeds = T._feed(conf, int(rng.integers(0, 10000)), empty_rows=tuple(int(x) for x in rng.integers(0, conf.batch, 2)))
        titles = T._titles(conf.batch, seed=int(rng.integers(0, 10000)))
        use = (rng.random(conf.batch) < rng.choice([0.3, 0.8, 1.0])).astype(np.float32)
        if not use.any():
            use[0] = 1.0
        want = m.recommend(pos, ones, seeds, k=k, titles=titles, titles_use=use, dtype="f32")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = m.recommend(pos, ones, seeds, k=k, titles=titles, titles_use=use, dtype="exact_bf16")
        same = np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        st = m.title_model.ctx.exact_stats_read()
        log("case %2d tracks %5d batch %3d %5s x%-4g feat x%-4g out x%-4g k %3d: %s  cand %.0f recomputed %.0f fallbacks %d%s"
              % (case, nt, conf.batch, bias, w_scale, feat_scale, out_scale, k, "same" if same else "DIFFERENT",
                 st["candidates_per_row"], st["recomputed_per_row"], 1000 * getattr(m, "_guard_fallbacks", 0) + getattr(m, "_guard_row_fallbacks", 0),
                 ("  (%s)" % str(w[0].message)[30:110]) if w else ""), )
        bad += 0 if same else 1
        del m
    log("cases %d, different %d" % (n_cases, bad))
    return bad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    sys.exit(1 if run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)


if __name__ == "__main__":
    main()
