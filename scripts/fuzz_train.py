#!/usr/bin/env python
"""Randomised shapes through the training step's bf16 path at hidden 256 (K5 + K7 fused, K6 with Adam in its epilogue is not
armed here): tests/test_gpu_train.py::test_train_step_bf16_gemms' comparison with the fp32 step on the same draws (cost within
3e-3, gradients within 2e-2 of their norm), and ::test_train_step_gradients' comparison of the fp32 step with the float64
reference.  usage: fuzz_train.py [n_cases] [seed]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    V = int(rng.choice([rng.integers(300, 3000), rng.integers(3000, 30000), rng.integers(30000, 90000)]))
    nt = int(rng.integers(V // 2, V))
    B = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 128, 200, 255, 256]))
    tied = bool(rng.integers(0, 2))
    try:
        T.test_train_step_bf16_gemms(V, nt, 256, B, tied)
        if V < 20000:
            T.test_train_step_gradients(V, nt, 256, B, tied, 0.0, 0.75, 0.8)
        print("case %d V=%d nt=%d B=%d tied=%s: ok" % (case, V, nt, B, tied), flush=True)
    except AssertionError as e:
        bad += 1
        print("case %d V=%d nt=%d B=%d tied=%s: FAILED %r" % (case, V, nt, B, tied, e), flush=True)
print("fuzz_train: %d cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
