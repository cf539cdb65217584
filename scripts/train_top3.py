#!/usr/bin/env python
"""The training step's three largest kernels against their own roofs, from the rocprofv3 kernel stats of
scripts/gpu_round6.sh (profiles/r06_train_default_kernel_stats.csv, and the f32 / bf16 ones when present) -> profiles/r06_train_top3.json, which
bench.py attaches to its `training_step` rows.  Algorithmic work per launch at B = 256, V = 170 000, H = 256 (DESIGN.md
section 4 "Training"): every GEMM 2 B V H FLOP; bytes = what no schedule avoids (W read, dL/dz written / read, Adam state)."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, V, H = 256, 170000, 256
PEAK_F32, PEAK_BF16, PEAK_HBM = 157.3e12, 2.5e15, 8.0e12
mat = 4.0 * V * H
gemm = 2.0 * B * V * H


def work(kernel, bf16, fused):
    """-> (flop, bytes) of one launch, or None for kernels that are not rated."""
    dz = (2.0 if bf16 else 4.0) * B * V
    if "decode_loss_dh" in kernel:          # K5 + K7 in one launch: two GEMMs; W once, dz^T written, one dh partial per workgroup
        return 2 * gemm, mat + dz + 256 * 4.0 * B * H
    if "decode_loss_" in kernel:
        return gemm, mat + dz
    if "grad_wdec" in kernel:
        return gemm, dz + (6 * mat if fused else mat)
    if "grad_hidden" in kernel:
        return gemm, dz + mat
    if kernel.startswith("adam_kernel"):
        return 0.0, 7 * mat                      # the large launches (W_enc, W_dec); the two bias launches are noise in the average
    if "adam_rows_flush" in kernel:
        return 0.0, 6 * mat
    return None


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


out = {"shape": [B, V, H], "source": "rocprofv3 --kernel-trace --stats of scripts/bench_train.py [--bf16 | --default] (scripts/gpu_round6.sh): "
       "profiles/r06_train_*_kernel_stats.csv; frac = the larger of (FLOP / MFMA peak of the operand type) and "
       "(bytes / 8 TB/s), over the measured average",
       "train_hip_sha256_16": hashlib.sha256(open(os.path.join(ROOT, "spotify_recsys_challenge_2018_amd", "csrc", "train.hip"), "rb").read()).hexdigest()[:16],
       "decode_hip_sha256_16": hashlib.sha256(open(os.path.join(ROOT, "spotify_recsys_challenge_2018_amd", "csrc", "decode_f32.hip"), "rb").read()).hexdigest()[:16]}
for key, fname, bf16, fused in (("f32", "r06_train_f32_kernel_stats.csv", False, False), ("bf16_gemms", "r06_train_bf16_kernel_stats.csv", True, False),
                                ("model_default_bf16", "r06_train_default_kernel_stats.csv", True, True)):
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        continue
    rows = []
    for r in csv.DictReader(open(path)):
        name = short(r["Name"])
        w = work(name, bf16, fused)
        if w is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        if name.startswith("adam_kernel"):
            # four launches per step, two of them on 174 MB matrices: the average is dominated by those (bias launches ~5 us)
            us = (us * 4 - 2 * 5.0) / 2 if not fused else us
            if fused:
                continue
        t_mfma = w[0] / (PEAK_BF16 if bf16 else PEAK_F32) * 1e6
        t_hbm = w[1] / PEAK_HBM * 1e6
        steps = 35.0 if fused else 23.0                  # launches of the step function in that run (warm-up included)
        per_step = us * (2 if name.startswith("adam_kernel") else int(r["Calls"]) / steps)
        rows.append({"kernel": name, "avg_us": round(us, 1), "per_step_us": round(per_step, 1), "bound": "hbm" if t_hbm > t_mfma else "mfma",
                     "t_mfma_us": round(t_mfma, 1), "t_hbm_us": round(t_hbm, 1), "frac": round(max(t_mfma, t_hbm) / us, 3)})
    rows.sort(key=lambda x: -x["per_step_us"])
    out[key] = rows[:3]
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_train_top3.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
