#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc_round6.sh into the two JSON files bench.py quotes
(`roofline.traffic`, `roofline_encode.traffic`): HBM-side bytes per launch = FETCH_SIZE x 2 (gfx950 correction for
wide coalesced reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, counters in KB."""
import collections
import csv
import json
import os
import sys

root = sys.argv[1]
PFX = sys.argv[2] if len(sys.argv) > 2 else "r03"        # file prefix of the passes: <PFX>_pmc_<cfg>_<set>.csv


def per_kernel(path):
    """kernel name -> counter -> list of per-dispatch values, dispatch order kept."""
    out = collections.OrderedDict()
    if not os.path.exists(path):
        return out
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        out.setdefault(r["Kernel_Name"], collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(
            float(r["Counter_Value"]))
    return out


def short_name(k):
    """'void (anonymous namespace)::foo<1, 2>((anonymous namespace)::P)' -> 'foo<1, 2>' (what bench.py reports)."""
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0].strip()


def mean(v):
    return sum(v) / max(len(v), 1)


def find(d, sub):
    for k in d:
        if sub in k:
            return k, d[k]
    return None, {}


def src_sha(name):
    """sha256[:16] of a kernel source file as it is on the box that ran the passes: bench.py quotes a counter figure only
    while the file it was measured on is the file in the tree (no hand-stamped commit ids)."""
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        return hashlib.sha256(open(os.path.join(here, "spotify_recsys_challenge_2018_amd", "csrc", name), "rb").read()).hexdigest()[:16]
    except Exception:
        return None


sha = {n: src_sha(n) for n in ("decode_f32.hip", "refine.hip", "encode.hip", "train.hip", "dae_internal.h")}
dec = {"source": "rocprofv3 --kernel-trace --pmc <set>, one pass per counter set (scripts/gpu_pmc_round6.sh): bench.py "
                 "--streams 1 --steps 6, default workload (B=256, V=170000, H=256); raw CSVs profiles/%s_pmc_*.csv" % PFX,
       "fetch_correction": "hbm_bytes_per_launch = FETCH_SIZE x 2 (gfx950: wide coalesced reads are tallied at half) + "
                           "WRITE_SIZE; counters are KB", "kernel_source_sha256_16": sha}
enc = {"source": dec["source"], "fetch_correction": dec["fetch_correction"], "kernel_source_sha256_16": sha}
for cfg, ksub, peak_mops in (("f32", "decode_f32_h256_filter_kernel", "SQ_INSTS_VALU_MFMA_MOPS_F32"),
                             ("bf16", "decode_bf16_h256_filter_kernel", "SQ_INSTS_VALU_MFMA_MOPS_BF16")):
    f = per_kernel(os.path.join(root, "%s_pmc_%s_fetch.csv" % (PFX, cfg)))
    w = per_kernel(os.path.join(root, "%s_pmc_%s_write.csv" % (PFX, cfg)))
    m = per_kernel(os.path.join(root, "%s_pmc_%s_mfma.csv" % (PFX, cfg)))
    kname, fd = find(f, ksub)
    _, wd = find(w, ksub)
    _, md = find(m, ksub)
    if not kname:
        continue
    short = short_name(kname)
    fetch, write = mean(fd.get("FETCH_SIZE", [0])), mean(wd.get("WRITE_SIZE", [0]))
    row = {"kernel": short, "launches": len(fd.get("FETCH_SIZE", [])), "fetch_size_kb": round(fetch, 1),
           "write_size_kb": round(write, 1), "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    if md:
        busy, gui = mean(md.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), mean(md.get("GRBM_GUI_ACTIVE", [1]))
        row["mfma"] = {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, peak_mops: mean(md.get(peak_mops, [0])),
                       "SQ_BUSY_CU_CYCLES": mean(md.get("SQ_BUSY_CU_CYCLES", [0])),
                       # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
                       "mfma_busy_frac_of_kernel": round(busy / 1024.0 / max(gui / 8.0, 1.0), 4)}
    dec[cfg] = row
    if cfg == "f32":
        # encode rows of the same run (bench.py times K1 after the step loop): the step's batch, then 8192 playlists
        # with Zipf ids, then 8192 with uniform ids (same kernel symbol: told apart by dispatch order)
        for key, sub, sl in (("step_batch", "encode_split_kernel", None), ("large_batch", "encode_kernel<4>", slice(0, 7)),
                             ("uniform_ids", "encode_kernel<4>", slice(7, 14))):
            kn, fe = find(f, sub)
            _, we = find(w, sub)
            if not kn:
                continue
            fv, wv = fe.get("FETCH_SIZE", []), we.get("WRITE_SIZE", [])
            if sl is not None:
                fv, wv = fv[sl], wv[sl]
            else:
                fv, wv = fv[-20:], wv[-20:]              # the 20 launches of the K1 timing loop
            if not fv:
                continue
            enc[key] = {"kernel": short_name(kn),
                        "launches": len(fv), "fetch_size_kb": round(mean(fv), 1), "write_size_kb": round(mean(wv), 1),
                        "hbm_bytes_per_launch": int((2 * mean(fv) + mean(wv)) * 1024)}
# the exact mode's refine launch (DAE_DTYPE_BF16_EXACT): bytes its row gathers really pull through the fabric
f = per_kernel(os.path.join(root, "%s_pmc_exact_fetch.csv" % PFX))
w = per_kernel(os.path.join(root, "%s_pmc_exact_write.csv" % PFX))
for key, sub in (("exact_refine", "exact_refine"), ("exact_filter", "decode_bf16_h256_filter_kernel")):
    kn, fd = find(f, sub)
    _, wd = find(w, sub)
    if kn:
        fetch, write = mean(fd.get("FETCH_SIZE", [0])), mean(wd.get("WRITE_SIZE", [0]))
        dec[key] = {"kernel": short_name(kn), "launches": len(fd.get("FETCH_SIZE", [])), "fetch_size_kb": round(fetch, 1),
                    "write_size_kb": round(write, 1), "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
# the training step's HBM-bound launches (scripts/bench_train.py --default under --pmc)
f = per_kernel(os.path.join(root, "%s_pmc_train_fetch.csv" % PFX))
w = per_kernel(os.path.join(root, "%s_pmc_train_write.csv" % PFX))
trn = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python scripts/bench_train.py --default",
       "fetch_correction": dec["fetch_correction"], "kernel_source_sha256_16": sha}
for key, sub in (("grad_wdec_adam", "grad_wdec_t"), ("loss_forward", "decode_loss_"),
                 ("grad_hidden", "grad_hidden_kernel"), ("adam_rows_apply", "adam_rows_kernel<1>"), ("adam_rows_begin", "adam_rows_kernel<0>")):
    kn, fd = find(f, sub)
    _, wd = find(w, sub)
    if kn:
        fetch, write = mean(fd.get("FETCH_SIZE", [0])), mean(wd.get("WRITE_SIZE", [0]))
        trn[key] = {"kernel": short_name(kn), "launches": len(fd.get("FETCH_SIZE", [])), "fetch_size_kb": round(fetch, 1),
                    "write_size_kb": round(write, 1), "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
if len(trn) > 3:
    json.dump(trn, open(os.path.join(root, "traffic_train.json"), "w"), indent=1)
    print(json.dumps(trn, indent=1))
# the exact title mix (scripts/time_title.py exact_bf16 under --pmc): 750 rows per launch, 140 000 rankable columns
f = per_kernel(os.path.join(root, "%s_pmc_title_fetch.csv" % PFX))
w = per_kernel(os.path.join(root, "%s_pmc_title_write.csv" % PFX))
ttl = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python scripts/time_title.py exact_bf16 20",
       "fetch_correction": dec["fetch_correction"], "kernel_source_sha256_16": dict(sha, **{"mixexact.hip": src_sha("mixexact.hip")})}
for key, sub in (("mix_filter", "mix_bf16_kernel<16, 28, 3, 8, 8, 0>"), ("mix_sample", "mix_bf16_kernel<16, 28, 3, 8, 8, 1>"),
                 ("mix_refine", "mix_refine_kernel"), ("title_features", "title_features_mfma_kernel")):
    kn, fd = find(f, sub)
    _, wd = find(w, sub)
    if kn:
        fetch, write = mean(fd.get("FETCH_SIZE", [0])), mean(wd.get("WRITE_SIZE", [0]))
        ttl[key] = {"kernel": short_name(kn), "launches": len(fd.get("FETCH_SIZE", [])), "fetch_size_kb": round(fetch, 1),
                    "write_size_kb": round(write, 1), "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
if len(ttl) > 3:
    json.dump(ttl, open(os.path.join(root, "traffic_title.json"), "w"), indent=1)
    print(json.dumps(ttl, indent=1))
json.dump(dec, open(os.path.join(root, "traffic_decode.json"), "w"), indent=1)
json.dump(enc, open(os.path.join(root, "traffic_encode.json"), "w"), indent=1)
print(json.dumps(dec, indent=1))
print(json.dumps(enc, indent=1))
