#!/usr/bin/env python
"""Prints the rows of bench.py JSON lines (files given as arguments) in a few lines each."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:
        print(path, "FAILED", repr(e))
        continue
    r = d["roofline"]
    print("%s: value=%.0f ms=%.4f kernel=%s %.4f frac=%.3f step=%s isolated=%s traffic=%s" % (
        path.split("/")[-1], d["value"], d["ms_per_step"], r["kernel"][:34], r["avg_launch_ms"], r["frac"],
        (r.get("step_level") or {}).get("frac"), (r.get("isolated") or {}).get("frac"), r.get("traffic")))
    print("   rows", {k: (v.get("value"), v.get("ms_per_step")) for k, v in d.items() if isinstance(v, dict) and "value" in v and k != "roofline"})
    for k in ("exact_bf16_hard", "trained_model"):
        for kk, v in (d.get(k) or {}).items():
            if isinstance(v, dict) and ("f32" in v or "value" in v):
                print("   ", k, kk, {a: b.get("value") for a, b in v.items() if isinstance(b, dict) and "value" in b} or v.get("value"))
    if "drivers_loop" in d:
        print("    loop", {a: b.get("value") for a, b in d["drivers_loop"].items() if isinstance(b, dict) and "value" in b})
    if "titled" in d:
        t = d["titled"]
        print("    titled", {a: b.get("value") for a, b in t.items() if isinstance(b, dict) and "value" in b} or t,
              {a: b for a, b in (t.get("exact_bf16") or {}).items() if a != "value" and a != "feeds"})
    if "training_step" in d:
        print("    train", {a: b.get("ms_per_step") for a, b in d["training_step"].items() if isinstance(b, dict)})
    if "phases" in d:
        print("    phases", {a: b for a, b in d["phases"].items() if a.endswith("_ms")})
    for k in ("bf16_decode", "exact_bf16_decode", "exact_b1024"):
        if k in d and "roofline" in d[k]:
            print("   ", k, "isolated", d[k]["roofline"]["isolated"].get("avg_launch_ms"), d[k]["roofline"]["isolated"].get("hbm_frac"),
                  d[k]["roofline"]["isolated"].get("mfma_frac"))
