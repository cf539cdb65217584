# kernel timeline of scripts/time_modes.py: trace_modes.sh <B> <bias> <mode> <streams> <tag>  -> gpurun_out/trace_<tag>.csv (+ summary)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$5 -o out -- python $GRAFT_REPO_ROOT/scripts/time_modes.py $1 $2 $3 $4 > /tmp/tr_$5.log 2>&1
grep -v 'simple_timer\|amdgpu.ids\|output_stream' /tmp/tr_$5.log | tail -2 | cut -c1-100
f=$(find /tmp/tr_$5 -name "*kernel_trace.csv" | head -1)
python - "$f" "$GRAFT_REPO_ROOT/gpurun_out/trace_$5.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state: the last 40 % of the launches
n = len(rows)
rows = rows[int(n * 0.6):]
t0 = int(rows[0]["Start_Timestamp"])
def short(nm):
    for key, s in (("exact_refine", "X refine"), ("h256_filter", "F filter"), ("encode", "E encode"), ("tau_select", "T tau"), ("topk_kernel", "K topk"), ("decode_f32_kernel", "A sample")):
        if key in nm:
            return s
    return nm[:30]
dur = collections.defaultdict(list)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    dur[short(r["Kernel_Name"])].append((e - s) / 1e3)
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
span = ev[-1][0] - ev[0][0]
print("kernels in steady window: %d, span %.1f us" % (len(rows), span / 1e3))
for k, v in sorted(dur.items()):
    print("  %-10s n=%4d avg %.2f us  (min %.2f max %.2f)" % (k, len(v), sum(v) / len(v), min(v), max(v)))
print("  concurrency histogram (fraction of time with c kernels running):", {c: round(b / span, 3) for c, b in sorted(busy.items())})
with open(sys.argv[2], "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "queue", "start_us", "end_us"])
    for r in rows[:400]:
        w.writerow([short(r["Kernel_Name"]), r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3])
PY
