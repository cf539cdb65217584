# kernel timeline of a command: trace_loop.sh <tag> <command...>  -> gpurun_out/loop_<tag>.txt (per-kernel durations in the steady
# part, concurrency histogram, busy fraction per stream)
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/tl_$tag
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o out -- "$@" > /tmp/tl_$tag.log 2>&1
grep "playlists/s" /tmp/tl_$tag.log | cut -c1-120
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/loop_$tag.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(n * 0.55):int(n * 0.95)]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
span = (t1 - t0) / 1e3
dur = collections.defaultdict(list)
perq = collections.defaultdict(float)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    dur[nm].append((e - s) / 1e3)
    perq[r.get("Queue_Id", "?")] += (e - s) / 1e3
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = collections.Counter(); cur = 0; last = ev[0][0]
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
print("steady window: %d kernels, %.0f us" % (len(rows), span))
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-48s n=%4d avg %7.2f us  total %5.1f%% of kernel time, %5.1f%% of the window" % (k, len(v), sum(v) / len(v), 100 * sum(v) / tot, 100 * sum(v) / span))
print("  concurrency (fraction of the window with c kernels running):", {c: round(b / 1e3 / span, 3) for c, b in sorted(busy.items())})
print("  busy fraction per queue:", {q: round(b / span, 2) for q, b in sorted(perq.items())})
PY
