# kernel timeline of the drivers' loop: trace_iter.sh <tag> [env assignments...] -> summary
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
env "$@" timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_$tag -o out -- python $GRAFT_REPO_ROOT/scripts/attic/prof_iter_host.py 256 ${MODE:-f32} > /tmp/tr_$tag.log 2>&1
grep "playlists/s" /tmp/tr_$tag.log | tail -2
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/tr_$tag -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$m" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); rows = rows[int(n * 0.5):int(n * 0.9)]
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
dur = collections.defaultdict(list)
fbusy = []
for r in rows:
    nm = r["Kernel_Name"]
    key = nm.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if True:
        fbusy.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
fbusy.sort(); tot = 0; cur_s, cur_e = fbusy[0]
for s, e in fbusy[1:]:
    if s > cur_e:
        tot += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
tot += cur_e - cur_s
print("window %.1f ms; ANY kernel running %.3f of it" % ((t1 - t0) / 1e6, tot / (t1 - t0)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  %-42s n=%4d avg %8.2f us total %8.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
if len(sys.argv) > 2 and sys.argv[2]:
    mc = list(csv.DictReader(open(sys.argv[2])))
    d = collections.defaultdict(list)
    for r in mc:
        d[r.get("Direction", r.get("Kind", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in d.items():
        print("  copies %-28s n=%4d avg %8.2f us" % (k, len(v), sum(v) / len(v)))
PY
