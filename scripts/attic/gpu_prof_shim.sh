cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_shim
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shim -o shim -- python $GRAFT_REPO_ROOT/scripts/bench_shim.py 256 --bf16 > /tmp/prof_shim.log 2>&1
grep recommend /tmp/prof_shim.log
f=$(find /tmp/prof_shim -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-90s calls=%6s total_ms=%10.3f avg_us=%10.2f pct=%6s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
