# usage: bash scripts/gpu_prof.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/prof_$tag.log 2>&1
echo "rocprof rc=$?"; grep -v "simple_timer\|amdgpu.ids" /tmp/prof_$tag.log | tail -15
find /tmp/prof_$tag -type f | head -20
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls=%6s total_ms=%10.3f avg_us=%10.2f pct=%6s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
fi
