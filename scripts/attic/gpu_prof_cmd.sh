# usage: bash scripts/gpu_prof_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
script=$1; shift
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/$script "$@" > /tmp/prof_$tag.log 2>&1
echo "rocprof rc=$?"; grep -v "simple_timer\|amdgpu.ids" /tmp/prof_$tag.log | tail -3
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-80s calls=%5s avg_us=%9.2f pct=%6s" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
fi
