# usage: bash scripts/gpu_kprof.sh <tag> <n_lines> <command...>   -> gpurun_out/<tag>_kernel_stats.csv + the top kernels
tag=$1; n=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- "$@" > /tmp/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $root/gpurun_out/${tag}_kernel_stats.csv; python - "$f" $n <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print("%-86s calls=%6s avg_us=%9.2f pct=%6s" % (r["Name"][:86], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
else tail -20 /tmp/prof_$tag.log; fi
cd $root
