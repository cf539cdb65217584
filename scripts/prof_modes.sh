# rocprofv3 kernel stats of scripts/time_modes.py for one mode: prof_modes.sh <B> <bias> <mode> <tag>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$4; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$4 -o out -- python $GRAFT_REPO_ROOT/scripts/time_modes.py $1 $2 $3 > /tmp/prof_$4.log 2>&1
grep -v 'simple_timer\|amdgpu.ids' /tmp/prof_$4.log | tail -3 | cut -c1-100
f=$(find /tmp/prof_$4 -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $f $GRAFT_REPO_ROOT/gpurun_out/stats_$4.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-90s calls=%6s avg_us=%9.2f pct=%5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
