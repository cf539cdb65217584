cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 3; do
rm -rf /tmp/ptr; DAE_DBG_K6=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o tr -- python $GRAFT_REPO_ROOT/scripts/bench_train.py > /tmp/ptr.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("/tmp/ptr/tr_kernel_stats.csv")):
    if "grad_wdec" in r["Name"]: print("K6 dbg=$m avg_us=%.1f" % (float(r["AverageNs"])/1e3))
PY
done
