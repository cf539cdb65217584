mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
d=/tmp/pmc6_train_$c; rm -rf $d
timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --default > $d.log 2>&1
f=$(find $d -name "*counter_collection.csv" | head -1)
python - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print("%-62s launches %3d  %s avg %.0f KB" % (k, len(v), sys.argv[2], sum(v) / len(v)))
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_train_quick.txt
