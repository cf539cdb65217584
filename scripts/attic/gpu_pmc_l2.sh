# usage: bash scripts/gpu_pmc_l2.sh <tag> <bench args...>   L2 / HBM-side counters of the decode kernels, one pass per set
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pl_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > /tmp/pl_$i.log 2>&1
  f=$(find /tmp/pl_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_l2_${tag}_pmc_$i.csv || tail -5 /tmp/pl_$i.log
done
python - $tag <<'PY'
import csv, glob, os, collections, sys
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for f in sorted(glob.glob(root + "/r01_l2_%s_pmc_*.csv" % sys.argv[1])):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "decode" in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
