# HBM-side traffic of the dominant kernel: FETCH_SIZE and WRITE_SIZE, one rocprofv3 --pmc pass each
# usage: bash scripts/gpu_pmc_traffic.sh <tag> [bench args]  -> gpurun_out/<tag>_pmc_{1,2}.csv
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pt_${tag}_$i
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pt_${tag}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --streams 1 "$@" > /tmp/pt_${tag}_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
  f=$(find /tmp/pt_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$i.csv
done
python - $tag <<'PY'
import csv, glob, os, collections, sys
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for f in sorted(glob.glob(root + "/%s_pmc_*.csv" % sys.argv[1])):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "decode" in k or "encode" in k:
            print("  %-70s %s" % (k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()}))
PY
