# usage: bash scripts/gpu_pmc.sh <tag> -> gpurun_out/<tag>_pmc_*.csv  (one rocprofv3 pass per counter set)
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$tag_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmc_${tag}_$i.log 2>&1
  echo "pass $i ($set) rc=$?"
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_$i.csv
done
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for f in sorted(glob.glob(root + "/*_pmc_*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(f))
    for k, d in agg.items():
        if "decode" in k or "encode" in k or "topk" in k:
            print("  %-70s %s" % (k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()}))
PY
