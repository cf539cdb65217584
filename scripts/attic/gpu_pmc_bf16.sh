# usage: [DAE_BF16_ONE_TILE=1] bash scripts/gpu_pmc_bf16.sh <tag>   (one rocprofv3 --pmc pass per counter set)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); rm -rf /tmp/pb_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pb_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --streams 1 --batch-per-gpu 1024 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pb_$i.log 2>&1
  f=$(find /tmp/pb_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_bf16_${tag}_pmc_$i.csv || tail -5 /tmp/pb_$i.log
done
python - $tag <<'PY'
import csv, glob, os, collections, sys
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for f in sorted(glob.glob(root + "/r01_bf16_%s_pmc_*.csv" % sys.argv[1])):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "decode" in k and ("filter" in k or "1, 16" in k):
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
