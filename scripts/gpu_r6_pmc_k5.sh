# where the training step's big launches spend their wave cycles (SQ counters, quad-cycles) and what their loads do in the TCP / L2.
# usage: bash scripts/gpu_r6_pmc_k5.sh [--default | --default-f32]
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
MODE=${1:---default}
export TMPDIR=/tmp
pass() {
  d=/tmp/pmc_k5_$1; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $d -o p -- python $R/scripts/bench_train.py $MODE > $d.log 2>&1 )
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $1: no output"; tail -5 $d.log; return; fi
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    for key in ("decode_loss_","grad_hidden","grad_wdec_t"):
        if key in k: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    print(k, {n: round(sum(v)/len(v),1) for n,v in c.items()})
PY
}
pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
pass b "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
pass c "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
pass e "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
