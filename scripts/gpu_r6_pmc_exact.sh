# where the launches of the exact mode (256 rows, one batch in flight) spend their wave cycles, and what their loads do:
# SQ / TA / TCP / TCC counters per kernel (quad-cycles for SQ_*; means per launch).  usage: bash scripts/gpu_r6_pmc_exact.sh [B] [mode]
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
B=${1:-256}; MODE=${2:-exact}
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=32
pass() {
  d=/tmp/pmc_ex_$1; rm -rf $d
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $d -o p -- python $R/scripts/time_modes.py $B zipf $MODE 1 > $d.log 2>&1 )
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $1: no output"; tail -5 $d.log; return; fi
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:44]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    n=max(len(v) for v in c.values())
    if n>=50: print("%-44s n=%4d"%(k,n), {a: round(sum(v)/len(v),1) for a,v in c.items()})
PY
}
pass a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
pass b "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES"
pass c "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
pass d "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
pass e "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
