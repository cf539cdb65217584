# Round-6 counter passes on the SHIPPED kernels (one rocprofv3 --pmc pass per counter set, kernel trace only:
# MI355X_MICROARCH.md HBM / rocprofv3 section).  Writes gpurun_out/r06_pmc_<cfg>_<set>.csv and the two JSON
# summaries bench.py reads (copy them to profiles/: traffic_decode.json, traffic_encode.json).
#   usage: bash scripts/gpu_pmc_round6.sh
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() {  # cfg set-name counters... -- bench args
  cfg=$1; name=$2; shift 2; ctrs=""
  while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  d=/tmp/pmc6_${cfg}_${name}; rm -rf $d
  timeout 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --prime-ms 0 --no-cpu-baseline --no-train-row --no-bf16-row --no-extra-rows --streams 1 "$@" > $d.log 2>&1
  echo "pass $cfg/$name rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_${cfg}_${name}.csv
}
for cfg in f32 bf16; do
  pass $cfg fetch FETCH_SIZE -- --dtype $cfg
  pass $cfg write WRITE_SIZE -- --dtype $cfg
done
pass exact fetch FETCH_SIZE -- --dtype exact_bf16
pass exact write WRITE_SIZE -- --dtype exact_bf16
pass f32 mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- --dtype f32
pass bf16 mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- --dtype bf16
# the training step's two HBM-bound launches (the fused decoder-gradient + Adam kernel, the dense Adam): bytes through the fabric
tpass() {  # set-name counter
  d=/tmp/pmc6_train_$1; rm -rf $d
  timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --default > $d.log 2>&1
  echo "pass train/$1 rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_train_$1.csv
}
tpass fetch FETCH_SIZE
tpass write WRITE_SIZE
# the exact title mix (dae_mix_topk_exact): the two-GEMM filter launch and the refine launch, 750 rows per launch
mpass() {  # set-name counter
  d=/tmp/pmc6_title_$1; rm -rf $d
  timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/scripts/time_title.py exact_bf16 20 > $d.log 2>&1
  echo "pass title/$1 rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_title_$1.csv
}
mpass fetch FETCH_SIZE
mpass write WRITE_SIZE
python $GRAFT_REPO_ROOT/scripts/pmc_summarise.py $GRAFT_REPO_ROOT/gpurun_out r06
