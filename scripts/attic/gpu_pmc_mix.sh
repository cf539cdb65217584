# Instruction mix / stall split of every kernel of the step (one rocprofv3 --pmc pass, kernel trace only).
#   usage: bash scripts/gpu_pmc_mix.sh <tag> [bench args]   -> gpurun_out/<tag>_pmc_mix.csv
tag=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmcmix_$tag; rm -rf $d
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --prime-ms 0 --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 "$@" > $d.log 2>&1 < /dev/null
echo "pass rc=$?"; tail -2 $d.log | cut -c1-200
f=$(find $d -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_mix.csv
