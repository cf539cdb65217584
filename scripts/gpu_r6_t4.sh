cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
bash scripts/trace_loop.sh exact_loop python $GRAFT_REPO_ROOT/scripts/bench_loop.py 256 native exact_bf16 3 2>&1 | tee $o/r06_t4.log
bash scripts/trace_loop.sh exact_dev python $GRAFT_REPO_ROOT/scripts/time_modes.py 2048 zipf exact 3 2>&1 | tee -a $o/r06_t4.log
cd $GRAFT_REPO_ROOT
for B in 256 1024; do python scripts/time_cumask.py $B exact 4 2>&1 | grep partition=; done | tee -a $o/r06_t4.log
python scripts/time_cumask.py 256 exact 2 2>&1 | grep partition= | tee -a $o/r06_t4.log
python scripts/time_cumask.py 256 bf16 4 2>&1 | grep partition= | tee -a $o/r06_t4.log
