# training step: A/B of the tree's library against the experiments build's switches is not needed here -- two runs of each bench
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
for rep in 1 2; do
python scripts/bench_train.py --default 2>&1 | grep ms_per_step | cut -c1-120
python scripts/bench_train.py --bf16 2>&1 | grep ms_per_step | cut -c1-120
python scripts/bench_train.py 2>&1 | grep ms_per_step | cut -c1-120
done | tee $o/r06_train.log
bash scripts/gpu_kprof.sh r06_train_default 12 python $GRAFT_REPO_ROOT/scripts/bench_train.py --default 2>&1 | tee -a $o/r06_train.log
