# the fp32 training step (bench_train.py --default-f32) after a change: tests, kernel times and steps, against an experiments-build
# switch ($1, e.g. DAE_K6_GENERIC) or scripts/probe/libdae_hip_old.so
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
SW=${1:-NONE}
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_train_sharded.py -x -q 2>&1 | tail -4
echo "=== new"; bash scripts/gpu_kprof.sh k5fnew 4 python $R/scripts/bench_train.py --default-f32
if [ $SW != NONE ]; then export DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so; export $SW=1; else export DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so; fi
echo "=== old ($SW)"; bash scripts/gpu_kprof.sh k5fold 4 python $R/scripts/bench_train.py --default-f32
unset DAE_LIB_AB
for i in 1 2; do python scripts/bench_train.py --default-f32 | tail -1 | cut -c1-120
  if [ $SW != NONE ]; then DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/bench_train.py --default-f32 | tail -1 | cut -c1-120 | sed 's/^/OLD /'; fi; done
