# the fp32 training step (bench_train.py --default-f32) after a change, against scripts/probe/libdae_hip_old.so: tests, kernel times, steps
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_train_sharded.py -x -q 2>&1 | tail -4
echo "=== new"; bash scripts/gpu_kprof.sh k5fnew 4 python $R/scripts/bench_train.py --default-f32
echo "=== old"; DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so bash scripts/gpu_kprof.sh k5fold 4 python $R/scripts/bench_train.py --default-f32
for i in 1 2; do python scripts/bench_train.py --default-f32 | tail -1 | cut -c1-120; DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so python scripts/bench_train.py --default-f32 | tail -1 | cut -c1-120 | sed 's/^/OLD /'; done
