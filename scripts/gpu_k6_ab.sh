# K6 with its Adam streams kept in flight (32-row tiles) against the 64-row form (experiments build, DAE_K6_T64)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_train_sharded.py -x -q 2>&1 | tail -4
echo "=== new (default build)"; bash scripts/gpu_kprof.sh k6new 4 python $R/scripts/bench_train.py --default
echo "=== 64-row tiles"; DAE_K6_T64=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so bash scripts/gpu_kprof.sh k6t64 4 python $R/scripts/bench_train.py --default
for i in 1 2; do python scripts/bench_train.py --default | tail -1 | cut -c1-120; DAE_K6_T64=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/bench_train.py --default | tail -1 | cut -c1-120; done
