# K5 through LDS: tests, then per-kernel times against the 128-row kernel (experiments build, DAE_K5_ROWS128)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
echo "=== new (default build)"; bash scripts/gpu_kprof.sh k5new 4 python $R/scripts/bench_train.py --default
echo "=== 128-row kernel"; DAE_K5_ROWS128=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so bash scripts/gpu_kprof.sh k5rows128 4 python $R/scripts/bench_train.py --default
for i in 1 2; do python scripts/bench_train.py --default | tail -1; DAE_K5_ROWS128=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/bench_train.py --default | tail -1; done
