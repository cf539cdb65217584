# the training step after a change: tests, per-kernel times new / old (scripts/probe/libdae_hip_old.so when present), step timings
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_sharded.py -x -q 2>&1 | tail -3
bash scripts/gpu_kprof.sh trainab 14 python $R/scripts/bench_train.py --default
if [ -f scripts/probe/libdae_hip_old.so ]; then echo "=== old"; DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so bash scripts/gpu_kprof.sh trainab_old 14 python $R/scripts/bench_train.py --default; fi
for i in 1 2 3; do python scripts/bench_train.py --default | tail -1 | cut -c1-120
  if [ -f scripts/probe/libdae_hip_old.so ]; then DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so python scripts/bench_train.py --default | tail -1 | cut -c1-120 | sed 's/^/OLD /'; fi; done
