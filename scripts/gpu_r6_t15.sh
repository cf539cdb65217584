# K5 epilogue A/B (old build = scripts/probe/libdae_hip_old.so), training tests
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
for i in 1 2; do
  echo "--- new"; python scripts/bench_train.py --default 2>&1 | tail -2
  echo "--- old"; DAE_LIB_AB=scripts/probe/libdae_hip_old.so python scripts/bench_train.py --default 2>&1 | tail -2
done
echo "--- new bf16"; python scripts/bench_train.py --bf16 2>&1 | tail -2
echo "--- old bf16"; DAE_LIB_AB=scripts/probe/libdae_hip_old.so python scripts/bench_train.py --bf16 2>&1 | tail -2
