# K5 with one part removed at a time (experiments build, DAE_K5_X) + K7 with a workgroup's waves on one hidden half
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
for X in 0 1 2 3 4; do
  echo "=== DAE_K5_X=$X"
  DAE_K5_X=$X DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so bash scripts/gpu_kprof.sh k5x$X 3 python $R/scripts/bench_train.py --default
done
echo "=== default build"; bash scripts/gpu_kprof.sh k5def 4 python $R/scripts/bench_train.py --default
python scripts/bench_train.py --default | tail -1
