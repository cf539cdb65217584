# Round 5: the shared recomputation of the exact mode (refine.hip exact_rescore_shared_kernel) -- tests, then A/B through the
# experiments build (DAE_RF_SHARED=0/1) at 256 / 1024 / 2048 rows.
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_stream_loop.py -x -q 2>&1 | tail -8
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for B in 256 1024 2048; do
  for sh in 0 1; do
    DAE_RF_SHARED=$sh python scripts/time_modes.py $B zipf exact 1,4 2>&1 | grep streams= | sed "s/^/B=$B shared=$sh /"
  done
done | tee $o/r05_shared_ab.log
unset DAE_LIB_AB
for B in 1024 2048; do bash scripts/gpu_kprof.sh r05s_exact_b${B}_1stream 6 python $GRAFT_REPO_ROOT/scripts/time_modes.py $B zipf exact 1; done
