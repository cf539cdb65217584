# Round 6: the sample launch's own geometry -- DAE_SAMPLE_NB sweep (experiments build), step times alone / four batches in flight
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for B in 256 1024 2048; do
  for nb in 999 64 32 16 8; do
    DAE_SAMPLE_NB=$nb python scripts/time_modes.py $B zipf exact,bf16 1,4 2>&1 | grep streams= | sed "s/^/B=$B nb=$nb /"
  done
done | tee $o/r06_nb.log
for nb in 999 16; do
  DAE_SAMPLE_NB=$nb python scripts/time_modes.py 256 zeros exact 1,4 2>&1 | grep streams= | sed "s/^/B=256 zeros nb=$nb /"
  DAE_SAMPLE_NB=$nb TRAINED=1500 python scripts/time_modes.py 256 zipf exact 1,4 2>&1 | grep streams= | sed "s/^/B=256 trained nb=$nb /"
done | tee -a $o/r06_nb.log
unset DAE_LIB_AB
# parity of the default build on the paths the change touches
python -m pytest tests/test_gpu_exact.py tests/test_gpu_bf16.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5 | tee -a $o/r06_nb.log
