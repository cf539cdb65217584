cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for x in 1 4; do for om in 2 0; do
  REPS_X=$x DAE_PIPE_OUT=$om python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-230 | sed "s/^/reps_x=$x out=$om /"
done; done | tee $o/r06_t12.log
for om in 2 0; do
  DAE_PIPE_OUT=$om python scripts/time_title.py exact_bf16 200 2>&1 | tail -2 | cut -c1-200 | sed "s/^/title out=$om /"
done | tee -a $o/r06_t12.log
