cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for sk in 0 1 2 3; do
  DAE_PIPE_SKIP=$sk python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-100 | sed "s/^/skip=$sk /"
done | tee $o/r06_t7.log
DAE_PIPE_DIRECT=1 python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-100 | sed "s/^/direct /" | tee -a $o/r06_t7.log
python scripts/time_modes.py 2048 zipf exact,bf16 3,4 2>&1 | grep streams= | cut -c1-90 | tee -a $o/r06_t7.log
