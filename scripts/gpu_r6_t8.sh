cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for rep in 1 2; do
for om in 0 1 2; do
  DAE_PIPE_OUT=$om python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-220 | sed "s/^/out=$om /"
done; done | tee $o/r06_t8.log
DAE_PIPE_OUT=2 python scripts/bench_loop.py 150 native exact_bf16,f32 3 2>&1 | grep "playlists/s" | cut -c1-220 | sed "s/^/out=2 /" | tee -a $o/r06_t8.log
DAE_PIPE_OUT=0 python scripts/bench_loop.py 150 native exact_bf16,f32 3 2>&1 | grep "playlists/s" | cut -c1-220 | sed "s/^/out=0 /" | tee -a $o/r06_t8.log
