cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python scripts/probe/loop_in_context.py 0 2>&1 | grep "extra contexts" | tee $o/r06_t11.log
python scripts/probe/loop_in_context.py 7 2>&1 | grep "extra contexts" | tee -a $o/r06_t11.log
python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-100 | tee -a $o/r06_t11.log
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for nl in 2 3 4; do
DAE_PIPE_CUMASK=1 python scripts/bench_loop.py 256 native exact_bf16,bf16 $nl 2>&1 | grep "playlists/s" | cut -c1-100 | sed "s/^/cumask /"
python scripts/bench_loop.py 256 native exact_bf16,bf16 $nl 2>&1 | grep "playlists/s" | cut -c1-100 | sed "s/^/plain  /"
done | tee -a $o/r06_t11.log
