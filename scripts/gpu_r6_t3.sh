cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
./scripts/probe/cumask_probe 2>&1 | tee $o/r06_cumask_probe.log
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for rep in 1 2; do
DAE_PIPE_DIRECT=1 python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | sed "s/^/direct /"
python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | sed "s/^/outstr /"
done | tee $o/r06_t3.log
python scripts/bench_loop.py 256 native exact_bf16 2,4 2>&1 | grep "playlists/s" | sed "s/^/outstr /" | tee -a $o/r06_t3.log
python scripts/bench_loop.py 150 native exact_bf16,f32 3 2>&1 | grep "playlists/s" | sed "s/^/outstr /" | tee -a $o/r06_t3.log
unset DAE_LIB_AB
python -m pytest tests/test_gpu_stream_loop.py tests/test_gpu_title.py tests/test_gpu_title_exact.py tests/test_gpu_exact.py -x -q 2>&1 | tail -6 | tee -a $o/r06_t3.log
