cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python -m pytest tests/test_gpu_stream_loop.py tests/test_gpu_exact.py tests/test_gpu_title.py tests/test_gpu_title_exact.py -x -q 2>&1 | tail -6 | tee $o/r06_t6.log
for rep in 1 2; do python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s"; done | tee -a $o/r06_t6.log
python scripts/bench_loop.py 256 native exact_bf16 2,4 2>&1 | grep "playlists/s" | tee -a $o/r06_t6.log
python scripts/bench_loop.py 150 native exact_bf16,f32 3 2>&1 | grep "playlists/s" | tee -a $o/r06_t6.log
bash scripts/trace_loop.sh exact_loop3 python $GRAFT_REPO_ROOT/scripts/bench_loop.py 256 native exact_bf16 3 2>&1 | tee -a $o/r06_t6.log
bash scripts/gpu_r6_calib.sh 2>&1 | tee -a $o/r06_t6.log
