cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python -m pytest tests/test_gpu_stream_loop.py tests/test_gpu_csr.py tests/test_gpu_exact.py tests/test_gpu_title_exact.py tests/test_gpu_sharded_scoring.py -x -q 2>&1 | tail -8 | tee $o/r06_t1.log
python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | tee -a $o/r06_t1.log
python scripts/bench_loop.py 150 native exact_bf16 3 2>&1 | grep "playlists/s" | tee -a $o/r06_t1.log
