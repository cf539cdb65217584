cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python -m pytest tests/test_gpu_stream_loop.py tests/test_gpu_exact.py tests/test_gpu_csr.py tests/test_gpu_title.py -x -q 2>&1 | tail -6 | tee $o/r06_t5.log
for rep in 1 2; do python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s"; done | tee -a $o/r06_t5.log
python scripts/bench_loop.py 150 native exact_bf16 3 2>&1 | grep "playlists/s" | tee -a $o/r06_t5.log
bash scripts/trace_loop.sh exact_loop2 python $GRAFT_REPO_ROOT/scripts/bench_loop.py 256 native exact_bf16 3 2>&1 | tee -a $o/r06_t5.log
