cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python scripts/bench_loop.py 256 native exact_bf16,bf16,exact_bf16,bf16,exact_bf16,bf16,bf16,exact_bf16,exact_bf16 3 2>&1 | grep "playlists/s" | cut -c1-170 | tee $o/r06_t14.log
