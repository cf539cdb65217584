cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python scripts/bench_loop.py 256 native exact_bf16,bf16,exact_bf16,bf16,exact_bf16,bf16,bf16,exact_bf16,exact_bf16 3 2>&1 | grep "playlists/s" | cut -c1-120 | tee $o/r06_t15.log
python scripts/probe/loop_in_context.py 0 2>&1 | grep "extra contexts" | tee -a $o/r06_t15.log
python scripts/time_title.py exact_bf16 200 2>&1 | grep -i "playlists" | cut -c1-200 | tee -a $o/r06_t15.log
