cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-170 | sed "s/^/plain  /" | tee $o/r06_t13.log
FREEZE=1 python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-170 | sed "s/^/freeze /" | tee -a $o/r06_t13.log
HOLD=1 python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-170 | sed "s/^/hold   /" | tee -a $o/r06_t13.log
FREEZE=1 HOLD=1 python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | cut -c1-170 | sed "s/^/both   /" | tee -a $o/r06_t13.log
python scripts/probe/loop_in_context.py 0 2>&1 | grep "extra contexts" | tee -a $o/r06_t13.log
