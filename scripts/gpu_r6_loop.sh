cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python scripts/bench_loop.py 256 native exact_bf16,bf16 3 2>&1 | grep "playlists/s" | tee $o/r06_loop.log
python scripts/bench_loop.py 256 native exact_bf16 2,4 2>&1 | grep "playlists/s" | tee -a $o/r06_loop.log
for B in 256 1024 2048; do
  python scripts/time_modes.py $B zipf exact,bf16 1,4 2>&1 | grep streams= | sed "s/^/B=$B hint-default /"
done | tee -a $o/r06_loop.log
