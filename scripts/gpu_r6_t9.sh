cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python -m pytest tests/test_gpu_stream_loop.py tests/test_gpu_exact.py tests/test_gpu_title.py tests/test_gpu_title_exact.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6 | tee $o/r06_t9.log
python scripts/bench_loop.py 256 native exact_bf16 2,3,4 2>&1 | grep "playlists/s" | cut -c1-100 | tee -a $o/r06_t9.log
python scripts/bench_train.py --default 2>&1 | tail -4 | tee -a $o/r06_t9.log
python scripts/bench_train.py --bf16 2>&1 | tail -3 | tee -a $o/r06_t9.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trp && rocprofv3 --kernel-trace --stats -d /tmp/trp -o out -- python $GRAFT_REPO_ROOT/scripts/bench_train.py --default > /tmp/trp.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-150 | tee -a $GRAFT_REPO_ROOT/$o/r06_t9.log; cp "$f" $GRAFT_REPO_ROOT/$o/r06_train_default_kernel_stats.csv
