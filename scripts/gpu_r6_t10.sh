cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $o/r06_gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --verbose-out $o/r06_bench_verbose.json > $o/r06_bench_default.json 2> $o/err_default.txt
wc -c $o/r06_bench_default.json; tail -3 $o/err_default.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
