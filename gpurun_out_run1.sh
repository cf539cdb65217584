set -x
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
DAE_DECODE_WAVES=4 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench1_w4.json 2>> gpurun_out/bench1.err; cat gpurun_out/bench1_w4.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/prof1.log 2>&1
find /tmp/prof1 -name "*stats*" | head; 
for f in $(find /tmp/prof1 -name "*kernel_stats*.csv"); do cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats.csv; done
head -20 $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats.csv
