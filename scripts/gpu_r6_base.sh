# Round-6 baseline on one box: step times per mode / batches in flight, and kernel timelines with four batches in flight.
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
for B in 256 1024 2048; do
  python scripts/time_modes.py $B zipf exact,bf16 1,2,4 2>&1 | grep streams= | sed "s/^/B=$B /"
done | tee $o/r06_base.log
bash scripts/trace_modes.sh 256 zipf exact 4 r06_exact256_4 2>&1 | tee -a $o/r06_base.log
bash scripts/trace_modes.sh 1024 zipf exact 4 r06_exact1024_4 2>&1 | tee -a $o/r06_base.log
bash scripts/trace_modes.sh 256 zipf exact 1 r06_exact256_1 2>&1 | tee -a $o/r06_base.log
