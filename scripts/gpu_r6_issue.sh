# issue / completion times per step with several batches in flight (r06 notes 1): scripts/time_issue.py
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
for B in 256 1024; do
  for m in exact bf16; do
    timeout 300 python scripts/time_issue.py $B $m 4 2>&1 | grep "B=\|failed"
  done
done | tee $o/r06_issue.log
timeout 300 python scripts/time_issue.py 256 exact 8 2>&1 | grep "B=\|failed" | tee -a $o/r06_issue.log
