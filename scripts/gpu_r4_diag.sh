# round 4 diagnostic: the exact mode on hard models (rates, candidates per row, per-kernel times)
# usage: bash scripts/gpu_r4_diag.sh   (on the GPU box; writes gpurun_out/r4diag_*)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_exact.py -x -q 2>&1 | tail -5
run() {  # tag, env..., -- args
  tag=$1; shift
  ( env "$@" timeout 600 python scripts/time_modes.py 256 $BIAS f32,bf16,exact 1,4 ) > gpurun_out/r4diag_$tag.log 2>&1
  grep -v "^step\|simple_timer" gpurun_out/r4diag_$tag.log | tail -12
}
BIAS=zipf run zipf X=1
BIAS=zipf run zipf_x40 SCALE=40
BIAS=zeros run zeros X=1
BIAS=zeros run zeros_x40 SCALE=40
BIAS=zipf run trained TRAINED=${TRAIN_STEPS:-1500}
prof() { tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_$tag
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/scripts/time_modes.py 256 $BIAS exact 1 > /tmp/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/r4diag_${tag}_kernel_stats.csv && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-80s calls=%6s avg_us=%9.2f pct=%6s" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
  cd $GRAFT_REPO_ROOT
}
echo "== prof zipf_x40"; BIAS=zipf prof p_zipf_x40 SCALE=40
echo "== prof zeros"; BIAS=zeros prof p_zeros X=1
echo "== prof trained"; BIAS=zipf prof p_trained TRAINED=${TRAIN_STEPS:-1500}
