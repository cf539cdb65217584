cd $GRAFT_REPO_ROOT
for v in 0 1 2 4 5; do
  DAE_DBG_SCAN=$v python bench.py --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 --steps 60 --prime-ms 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('DBG_SCAN=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
DAE_DBG_NOSCAN=1 python bench.py --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 --steps 60 --prime-ms 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('NOSCAN', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
