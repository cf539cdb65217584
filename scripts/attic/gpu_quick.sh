cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac']))
" "$1"; }
DAE_TOPK_THREADS=512 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for e in "DAE_TOPK_PAIR512=0" "DAE_TOPK_PAIR512=1"; do
env $e python bench.py --no-cpu-baseline 2>/dev/null | short "f32 B256 2s $e"
env $e python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s $e"
env $e python bench.py --no-cpu-baseline --batch-per-gpu 1024 2>/dev/null | short "f32 B1024 2s $e"
env $e python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "bf16 B256 2s $e"
env $e python bench.py --no-cpu-baseline --sim-world 8 2>/dev/null | short "f32 sim-world 8 $e"
env $e python bench.py --no-cpu-baseline --sim-world 4 2>/dev/null | short "f32 sim-world 4 $e"
done
