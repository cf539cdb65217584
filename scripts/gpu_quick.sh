cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0)))
" "$1"; }
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for e in "" "DAE_F32_GENERIC=1"; do
env $e python bench.py --no-cpu-baseline 2>/dev/null | short "f32 B256 2s $e"
env $e python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s $e"
env $e python bench.py --no-cpu-baseline --streams 1 --batch-per-gpu 1024 2>/dev/null | short "f32 B1024 1s $e"
env $e python bench.py --no-cpu-baseline --streams 1 --batch-per-gpu 128 2>/dev/null | short "f32 B128 1s $e"
done
