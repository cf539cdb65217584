cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f plan=%s' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0), (d['config']['plan']['n_sample_tiles'], d['config']['plan']['n_filter_tiles'])))
" "$1"; }
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
DAE_SAMPLE=strided python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
for e in "" "DAE_NO_SAMPLE_PAD=1"; do
env $e python bench.py --no-cpu-baseline 2>/dev/null | short "f32 B256 2s $e"
env $e python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s $e"
env $e python bench.py --no-cpu-baseline --streams 1 --batch-per-gpu 1024 2>/dev/null | short "f32 B1024 1s $e"
done
