cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0)))
" "$1"; }
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
DAE_TOPK_LEAN=1 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>/dev/null | short "f32 B256 2s"
python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s"
DAE_NO_WHOLE_TILE=1 python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s ring4 phase A"
python bench.py --no-cpu-baseline --streams 3 2>/dev/null | short "f32 B256 3s"
python bench.py --no-cpu-baseline --batch-per-gpu 1024 --streams 1 2>/dev/null | short "f32 B1024 1s"
python bench.py --no-cpu-baseline --batch-per-gpu 1024 --streams 2 2>/dev/null | short "f32 B1024 2s"
python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "bf16 B256 2s"
python bench.py --no-cpu-baseline --dtype bf16 --streams 3 2>/dev/null | short "bf16 B256 3s"
python bench.py --no-cpu-baseline --dtype bf16 --batch-per-gpu 1024 --streams 2 2>/dev/null | short "bf16 B1024 2s"
