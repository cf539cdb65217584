cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0)))
" "$1"; }
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "rt256 bf16 B256 2s"
python bench.py --no-cpu-baseline --dtype bf16 --streams 1 2>/dev/null | short "rt256 bf16 B256 1s"
python bench.py --no-cpu-baseline --dtype bf16 --batch-per-gpu 1024 --streams 1 2>/dev/null | short "rt256 bf16 B1024 1s"
python bench.py --no-cpu-baseline --dtype bf16 --batch-per-gpu 1024 --streams 2 2>/dev/null | short "rt256 bf16 B1024 2s"
export DAE_BF16_RTILE=128
python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "rt128 bf16 B256 2s"
python bench.py --no-cpu-baseline --dtype bf16 --streams 1 2>/dev/null | short "rt128 bf16 B256 1s"
python bench.py --no-cpu-baseline --dtype bf16 --batch-per-gpu 1024 --streams 1 2>/dev/null | short "rt128 bf16 B1024 1s"
