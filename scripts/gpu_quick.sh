cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0)))
" "$1"; }
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
DAE_TOPK_FAST=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
DAE_TOPK_FAST=0 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
for f in 0 1 2; do
export DAE_TOPK_FAST=$f
python bench.py --no-cpu-baseline 2>/dev/null | short "f32 B256 2s fast=$f"
python bench.py --no-cpu-baseline --streams 1 2>/dev/null | short "f32 B256 1s fast=$f"
python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "bf16 B256 2s fast=$f"
done
