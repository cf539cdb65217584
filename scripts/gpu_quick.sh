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
python bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | short "bf16 B256 2s"
python bench.py --no-cpu-baseline --dtype bf16 --streams 1 2>/dev/null | short "bf16 B256 1s"
bash scripts/gpu_prof.sh tkx --no-cpu-baseline --streams 1 2>&1 | grep "topk_kernel" | cut -c1-60,95-
