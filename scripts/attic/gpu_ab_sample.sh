# A/B: bias-ordered threshold sample vs strided sample, both bias variants, both dtypes
cd $GRAFT_REPO_ROOT
short() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%-34s value=%9.0f ms=%.4f kern_ms=%.4f frac=%.3f iso=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('isolated',{}).get('avg_launch_ms',0)))
" "$1"; }
for dt in f32 bf16; do
 for bias in zipf zeros; do
  python bench.py --no-cpu-baseline --dtype $dt --bias $bias 2>/dev/null | short "ordered $dt $bias"
  DAE_SAMPLE=strided python bench.py --no-cpu-baseline --dtype $dt --bias $bias 2>/dev/null | short "strided $dt $bias"
 done
done
python bench.py --no-cpu-baseline --dtype bf16 --batch-per-gpu 1024 --streams 1 2>/dev/null | short "ordered bf16 B1024 1s"
python bench.py --no-cpu-baseline --dtype f32 --batch-per-gpu 1024 --streams 1 2>/dev/null | short "ordered f32 B1024 1s"
