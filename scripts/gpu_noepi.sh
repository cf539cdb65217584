p() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', d['value'], 'kernel', r['avg_launch_ms'], r['frac'])"; }
python bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | p normal
DAE_DBG_NOEPI=1 python bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | p noepi
python bench.py --dtype bf16 --streams 1 --batch-per-gpu 1024 --steps 20 --warmup 3 2>&1 | p bf16_1024
DAE_DBG_NOEPI=1 python bench.py --dtype bf16 --streams 1 --batch-per-gpu 1024 --steps 20 --warmup 3 2>&1 | p bf16_1024_noepi
