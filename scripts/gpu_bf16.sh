p() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', d['value'], 'kernel', r['avg_launch_ms'], r['frac'])"; }
for st in 0 1 2 3 5; do DAE_STAGGER=$st python bench.py --dtype bf16 --streams 1 --batch-per-gpu 1024 --steps 10 --warmup 2 2>&1 | p "bf16 B1024 stagger=$st"; done
DAE_STAGGER=2 python bench.py --dtype bf16 --streams 1 --steps 30 --warmup 3 2>&1 | p "bf16 B256 stagger=2"
DAE_STAGGER=0 python bench.py --dtype bf16 --streams 1 --steps 30 --warmup 3 2>&1 | p "bf16 B256 stagger=0"
