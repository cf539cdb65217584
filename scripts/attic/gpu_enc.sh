p() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline_encode']; print('$1', d['value'], r['avg_launch_ms'], r['achieved'], r.get('large_batch', {}).get('achieved'))"; }
python bench.py --streams 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | p "B256 1stream"
python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | p "B256 2streams"
