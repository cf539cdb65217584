p() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], r['frac'], d['config']['plan']['S'])"; }
for w in 4 8; do
  DAE_DECODE_WAVES=$w python bench.py --streams 1 --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | p "f32 B256 waves=$w"
  DAE_DECODE_WAVES=$w python bench.py --streams 1 --batch-per-gpu 1024 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | p "f32 B1024 waves=$w"
  DAE_DECODE_WAVES_BF16=$w python bench.py --dtype bf16 --streams 1 --steps 40 --warmup 5 2>&1 | p "bf16 B256 waves=$w"
  DAE_DECODE_WAVES_BF16=$w python bench.py --dtype bf16 --streams 1 --batch-per-gpu 1024 --steps 20 --warmup 3 2>&1 | p "bf16 B1024 waves=$w"
done
