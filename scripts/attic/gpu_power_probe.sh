# usage: bash scripts/gpu_power_probe.sh <streams> <dtype>  -> samples rocm-smi power / sclk while bench.py runs a long timed region
cd $GRAFT_REPO_ROOT
s=${1:-2}; dt=${2:-f32}
python bench.py --dtype $dt --no-cpu-baseline --no-train-row --no-bf16-row --streams $s --steps 60000 --warmup 40 > /tmp/pp.json 2>/dev/null < /dev/null &
bp=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction|edge" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 1
done
wait $bp
python -c "import json; d=json.loads(open('/tmp/pp.json').readlines()[-1]); print('RES', d['value'], d['ms_per_step'])"
