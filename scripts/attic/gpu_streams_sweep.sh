# usage: bash scripts/gpu_streams_sweep.sh [extra bench args]   -> one "S <streams> <dtype> <playlists/s> <ms/step>" line per run
cd $GRAFT_REPO_ROOT
for dt in f32 bf16; do for s in 1 2 3 4; do
  timeout 150 python bench.py --dtype $dt --no-cpu-baseline --no-train-row --no-bf16-row --streams $s --steps 400 --warmup 40 "$@" 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('S', $s, '$dt', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
