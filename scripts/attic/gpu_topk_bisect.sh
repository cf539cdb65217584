# stage costs of the phase-A top-k launch (DAE_TOPK_STOP=n returns after stage n), plus SLOWPATH rows
cd $GRAFT_REPO_ROOT
for st in 1 2 3 4 5; do
  echo "=== DAE_TOPK_STOP=$st"
  DAE_TOPK_STOP=$st bash scripts/gpu_prof.sh bis --steps 10 --warmup 2 --no-cpu-baseline --streams 1 2>&1 | grep "topk_kernel" | sed 's/void (anonymous namespace):://; s/(anonymous namespace):://g' | cut -c1-40,85-
  cd $GRAFT_REPO_ROOT
done
DAE_TOPK_STOP=99 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 1 2>&1 | grep -c SLOWPATH
DAE_TOPK_STOP=99 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --streams 1 2>&1 | grep SLOWPATH | head -5
