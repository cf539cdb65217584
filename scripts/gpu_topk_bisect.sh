for st in 1 2 3 4 5 -1 -2 -3 -4 -5; do
  echo "=== DAE_TOPK_STOP=$st"
  DAE_TOPK_STOP=$st DAE_DECODE_WAVES=4 bash scripts/gpu_prof.sh bis --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep "topk_kernel" | sed 's/void (anonymous namespace):://; s/(anonymous namespace):://g' | cut -c1-40,85-
  cd $GRAFT_REPO_ROOT
done
