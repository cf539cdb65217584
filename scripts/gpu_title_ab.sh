# the titled loop after a change: tests, kernel times and rates, new / old (scripts/probe/libdae_hip_old.so when present)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_title_exact.py tests/test_gpu_title.py -x -q 2>&1 | tail -3
bash scripts/gpu_kprof.sh titleab 5 python $R/scripts/time_title.py exact_bf16
if [ -f scripts/probe/libdae_hip_old.so ]; then echo "=== old"; DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so bash scripts/gpu_kprof.sh titleab_old 5 python $R/scripts/time_title.py exact_bf16; fi
for i in 1 2 3; do python scripts/time_title.py exact_bf16 2>&1 | grep "playlists/s"
  if [ -f scripts/probe/libdae_hip_old.so ]; then DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so python scripts/time_title.py exact_bf16 2>&1 | grep "playlists/s" | sed 's/^/OLD /'; fi; done
