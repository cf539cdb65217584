# the title bound on 2-norms: tests, candidates per row and the titled loop's rate (old build = scripts/probe/libdae_hip_old.so when present)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_title_exact.py tests/test_gpu_title.py -x -q 2>&1 | tail -4
for i in 1 2; do
  python scripts/time_title.py exact_bf16 2>&1 | grep "playlists/s\|last launch" | tail -3
  if [ -f scripts/probe/libdae_hip_old.so ]; then DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_old.so python scripts/time_title.py exact_bf16 2>&1 | grep "playlists/s\|last launch" | tail -3 | sed 's/^/OLD /'; fi
done
timeout 900 python scripts/fuzz_title_exact.py 60 71 2>&1 | tail -3
