cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
for i in 1 2 3; do
  DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/time_modes.py 1024 zipf exact,bf16 4 2>&1 | grep streams= | cut -c1-70 | sed 's/^/NEW /'
  DAE_ENC_WG1=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/time_modes.py 1024 zipf exact,bf16 4 2>&1 | grep streams= | cut -c1-70 | sed 's/^/OLD /'
done
for i in 1 2; do
  DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/bench_loop.py 256 native exact_bf16 3 2>&1 | grep "playlists/s" | cut -c1-70 | sed 's/^/NEW /'
  DAE_ENC_WG1=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/bench_loop.py 256 native exact_bf16 3 2>&1 | grep "playlists/s" | cut -c1-70 | sed 's/^/OLD /'
done
