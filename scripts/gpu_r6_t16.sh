cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python -m pytest tests/test_gpu_title.py tests/test_gpu_title_exact.py tests/test_gpu_stream_loop.py -x -q 2>&1 | tail -3 | tee $o/r06_t16.log
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for rep in 1 2; do
for cfg in "0 999" "2 999" "2 8" "2 16"; do
  set -- $cfg
  DAE_PIPE_OUT=$1 DAE_MIX_SAMPLE_NB=$2 python scripts/time_title.py exact_bf16 300 2>&1 | grep -i "playlists/s" | cut -c1-120 | sed "s/^/out=$1 nb=$2 /"
done; done | tee -a $o/r06_t16.log
unset DAE_LIB_AB
python scripts/time_title.py f32 60 2>&1 | grep -i "playlists/s" | cut -c1-120 | tee -a $o/r06_t16.log
python scripts/time_title.py bf16 200 2>&1 | grep -i "playlists/s" | cut -c1-120 | tee -a $o/r06_t16.log
