export GPU_MAX_HW_QUEUES=32
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_exact.py -x -q 2>&1 | tail -2
export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so
for v in "DAE_RF_SHAPE=2" "DAE_RF_SHAPE=1"; do
  env DAE_DBG_R=1 $v python scripts/time_modes.py 256 zipf exact 1 2>&1 | grep -E "REFINE|streams=" | tail -3 | cut -c1-250 | sed "s/^/[$v] /"
  env $v python scripts/time_modes.py 256 zipf exact 4 2>&1 | grep -E "streams=" | cut -c1-70 | sed "s/^/[$v] /"
done
unset DAE_LIB_AB
python scripts/time_modes.py 1024 zipf exact 1,4 2>&1 | grep -E "streams=" | cut -c1-70
