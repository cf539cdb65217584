# Round-5 A/B on one box: the library of the tree against scripts/probe/libdae_hip_r04.so (round 4's sources), same commands.
# usage: bash scripts/gpu_r5_ab.sh [modes] [streams]
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
modes=${1:-exact}; streams=${2:-1,4}
for B in 256 1024; do
  for lib in new r04; do
    if [ $lib = r04 ]; then export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_r04.so; else unset DAE_LIB_AB; fi
    python scripts/time_modes.py $B zipf $modes $streams 2>&1 | grep streams= | sed "s/^/B=$B $lib /"
  done
done | tee $o/r05_ab.log
unset DAE_LIB_AB
