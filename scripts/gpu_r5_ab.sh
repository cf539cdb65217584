# Round-5 A/B on one box: the library of the tree against other builds under scripts/probe/ (libdae_hip_<tag>.so), same commands.
# usage: bash scripts/gpu_r5_ab.sh "<modes>" "<streams>" "<batches>" "<tags: new r04 ...>" [bias]
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
modes=${1:-exact}; streams=${2:-1,4}; batches=${3:-256 1024}; tags=${4:-new r04}; bias=${5:-zipf}
for B in $batches; do
  for lib in $tags; do
    if [ $lib = new ]; then unset DAE_LIB_AB; else export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_$lib.so; fi
    python scripts/time_modes.py $B $bias $modes $streams 2>&1 | grep streams= | sed "s/^/B=$B $lib /"
  done
done | tee -a $o/r05_ab.log
unset DAE_LIB_AB
