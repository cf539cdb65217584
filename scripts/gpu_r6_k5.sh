cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
for rep in 1 2; do
for lib in default k5qr8 k5qr16; do
  if [ $lib = default ]; then unset DAE_LIB_AB; else export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_$lib.so; fi
  python scripts/bench_train.py --default 2>&1 | grep ms_per_step | cut -c1-110 | sed "s/^/$lib /"
done; done | tee $o/r06_k5.log
for lib in k5qr8 k5qr16; do
  export DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_$lib.so
  bash scripts/gpu_kprof.sh r06_train_$lib 4 python $GRAFT_REPO_ROOT/scripts/bench_train.py --default 2>&1 | sed "s/^/$lib /" | tee -a $o/r06_k5.log
done
