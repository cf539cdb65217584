# per-kernel times of the training step, new build against scripts/probe/libdae_hip_old.so
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash scripts/gpu_kprof.sh train_new 8 python $R/scripts/bench_train.py --default
DAE_LIB_AB=$R/scripts/probe/libdae_hip_old.so bash scripts/gpu_kprof.sh train_old 8 python $R/scripts/bench_train.py --default
