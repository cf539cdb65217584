cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_sharded.py -x -q 2>&1 | tail -4
bash scripts/gpu_r6_pmc_exact.sh 256 exact
