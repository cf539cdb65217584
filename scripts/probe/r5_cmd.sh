cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
