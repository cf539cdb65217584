cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
for i in 1 2 3; do TITLE_GAPS=1 python scripts/time_title.py exact_bf16 800 2>&1 | grep -E "titled rec|gaps" | cut -c1-330; done
