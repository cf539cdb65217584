cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export GPU_MAX_HW_QUEUES=32
for l in 2 3 4; do TITLE_LANES=$l python scripts/time_title.py exact_bf16 800 2>&1 | grep -E "titled rec" | sed "s/^/[lanes $l] /"; done
for c in 3 8; do TITLE_COALESCE=$c python scripts/time_title.py exact_bf16 800 2>&1 | grep -E "titled rec" | sed "s/^/[coalesce $c] /"; done
python scripts/time_title.py f32 200 2>&1 | grep -E "titled rec"
python scripts/time_title.py bf16 400 2>&1 | grep -E "titled rec"
TITLE_ENGINE=python python scripts/time_title.py exact_bf16 800 2>&1 | grep -E "titled rec" | sed "s/^/[python] /"
