# A/B of the exact title mix's kernels (experiments build): kernel times per variant
export GPU_MAX_HW_QUEUES=32
R=${GRAFT_REPO_ROOT:-$PWD}
export DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so
for v in "$@"; do
  echo "== $v"
  env $v bash $R/scripts/gpu_kprof.sh title_ab 4 python $R/scripts/time_title.py exact_bf16 20 2>&1 | grep -v "^W2026\|amdgpu.ids"
done
