# encode_split_kernel: a workgroup per row (4 waves) against a workgroup per (row, quarter) (experiments build, DAE_ENC_WG1): the
# headline loop of bench.py through scripts/probe/bench_ab.py, three pairs
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
A="--gpus 1 --steps 20 --warmup 5 --no-extra-rows --no-train-row --no-cpu-baseline --no-bf16-row"
for i in 1 2 3; do
  DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/probe/bench_ab.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NEW', d['value'], d['ms_per_step'], d.get('value_runs_M'))"
  DAE_ENC_WG1=1 DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so python scripts/probe/bench_ab.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OLD', d['value'], d['ms_per_step'], d.get('value_runs_M'))"
done
