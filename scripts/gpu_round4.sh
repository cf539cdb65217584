# Round-4 session on the GPU box: the JSON lines and rocprofv3 kernel stats quoted in DESIGN.md / profiles/r04_notes.md
# (copy gpurun_out/r04_* into profiles/ afterwards).  usage: bash scripts/gpu_round4.sh [quick]
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
python bench.py > $o/r04_bench_default.json 2> $o/err_default.txt
python bench.py --dtype bf16 --no-train-row --no-cpu-baseline > $o/r04_bench_bf16.json 2> $o/err_bf16.txt
python bench.py --dtype exact_bf16 --no-train-row > $o/r04_bench_exact.json 2> $o/err_exact.txt
python bench.py --force-dist --no-train-row --no-cpu-baseline --no-bf16-row --no-extra-rows > $o/r04_bench_forcedist.json 2> $o/err_fd.txt
for f in default bf16 exact forcedist; do python - $o/r04_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    ex={k:(v.get('value'), v.get('ms_per_step')) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='roofline'}
    print('%-12s value=%10.0f ms=%.4f kern=%s %.4f frac=%.3f %s' % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel'][:30], r['avg_launch_ms'], r['frac'], ex))
    for k in ('exact_bf16_hard', 'trained_model'):
        for kk, v in (d.get(k) or {}).items():
            if isinstance(v, dict):
                print('   ', k, kk, {a: (b.get('value') if isinstance(b, dict) else b) for a, b in v.items() if a in ('f32', 'exact_bf16', 'bf16', 'error')})
    if 'phases' in d: print('    phases', {a: b for a, b in d['phases'].items() if a.endswith('_ms')})
    if 'training_step' in d: print('    train', {a: b.get('ms_per_step') for a, b in d['training_step'].items() if isinstance(b, dict)})
    if 'drivers_loop' in d: print('    loop', {a: b.get('value') for a, b in d['drivers_loop'].items() if isinstance(b, dict) and 'value' in b})
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('r04_bench_','err_').replace('.json','.txt')).read()[-800:] if False else '')
PY
done
bash scripts/gpu_kprof.sh r04_headline 7 python $GRAFT_REPO_ROOT/bench.py --no-extra-rows --no-train-row --no-cpu-baseline --no-bf16-row
bash scripts/gpu_kprof.sh r04_1stream 7 python $GRAFT_REPO_ROOT/bench.py --streams 1 --no-train-row --no-cpu-baseline --no-bf16-row --no-extra-rows
bash scripts/gpu_kprof.sh r04_bf16_1stream 7 python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --streams 1 --no-train-row --no-cpu-baseline
bash scripts/gpu_kprof.sh r04_exact_1stream 7 python $GRAFT_REPO_ROOT/bench.py --dtype exact_bf16 --streams 1 --no-train-row --no-cpu-baseline
bash scripts/gpu_kprof.sh r04_exact_4streams 7 python $GRAFT_REPO_ROOT/bench.py --dtype exact_bf16 --no-train-row --no-cpu-baseline
# training step, per kernel: fp32 / bf16 GEMMs / the model's default step
bash scripts/gpu_kprof.sh r04_train_f32 12 python $GRAFT_REPO_ROOT/scripts/bench_train.py
bash scripts/gpu_kprof.sh r04_train_bf16 12 python $GRAFT_REPO_ROOT/scripts/bench_train.py --bf16
bash scripts/gpu_kprof.sh r04_train_default 14 python $GRAFT_REPO_ROOT/scripts/bench_train.py --default
# the exact mode on hard models, per kernel (one batch in flight)
SCALE=40 bash scripts/gpu_kprof.sh r04_exact_x40 7 python $GRAFT_REPO_ROOT/scripts/time_modes.py 256 zipf exact 1
bash scripts/gpu_kprof.sh r04_exact_zeros 7 python $GRAFT_REPO_ROOT/scripts/time_modes.py 256 zeros exact 1
TRAINED=1500 bash scripts/gpu_kprof.sh r04_exact_trained 16 python $GRAFT_REPO_ROOT/scripts/time_modes.py 256 zipf exact 1
# the reference's real --challenge path (every batch title-mixed), per kernel: fp32 / both GEMMs bf16 / exact_bf16
bash scripts/gpu_kprof.sh r04_title_f32 8 python $GRAFT_REPO_ROOT/scripts/time_title.py f32
bash scripts/gpu_kprof.sh r04_title_bf16 8 python $GRAFT_REPO_ROOT/scripts/time_title.py bf16
bash scripts/gpu_kprof.sh r04_title_exact 12 python $GRAFT_REPO_ROOT/scripts/time_title.py exact_bf16
(python scripts/time_title.py f32 200; python scripts/time_title.py bf16 200; python scripts/time_title.py exact_bf16 400) 2>&1 | grep -v amdgpu > $o/r04_title_loop.log
cat $o/r04_title_loop.log
# the drivers' loop per engine (scripts/bench_loop.py)
(python scripts/bench_loop.py 256 native,python; python scripts/bench_loop.py 150 native,python) 2>&1 | grep playlists | cut -c1-90 > $o/r04_bench_loop.log
cat $o/r04_bench_loop.log
