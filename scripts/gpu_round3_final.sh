# Round-3 closing session on the GPU box: the JSON lines and rocprofv3 kernel stats quoted in DESIGN.md / profiles/r03_notes.md
cd $GRAFT_REPO_ROOT
o=gpurun_out; mkdir -p $o
python bench.py > $o/r03_bench_default.json 2> $o/err_default.txt
python bench.py --dtype bf16 --no-train-row --no-cpu-baseline > $o/r03_bench_bf16.json 2> $o/err_bf16.txt
python bench.py --dtype exact_bf16 --no-train-row > $o/r03_bench_exact.json 2> $o/err_exact.txt
python bench.py --streams 1 --no-train-row --no-cpu-baseline --no-bf16-row --no-extra-rows > $o/r03_bench_1stream.json 2> $o/err_1s.txt
python bench.py --bias zeros --dtype exact_bf16 --no-train-row --no-cpu-baseline > $o/r03_bench_exact_bias_zeros.json 2> $o/err_exz.txt
python bench.py --batch-per-gpu 1024 --dtype exact_bf16 --no-train-row --no-cpu-baseline > $o/r03_bench_exact_b1024.json 2> $o/err_ex1024.txt
for f in default bf16 exact 1stream exact_bias_zeros exact_b1024; do python - $o/r03_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    ex={k:(v.get('value'), v.get('ms_per_step')) for k,v in d.items() if isinstance(v,dict) and 'value' in v and k!='roofline'}
    print('%-18s value=%10.0f ms=%.4f kern=%s %.4f frac=%.3f %s' % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel'][:30], r['avg_launch_ms'], r['frac'], ex))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
bash scripts/gpu_prof.sh r03_default
bash scripts/gpu_prof.sh r03_headline --no-extra-rows --no-train-row --no-cpu-baseline --no-bf16-row
bash scripts/gpu_prof.sh r03_1stream --streams 1 --no-train-row --no-cpu-baseline --no-bf16-row --no-extra-rows
bash scripts/gpu_prof.sh r03_bf16_1stream --dtype bf16 --streams 1 --no-train-row --no-cpu-baseline
bash scripts/gpu_prof.sh r03_exact_1stream --dtype exact_bf16 --streams 1 --no-train-row --no-cpu-baseline
bash scripts/gpu_prof.sh r03_exact_4streams --dtype exact_bf16 --no-train-row --no-cpu-baseline
bash scripts/gpu_prof.sh r03_bf16_4streams --dtype bf16 --no-train-row --no-cpu-baseline
# the drivers' loop (host feeds in, host lists out) and the N > 1 code paths on one GPU
(python scripts/bench_shim.py 256; python scripts/bench_shim.py 256 --bf16; python scripts/bench_shim.py 256 --exact; python scripts/bench_shim.py 150; python scripts/bench_shim.py 150 --bf16; python scripts/bench_shim.py 150 --exact; python scripts/bench_shim.py 250 --exact) 2>&1 | grep "recommend" > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_shim.log
grep "recommend_iter" $GRAFT_REPO_ROOT/gpurun_out/r03_bench_shim.log
bash scripts/gpu_dist_r3.sh
