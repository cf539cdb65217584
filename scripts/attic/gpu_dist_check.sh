cd $GRAFT_REPO_ROOT
o=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sharded_scoring.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -3)
for n in 4 8; do python bench.py --sim-world $n --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_sim$n.json 2>/dev/null; done
python bench.py --force-dist --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_forcedist.json 2> $o/r02f_bench_forcedist.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 20 --warmup 4 --no-train-row --no-cpu-baseline > $o/r02f_bench_gloo2.json 2> $o/r02f_bench_gloo2.err
for f in sim4 sim8 forcedist gloo2; do python - $o/r02f_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('%-12s value=%10.0f ms=%.4f kern=%s %.4f frac=%.3f extra=%s' % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel'][:34], r['avg_launch_ms'], r['frac'], {k:v for k,v in d.items() if k in ('allgather_exchange','playlist_sharded')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
grep -B3 -A8 Traceback $o/r02f_bench_gloo2.err | head -30
