# the measurements DESIGN.md section 5 quotes for round 2 (one gpurun call); outputs under gpurun_out/r02f_*
cd $GRAFT_REPO_ROOT
o=gpurun_out
t0=$(date +%s.%N); python bench.py > $o/r02f_bench_default.json 2> $o/r02f_bench_default.err; echo "default bench.py wall $(( $(date +%s) - ${t0%.*} )) s"
python bench.py --dtype bf16 --no-train-row > $o/r02f_bench_bf16.json 2>/dev/null
python bench.py --batch-per-gpu 1024 --no-train-row --no-cpu-baseline > $o/r02f_bench_b1024.json 2>/dev/null
python bench.py --streams 1 --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_1stream.json 2>/dev/null
python bench.py --bias zeros --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_bias_zeros.json 2>/dev/null
for n in 2 4 8; do python bench.py --sim-world $n --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_sim$n.json 2>/dev/null; done
python bench.py --force-dist --no-train-row --no-cpu-baseline --no-bf16-row > $o/r02f_bench_forcedist.json 2> $o/r02f_bench_forcedist.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 20 --warmup 4 --no-train-row --no-cpu-baseline > $o/r02f_bench_gloo2.json 2> $o/r02f_bench_gloo2.err
bash scripts/gpu_prof.sh r02f_2streams --no-cpu-baseline --no-train-row --no-bf16-row > $o/r02f_prof_2s.log 2>&1
bash scripts/gpu_prof.sh r02f_1stream --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 > $o/r02f_prof_1s.log 2>&1
bash scripts/gpu_prof.sh r02f_bf16_1stream --no-cpu-baseline --no-train-row --dtype bf16 --streams 1 > $o/r02f_prof_bf16.log 2>&1
bash scripts/gpu_prof.sh r02f_b1024_1stream --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 --batch-per-gpu 1024 --steps 50 > $o/r02f_prof_b1024.log 2>&1
python scripts/bench_shim.py > $o/r02f_shim.log 2>&1
python scripts/bench_title.py > $o/r02f_title.log 2>&1
python scripts/bench_epoch.py > $o/r02f_epoch.log 2>&1
python scripts/bench_epoch.py --bf16 > $o/r02f_epoch_bf16.log 2>&1
for f in default bf16 b1024 1stream bias_zeros sim2 sim4 sim8 forcedist gloo2; do python - $o/r02f_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print('%-12s value=%10.0f ms=%.4f kern=%s %.4f frac=%.3f extra=%s' % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel'][:34], r['avg_launch_ms'], r['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.items() if k in ('bf16_decode','tracks_only','allgather_exchange','playlist_sharded')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
tail -3 $o/r02f_shim.log; tail -6 $o/r02f_title.log; tail -2 $o/r02f_epoch.log $o/r02f_epoch_bf16.log; tail -5 $o/r02f_bench_gloo2.err
python scripts/bench_challenge.py > $o/r02f_challenge.log 2>&1; tail -1 $o/r02f_challenge.log
