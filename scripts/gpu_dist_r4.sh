# round 4: the N > 1 code paths on one GPU -- simulated ranks (compute only), 1-rank RCCL, 2-rank gloo rehearsal
cd $GRAFT_REPO_ROOT
o=gpurun_out; mkdir -p $o
X="--no-train-row --no-cpu-baseline --no-bf16-row --no-extra-rows"
for n in 2 4 8; do python bench.py --sim-world $n $X > $o/r04_bench_sim$n.json 2> $o/err_sim$n.txt; done
python bench.py --sim-world 8 --sim-rank 7 $X > $o/r04_bench_sim8_rank7.json 2> $o/err_sim8r7.txt
python bench.py --sim-world 8 --no-tau-exchange $X > $o/r04_bench_sim8_notau.json 2> $o/err_sim8notau.txt
python bench.py --sim-world 8 --sim-rank 7 --no-tau-exchange $X > $o/r04_bench_sim8_rank7_notau.json 2> $o/err_sim8r7notau.txt
python bench.py --sim-world 8 --dtype bf16 $X > $o/r04_bench_sim8_bf16.json 2> $o/err_sim8bf.txt
python bench.py --force-dist $X > $o/r04_bench_forcedist.json 2> $o/err_forcedist.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --steps 20 --warmup 4 $X > $o/r04_bench_gloo2.json 2> $o/err_gloo2.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --backend gloo --steps 20 --warmup 4 --no-tau-exchange $X > $o/r04_bench_gloo2_notau.json 2> $o/err_gloo2notau.txt
for f in sim2 sim4 sim8 sim8_rank7 sim8_notau sim8_rank7_notau sim8_bf16 forcedist gloo2 gloo2_notau; do python - $o/r04_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    ex={k:{a:b for a,b in v.items() if a!='note'} for k,v in d.items() if k.endswith('_exchange') or k in ('playlist_sharded','collective')}
    print("%-15s value=%10.0f ms=%.4f kern=%.4f frac=%.3f %s phases=%s" % (sys.argv[2], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], ex, {k_: v_ for k_, v_ in (d.get("phases") or {}).items() if k_.endswith("_ms")}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
for f in $o/err_*.txt; do if grep -q Traceback $f; then echo "== $f"; grep -A12 Traceback $f | head -30; fi; done
