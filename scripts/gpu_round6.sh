# Round-6 session on the GPU box: tests, the driver's bench command, and the rocprofv3 kernel stats profiles/r06_notes.md quotes
# (copy gpurun_out/r06_* into profiles/ afterwards).  usage: bash scripts/gpu_round6.sh [notests]
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
o=gpurun_out; mkdir -p $o
if [ "$1" != notests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $o/r06_gputests.log
fi
python bench.py --gpus 1 --steps 20 --warmup 5 --verbose-out $o/r06_bench_verbose.json > $o/r06_bench_default.json 2> $o/err_default.txt
tail -c 7000 $o/r06_bench_default.json; echo; wc -c $o/r06_bench_default.json
bash scripts/gpu_kprof.sh r06_headline 7 python $GRAFT_REPO_ROOT/bench.py --no-extra-rows --no-train-row --no-cpu-baseline --no-bf16-row
for B in 256 1024 2048; do
  bash scripts/gpu_kprof.sh r06_exact_b${B}_1stream 8 python $GRAFT_REPO_ROOT/scripts/time_modes.py $B zipf exact 1
done
bash scripts/gpu_kprof.sh r06_bf16_b2048_1stream 8 python $GRAFT_REPO_ROOT/scripts/time_modes.py 2048 zipf bf16 1
bash scripts/gpu_kprof.sh r06_title_native 14 python $GRAFT_REPO_ROOT/scripts/time_title.py exact_bf16
bash scripts/gpu_kprof.sh r06_train_default 14 python $GRAFT_REPO_ROOT/scripts/bench_train.py --default
(python scripts/time_modes.py 256 zipf exact 1,4; python scripts/time_modes.py 1024 zipf exact 1,4) 2>&1 | grep streams= | tee $o/r06_modes.log
# four batches in flight (the overlap hint: the sample launch on its own geometry) and the loop's timeline
for B in 256 1024; do
  bash scripts/gpu_kprof.sh r06_exact_b${B}_4streams 8 python $GRAFT_REPO_ROOT/scripts/time_modes.py $B zipf exact 4
done
python scripts/bench_loop.py 256 native f32,exact_bf16,bf16 3 2>&1 | grep "playlists/s" | tee $o/r06_bench_loop.log
python scripts/bench_loop.py 150 native exact_bf16 3 2>&1 | grep "playlists/s" | tee -a $o/r06_bench_loop.log
