# usage: bash scripts/gpu_bisect.sh  -> kernel averages (us) of the 1-stream default step under a few debug switches
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; rm -rf /tmp/bs_$tag; env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-row --no-bf16-row --streams 1 --steps 60 --prime-ms 50 > /tmp/bs_$tag.log 2>&1
  f=$(find /tmp/bs_$tag -name "*kernel_stats.csv" | head -1)
  python - "$f" "$tag" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2], " | ".join("%s %.1f" % (r["Name"].split("(")[0].split("::")[-1][:34], float(r["AverageNs"])/1e3) for r in rows[:6]))
PY
}
run base A=1

for n in 1 2 3 4 5; do run topkstop$n DAE_TOPK_STOP=-$n; done
