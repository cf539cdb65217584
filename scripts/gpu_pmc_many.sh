# Counter passes (FETCH_SIZE / WRITE_SIZE, one pass each) on the launches of many rows: scripts/time_modes.py <B> zipf exact 1.
# usage: bash scripts/gpu_pmc_many.sh [B]   -> prints per-kernel means (KB; hbm bytes = 2 x FETCH + WRITE on gfx950) and keeps the CSVs
B=${1:-2048}
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=32
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcm_${B}_$c; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $root/scripts/time_modes.py $B zipf exact 1 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $root/gpurun_out/r05_pmc_many${B}_$c.csv
done
python - $root/gpurun_out/r05_pmc_many${B}_FETCH_SIZE.csv $root/gpurun_out/r05_pmc_many${B}_WRITE_SIZE.csv <<'PY'
import csv, sys, collections
def load(p, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            d[k].append(float(r["Counter_Value"]))
    return d
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
for k in sorted(f, key=lambda k: -sum(f[k]) / len(f[k])):
    fm = sum(f[k]) / len(f[k]); wm = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    if len(f[k]) >= 100:
        print("%-48s launches %4d  FETCH %9.1f KB  WRITE %9.1f KB  hbm-side %8.2f MB" % (k[:48], len(f[k]), fm, wm, (2 * fm + wm) / 1024))
PY
