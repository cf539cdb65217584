# FETCH_SIZE per access width (r06 notes 7): scripts/probe/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -o out -- $GRAFT_REPO_ROOT/scripts/probe/fetch_calib > /tmp/fc.log 2>&1
tail -2 /tmp/fc.log
f=$(find /tmp/fc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r06_fetch_calib.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    # FETCH_SIZE is in KB (rocprofv3 derived metric); 256 MiB = 262144 KB
    print("%-62s launches %d  FETCH_SIZE avg %.0f  -> x %.3f of the 262144 KB read" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / 262144.0))
PY
