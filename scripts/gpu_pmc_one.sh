# usage: bash scripts/gpu_pmc_one.sh <COUNTER> <kernel substring> <command...>  -> mean counter value per launch of that kernel
ctr=$1; sub=$2; shift 2
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_one
timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_one -o p -- "$@" > /tmp/pmc_one.log 2>&1
f=$(find /tmp/pmc_one -name "*counter_collection.csv" | head -1)
python - "$f" "$sub" "$ctr" <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[3]]
print(sys.argv[3], sys.argv[2], "launches", len(v), "mean", sum(v)/max(len(v),1))
PY
