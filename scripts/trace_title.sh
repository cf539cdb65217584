# kernel timeline of a few titled launches: does title_features overlap the DAE's preamble?
R=${GRAFT_REPO_ROOT:-$PWD}
export DAE_LIB_AB=$R/scripts/probe/libdae_hip_exp.so
cd /tmp && export TMPDIR=/tmp
for v in "X=0" "DAE_TITLE_SIDE=1"; do
  rm -rf /tmp/tt; env $v rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o t -- python $R/scripts/time_title.py exact_bf16 20 > /tmp/tt.log 2>&1
  f=$(find /tmp/tt -name "*kernel_trace.csv" | head -1)
  echo "== $v"; grep "titled" /tmp/tt.log
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last launch: from the last title_features to the next topk
idx=[i for i,r in enumerate(rows) if "title_features" in r["Kernel_Name"]]
i0=idx[-2]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0-1:i0+22]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:40]
    print("%-42s start %8.1f us  dur %7.1f  queue %s" % (n,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id","?")))
PY
done
