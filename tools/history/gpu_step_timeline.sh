#!/bin/bash
# kernel-by-kernel timeline of one verification step of the bench (rocprofv3 --kernel-trace): start, duration, stream
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/step_trace -o run -- python3 $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-extras > /dev/null 2>&1
cd $OLDPWD
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/step_trace/run_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_pack_valid" in r["Kernel_Name"]]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b+1]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:60]
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print("%-62s start %9.1f dur %8.1f us  queue %s" % (n,(st-t0)/1e3,(en-st)/1e3, r.get("Queue_Id","?")))
PY
