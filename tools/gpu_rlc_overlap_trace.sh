#!/bin/bash
# kernel + memory-copy timeline of two verifiers on two threads running the BIP-340 whole-batch check from page-locked memory
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/rlc_trace; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o run -- python3 tools/rlc_two_threads_probe.py 20 3 > $O/out.txt 2>&1
tail -2 $O/out.txt
python3 - <<'PY'
import csv, glob
k=list(csv.DictReader(open("gpurun_out/rlc_trace/run_kernel_trace.csv")))
m=list(csv.DictReader(open(glob.glob("gpurun_out/rlc_trace/run_memory_copy_trace.csv")[0])))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K q%s %s"%(r.get("Queue_Id"),r["Kernel_Name"].split("(")[0][-40:])) for r in k]
ev+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"C %s %s B"%(r.get("Direction"),r.get("Bytes","?"))) for r in m]
ev.sort()
# the last 2-thread rlc phase: find the last big H2D copies
big=[e for e in ev if e[2].startswith("C") and "HOST_TO_DEVICE" in e[2] and int(e[2].split()[-2])>=30000000]
t_end=big[-1][0]
sel=[e for e in ev if t_end-12e6<=e[0]<=t_end+6e6 and (e[1]-e[0])>40000]
t0=sel[0][0]
for s,e,n in sel: print("%9.1f %8.1f us  %s"%((s-t0)/1e3,(e-s)/1e3,n))
PY
