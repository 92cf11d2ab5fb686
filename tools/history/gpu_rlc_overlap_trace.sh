#!/bin/bash
# kernel + memory-copy timeline of two verifiers on two threads, started half a period apart, running a whole-batch call
# (PROBE_TRACE=rlc | msm) from page-locked memory: the last 12 ms of the run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/rlc_trace; mkdir -p $O
export PROBE_TRACE=${1:-rlc}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o run -- python3 tools/rlc_two_threads_probe.py 20 4 > $O/out.txt 2>&1
grep staggered $O/out.txt
python3 - <<'PY'
import csv
k=list(csv.DictReader(open("gpurun_out/rlc_trace/run_kernel_trace.csv")))
m=list(csv.DictReader(open("gpurun_out/rlc_trace/run_memory_copy_trace.csv")))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K q%s %s"%(r.get("Queue_Id"),r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:40])) for r in k]
ev+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY %s"%r["Direction"].replace("MEMORY_COPY_","")) for r in m]
ev.sort()
t_hi=ev[-1][1]; t_lo=t_hi-12_000_000
for s,e,n in ev:
    if s>=t_lo and (e-s)>80000: print("%9.1f %8.1f us  %s"%((s-t_lo)/1e3,(e-s)/1e3,n))
PY
