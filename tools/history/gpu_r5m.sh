#!/bin/bash
# round 5, call m: parity of the three wave-per-signature ladders, the small-batch probe, its kernel trace
mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "small_batch" 2>&1 | tail -15
timeout 300 python tools/small_batch_probe.py > gpurun_out/r5m/small_batch_ab.txt 2>&1
echo "probe rc=$?"; grep log2_n gpurun_out/r5m/small_batch_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5m/trace -o small -- python3 $GRAFT_REPO_ROOT/tools/small_batch_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r5m/probe_traced.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r5m/trace/**/*kernel_trace.csv', recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm = r['Kernel_Name'].split('(')[0][:40]
    by[(nm, int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = open('gpurun_out/r5m/kernel_medians.txt', 'w')
for k, v in sorted(by.items()):
    v.sort(); line = "%-42s grid %8d  calls %4d  median %9.1f us" % (k[0], k[1], len(v), v[len(v)//2]); out.write(line + "\n")
    if "_row" in k[0] or "prep" in k[0] or "finish" in k[0]: print(line)
PY
rm -rf gpurun_out/r5m/trace
