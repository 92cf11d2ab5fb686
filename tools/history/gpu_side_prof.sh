#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/side_prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 tools/side_paths_prof.py > $O/out.txt 2>&1
grep "ms," $O/out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/side_prof/run_kernel_stats.csv")))
for r in rows[:32]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(4), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
