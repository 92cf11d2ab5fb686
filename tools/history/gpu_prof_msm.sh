#!/bin/bash
REPO=$PWD; O=$REPO/gpurun_out/prof_msm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 tools/profile_msm.py > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_msm/run_kernel_stats.csv")))
for r in rows[:30]:
    print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
PY
