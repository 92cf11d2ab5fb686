#!/bin/bash
# kernel times of the encoded entry point (k_parse_encoded beside the verification kernels), 2^20 DER signatures with 65-byte keys
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/parse_prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 tools/encoded_bench.py > $O/out.txt 2>&1
tail -3 $O/out.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/parse_prof/run_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
