#!/bin/bash
# Kernel-level breakdown of tools/bench_configs.py (MSM, BIP-340 per-signature, BIP-340 whole-batch).
TAG=${1:-r01x}
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg_$TAG -o run -- python3 tools/bench_configs.py > gpurun_out/prof_cfg_${TAG}.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_cfg_$TAG/run_kernel_stats.csv')))
for r in rows[:40]:
    print(f"{r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
