#!/bin/bash
# round 6, call ae: the evidence pass again after the ladder over per-key tables got its entry prefetch (instruction counts of the shipped kernels, traces, side counts)
REPO=$PWD
bash tools/collect_profiles_r04.sh r06x > gpurun_out/collect_r06x.log 2>&1; tail -3 gpurun_out/collect_r06x.log
cd /tmp && export TMPDIR=/tmp; cd $REPO
O=$REPO/gpurun_out/side_r06x
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o run -- python3 tools/side_counts.py run > $O.log 2>&1
python3 tools/side_counts.py summarize $O > gpurun_out/profiles_r06x/side_counts.json; head -c 600 gpurun_out/profiles_r06x/side_counts.json; rm -rf $O
