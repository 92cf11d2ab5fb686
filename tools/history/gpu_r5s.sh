#!/bin/bash
# round 5, call s: VALU instruction counts of the side ladders and of the five-wave small-call kernels (tools/side_counts.py)
O=$PWD/gpurun_out/r5s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/side_pmc -o run -- python3 tools/side_counts.py run > $O/side_run.log 2>&1
python3 tools/side_counts.py summarize $O/side_pmc > $O/side_counts.json; cat $O/side_counts.json
rm -rf $O/side_pmc
