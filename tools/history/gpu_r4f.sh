#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pl -- python3 $GRAFT_REPO_ROOT/tools/pipeline_trace.py 16 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cat $GRAFT_REPO_ROOT/$O/run.log | tail -3
python3 $GRAFT_REPO_ROOT/tools/summarize_pipeline_trace.py $GRAFT_REPO_ROOT/$O/prof/pl_kernel_trace.csv
