#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pl -- python3 $GRAFT_REPO_ROOT/tools/pipeline_trace.py 16 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
tail -2 $GRAFT_REPO_ROOT/$O/run.log; ls $GRAFT_REPO_ROOT/$O/prof
