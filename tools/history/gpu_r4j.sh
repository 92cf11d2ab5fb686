#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
python tools/two_contexts_probe.py | tail -2
for rep in 1 2; do for g in 0 1; do echo "two_streams=$g"; S2K_SUBMIT_TWO_STREAMS=$g python tools/pipeline_trace.py 24 | tail -1; done; done
echo "plain one-shot:"; S2K_SUBMIT_PLAIN=1 python tools/pipeline_trace.py 24 | tail -1
echo "no gate:"; S2K_SUBMIT_NO_GATE=1 python tools/pipeline_trace.py 24 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pl -- python3 $GRAFT_REPO_ROOT/tools/pipeline_trace.py 16 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_pipeline_trace.py $GRAFT_REPO_ROOT/$O/prof/pl_kernel_trace.csv | head -4
