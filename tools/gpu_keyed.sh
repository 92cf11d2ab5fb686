#!/bin/bash
# Run on the GPU box: the repeated-key tests, stage timings by signatures per key, per-kernel trace.
O=gpurun_out/keyed; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_keyed.py tests/test_gpu_hotpath.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/keyed_probe.py 20 10,16,20 > $O/probe.jsonl 2> $O/probe.err; echo "probe rc=$?"; cat $O/probe.jsonl; tail -3 $O/probe.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/keyed_probe.py 20 16,20 > /dev/null 2>&1
