#!/bin/bash
# Run on the GPU box: the repeated-key tests, stage timings by signatures per key, per-kernel trace.
O=gpurun_out/keyed; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_keyed.py tests/test_gpu_hotpath.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/keyed_probe.py 20 10,16,18,20 > $O/probe.jsonl 2> $O/probe.err; echo "probe rc=$?"; cat $O/probe.jsonl; tail -3 $O/probe.err
for pc in 40 60 80 100; do S2K_GP_FIRST_PERCENT=$pc PROBE_MODES=auto timeout 300 python tools/keyed_probe.py 20 16 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/pc=$pc /"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/keyed_probe.py 20 16,20 > /dev/null 2>&1
