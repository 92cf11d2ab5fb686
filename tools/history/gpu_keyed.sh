#!/bin/bash
# Run on the GPU box: the repeated-key tests, stage timings by signatures per key, per-kernel trace.
O=gpurun_out/keyed; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_keyed.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
S2K_KEYED_PARTS=1 timeout 600 python -m pytest tests/test_gpu_keyed.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do for parts in 1 2; do S2K_KEYED_PARTS=$parts PROBE_MODES=auto timeout 300 python tools/keyed_probe.py 20 10,16,17 2>/dev/null | cut -c1-200 | sed "s/^/parts=$parts /"; done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python3 $GRAFT_REPO_ROOT/tools/keyed_probe.py 20 16,20 > /dev/null 2>&1
