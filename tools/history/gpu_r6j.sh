#!/bin/bash
# round 6, call j: two multiscalar calls in flight, this tree against round 5's (same box)
for i in 1 2; do
S2K_PKG_ROOT=$PWD/tmp_r05 timeout 300 python3 tools/two_calls_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python3 tools/two_calls_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
done
