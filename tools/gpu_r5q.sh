#!/bin/bash
# round 5, call q: the small-call paths without their memsets - parity, randomised run, the probe
mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_c_harness.py -x -q -m gpu -k "small_batch or recover or smoke or harness or schnorr" 2>&1 | tail -4
timeout 900 python3 tools/stress_small.py 200 81 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
timeout 300 python tools/small_batch_probe.py > gpurun_out/r5q/small_batch_ab.txt 2>&1
echo "probe rc=$?"; grep log2_n gpurun_out/r5q/small_batch_ab.txt
