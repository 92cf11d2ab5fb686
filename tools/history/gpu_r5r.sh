#!/bin/bash
# round 5, call r: the wave-per-signature kernels with their preparation wave (one launch per call) - parity, randomised run, times
mkdir -p gpurun_out/r5r
timeout 1500 python -m pytest tests -x -q -m gpu -k "small_batch or recover or smoke or harness or schnorr or wycheproof or kats or random_batches or encoded or table_widths or bip340" 2>&1 | tail -3
timeout 900 python3 tools/stress_small.py 250 91 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
timeout 300 python tools/small_batch_probe.py > gpurun_out/r5r/small_batch_ab.txt 2>&1
echo "probe rc=$?"; grep log2_n gpurun_out/r5r/small_batch_ab.txt
