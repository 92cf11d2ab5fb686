#!/bin/bash
# round 5, call t: randomised long runs of the last tree (five-wave small-call kernels, host calls without DMA transfers)
O=gpurun_out/r5t; mkdir -p $O
{
echo "tools/gpu_r5t.sh, the tree with the five-wave small-call kernels and the DMA-free host calls:"
echo "== tools/stress_small.py 800 101"; timeout 2400 python3 tools/stress_small.py 800 101 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== S2K_SMALL_CALLS_DMA=1 tools/stress_small.py 150 102"; S2K_SMALL_CALLS_DMA=1 timeout 1200 python3 tools/stress_small.py 150 102 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_keyed.py 100 103"; timeout 2400 python3 tools/stress_keyed.py 100 103 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_pipeline.py 100 104"; timeout 2400 python3 tools/stress_pipeline.py 100 104 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
} | tee $O/stress.txt
