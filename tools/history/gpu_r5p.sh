#!/bin/bash
# round 5, call p: longer randomised runs of the final tree with other seeds
O=gpurun_out/r5p; mkdir -p $O
{
echo "tools/gpu_r5p.sh, other seeds:"
echo "== tools/stress_small.py 600 71"; timeout 2400 python3 tools/stress_small.py 600 71 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_small.py 300 72"; timeout 1800 python3 tools/stress_small.py 300 72 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_keyed.py 200 73"; timeout 2400 python3 tools/stress_keyed.py 200 73 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_pipeline.py 150 74"; timeout 2400 python3 tools/stress_pipeline.py 150 74 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_msm.py 300 75"; timeout 1800 python3 tools/stress_msm.py 300 75 2>&1 | grep -v amdgpu.ids | tail -1
} | tee $O/stress.txt
