#!/bin/bash
# round 6, call p: long randomised runs of the final tree (other seeds than the evidence pass), all against the oracle
mkdir -p gpurun_out/r6p
{
echo "tools/gpu_r6p.sh (long runs, final tree):"
echo "== tools/stress_small.py 1200 601"; timeout 2400 python3 tools/stress_small.py 1200 601 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_keyed.py 250 602"; timeout 2400 python3 tools/stress_keyed.py 250 602 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_pipeline.py 250 603"; timeout 2400 python3 tools/stress_pipeline.py 250 603 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_msm.py 1500 604"; timeout 2400 python3 tools/stress_msm.py 1500 604 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== S2K_MSM_SPLIT_WINDOW=2 S2K_MSM_B_WGS=3 tools/stress_msm.py 600 605"; S2K_MSM_SPLIT_WINDOW=2 S2K_MSM_B_WGS=3 timeout 2400 python3 tools/stress_msm.py 600 605 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== S2K_MSM_SPLIT_WINDOW=5 tools/stress_msm.py 600 606"; S2K_MSM_SPLIT_WINDOW=5 timeout 2400 python3 tools/stress_msm.py 600 606 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_rlc.py 300 607"; timeout 2400 python3 tools/stress_rlc.py 300 607 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/big_device_batch_check.py"; timeout 1800 python3 tools/big_device_batch_check.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
} | tee gpurun_out/r6p/stress_long.txt
