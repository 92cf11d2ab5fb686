#!/bin/bash
python3 tools/stress_msm.py 80 1 2>&1 | grep -v amdgpu.ids | tail -2
S2K_MSM_WIDE_PAIRS=1 python3 tools/stress_msm.py 40 2 2>&1 | grep -v amdgpu.ids | tail -2
S2K_MSM_SPLIT_WINDOW=3 python3 tools/stress_msm.py 40 3 2>&1 | grep -v amdgpu.ids | tail -2
S2K_MSM_LANES=4096 python3 tools/stress_msm.py 40 4 2>&1 | grep -v amdgpu.ids | tail -2
S2K_MSM_CHUNK_LOG2=5 python3 tools/stress_msm.py 30 5 2>&1 | grep -v amdgpu.ids | tail -2
