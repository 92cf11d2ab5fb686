#!/bin/bash
# round 6, call w: the multiscalar tests and times after k_msm_stitch went back to three waves per SIMD in the one-part flow
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc or group_whole" -x 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; done
S2K_MSM_SPLIT_WINDOW=2 timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
S2K_MSM_SPLIT_WINDOW=2 timeout 600 python3 tools/stress_msm.py 60 621 2>&1 | grep -v amdgpu.ids | tail -1
