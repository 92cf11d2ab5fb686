#!/bin/bash
# round 6, call c: the two-part multiscalar flow - results first (msm tests), then the time by split window and by workgroups per CU of the lower part
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -q -m gpu -k "msm or rlc or schnorr_bisect or multi_scalar" -x > gpurun_out/r6c/pytest_msm.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r6c/pytest_msm.log
{
for W in 0 1 2 3 4; do for B in 3 2 4; do
  echo "== split window $W, lower part $B workgroups per CU"; S2K_MSM_SPLIT_WINDOW=$W S2K_MSM_B_WGS=$B timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
  [ $W = 0 ] && break
done; done
} | tee gpurun_out/r6c/msm_split_ab.txt
