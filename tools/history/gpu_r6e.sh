#!/bin/bash
# round 6, call e: the front end of the multiscalar call in two kernels (points beside the sort) and the one-launch scan: results, then A/B
REPO=$PWD; mkdir -p gpurun_out/r6e
timeout 1500 python -m pytest tests -q -m gpu -k "msm or rlc or schnorr_bisect or multi_scalar or group_whole" -x > gpurun_out/r6e/pytest_msm.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r6e/pytest_msm.log
{
echo "== shipped"; timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
echo "== S2K_MSM_PARSE1=1 (one front-end kernel)"; S2K_MSM_PARSE1=1 timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
echo "== S2K_MSM_SCAN3=1 (three-launch scan)"; S2K_MSM_SCAN3=1 timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
echo "== both old"; S2K_MSM_PARSE1=1 S2K_MSM_SCAN3=1 timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
echo "== shipped again"; timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r6e/front_ab.txt
cd /tmp && export TMPDIR=/tmp; cd $REPO
O=$REPO/gpurun_out/r6e/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
python3 tools/msm_timeline.py $O | tee $REPO/gpurun_out/r6e/timeline.txt; rm -rf $O
timeout 600 python3 tools/stress_msm.py 60 61 2>&1 | grep -v amdgpu.ids | tail -2
