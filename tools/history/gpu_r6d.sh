#!/bin/bash
# round 6, call d: kernel timelines of the two-part multiscalar flow
REPO=$PWD; mkdir -p gpurun_out/r6d; cd /tmp && export TMPDIR=/tmp; cd $REPO
for CFG in ${CFGS:-"0 2" "2 2" "2 3" "3 2"}; do set -- $CFG
  O=$REPO/gpurun_out/r6d/trace_w$1_b$2
  S2K_MSM_SPLIT_WINDOW=$1 S2K_MSM_B_WGS=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== split window $1, lower part $2 workgroups per CU"; python3 tools/msm_timeline.py $O | tee $REPO/gpurun_out/r6d/timeline_w$1_b$2.txt
  rm -rf $O
  S2K_MSM_SPLIT_WINDOW=$1 S2K_MSM_B_WGS=$2 timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
done
