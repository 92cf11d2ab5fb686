#!/bin/bash
# round 6, call y: the affine result's inversion of the multiscalar call on the scalar unit, against the vector-lane form (same box, alternating)
REPO=$PWD; mkdir -p gpurun_out/r6y; cd /tmp && export TMPDIR=/tmp; cd $REPO
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc or group_whole or bisect" -x 2>&1 | tail -2
timeout 600 python3 tools/stress_msm.py 150 631 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2 3; do for V in shipped vinv; do
  L=""; [ $V = vinv ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.vinv.so
  O=$REPO/gpurun_out/r6y/trace
  S2K_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 8 > $O.log 2>&1
  echo "== $V"; python3 tools/msm_timeline.py $O | grep "final16\|span"; rm -rf $O
done; done | tee gpurun_out/r6y/inv_ab.txt
