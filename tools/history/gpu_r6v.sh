#!/bin/bash
# round 6, call v: k_msm_stitch at four waves per SIMD (128 VGPRs, 29 spilled) against three (166 VGPRs): kernel time from traces, call time
REPO=$PWD; mkdir -p gpurun_out/r6v; cd /tmp && export TMPDIR=/tmp; cd $REPO
for i in 1 2 3; do for V in shipped st3; do
  L=""; [ $V = st3 ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.st3.so
  O=$REPO/gpurun_out/r6v/trace
  S2K_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 8 > $O.log 2>&1
  echo "== $V"; python3 tools/msm_timeline.py $O | grep "stitch \|fold\|span"; rm -rf $O
done; done | tee gpurun_out/r6v/stitch_ab.txt
