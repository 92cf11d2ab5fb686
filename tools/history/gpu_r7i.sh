#!/bin/bash
# round 6, call ai: negative entries of the bucket pass negated on the limbs (nine instructions) against the eight-word add-with-carry chain: tests, stress, time (alternating)
REPO=$PWD; mkdir -p gpurun_out/r7i
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc or group_whole or bisect" -x 2>&1 | tail -2
timeout 900 python3 tools/stress_msm.py 400 651 2>&1 | grep -v amdgpu.ids | tail -1
S2K_MSM_SPLIT_WINDOW=2 timeout 900 python3 tools/stress_msm.py 100 652 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python3 tools/stress_rlc.py 60 653 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2 3; do timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nw.so timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids | sed 's/^/words: /'; done | tee gpurun_out/r7i/time.txt
cd /tmp && export TMPDIR=/tmp; cd $REPO
for V in limbs words; do
  L=""; [ $V = words ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nw.so
  O=$REPO/gpurun_out/r7i/pmc
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== $V"; python3 tools/summarize_pmc.py $O | grep "k_msm_accumulate"; rm -rf $O
done | tee gpurun_out/r7i/pmc.txt
