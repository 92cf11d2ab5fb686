#!/bin/bash
# round 6, call s: the bucket pass with the fetched record unpacked BEFORE a piece is flushed (the wave no longer waits for its 36 stores), against the variant without
REPO=$PWD; mkdir -p gpurun_out/r6s
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc" -x 2>&1 | tail -2
{
for i in 1 2 3; do
  echo "== shipped (unpack first)"; timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
  echo "== variant -DS2K_MSM_UNPACK_FIRST=0"; S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nouf.so timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r6s/unpack_ab.txt
cd /tmp && export TMPDIR=/tmp; cd $REPO
for V in shipped nouf shipped nouf; do
  L=""; [ $V = nouf ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nouf.so
  O=$REPO/gpurun_out/r6s/pmc_$V
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== $V"; python3 tools/summarize_pmc.py $O | grep "k_msm_accumulate"; rm -rf $O
done | tee gpurun_out/r6s/pmc.txt
