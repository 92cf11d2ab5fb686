#!/bin/bash
# round 6, call l: the bucket pass with the next border loaded one bucket ahead, against the variant without (same box, alternating), msm tests first
REPO=$PWD; mkdir -p gpurun_out/r6l
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc" -x 2>&1 | tail -2
{
for i in 1 2 3; do
  echo "== shipped (border ahead)"; timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
  echo "== variant -DS2K_MSM_BORDER_AHEAD=0"; S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.noba.so timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r6l/border_ab.txt
cd /tmp && export TMPDIR=/tmp; cd $REPO
for V in shipped noba; do
  L=""; [ $V = noba ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.noba.so
  O=$REPO/gpurun_out/r6l/pmc_$V
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== $V"; python3 tools/summarize_pmc.py $O | grep "k_msm_accumulate"; rm -rf $O
  O=$REPO/gpurun_out/r6l/trace_$V
  S2K_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  python3 tools/msm_timeline.py $O | grep "accumulate\|span"; rm -rf $O
done | tee gpurun_out/r6l/pmc.txt
