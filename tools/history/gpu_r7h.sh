#!/bin/bash
# round 6, call ah: flushed pieces of the bucket pass as 144-byte records (nine stores) against 36 planes (36 stores): tests, kernel traces alternating, counters
REPO=$PWD; mkdir -p gpurun_out/r7h; cd /tmp && export TMPDIR=/tmp; cd $REPO
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc or group_whole or bisect" -x 2>&1 | tail -2
timeout 600 python3 tools/stress_msm.py 200 641 2>&1 | grep -v amdgpu.ids | tail -1
for i in 1 2 3; do for V in records planes; do
  L=""; [ $V = planes ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.xpl.so
  O=$REPO/gpurun_out/r7h/trace
  S2K_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 8 > $O.log 2>&1
  echo "== $V"; python3 tools/msm_timeline.py $O | grep "accumulate\|stitch<\|span"; rm -rf $O
done; done | tee gpurun_out/r7h/ab.txt
for V in records planes; do
  L=""; [ $V = planes ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.xpl.so
  O=$REPO/gpurun_out/r7h/pmc
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== $V"; python3 tools/summarize_pmc.py $O | grep "k_msm_accumulate\|k_msm_stitch<"; rm -rf $O
done | tee gpurun_out/r7h/pmc.txt
for i in 1 2; do timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.xpl.so timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids | sed 's/^/planes: /'; done | tee gpurun_out/r7h/time.txt
