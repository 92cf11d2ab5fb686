#!/bin/bash
# k_msm_fold with the XCD-aware choice of column blocks against the variant in blockIdx order (-DS2K_FOLD_PLAIN): same box, alternating
REPO=$PWD; O=gpurun_out/fold_ab; mkdir -p $O
V=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.plainfold.so
timeout 900 python -m pytest tests -q -m gpu -k "msm or rlc" -x 2>&1 | tail -2
{
for i in 1 2 3 4; do
  echo "== shipped (XCD-aware)"; timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
  echo "== variant -DS2K_FOLD_PLAIN"; S2K_LIB=$V timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee $O/fold_ab.txt
cd /tmp && export TMPDIR=/tmp; cd $REPO
for W in shipped plain; do
  L=""; [ $W = plain ] && L=$V
  for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    P=$REPO/$O/pmc_$W
    S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P -o run -- python3 tools/profile_msm.py msm 6 > $P.log 2>&1
    echo "== $W $C"; python3 tools/summarize_pmc.py $P | grep "k_msm_fold\|k_msm_stitch<"; rm -rf $P
  done
  P=$REPO/$O/trace_$W
  S2K_LIB=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P -o run -- python3 tools/profile_msm.py msm 6 > $P.log 2>&1
  python3 tools/msm_timeline.py $P | grep "fold\|span"; rm -rf $P
done | tee $O/pmc.txt
