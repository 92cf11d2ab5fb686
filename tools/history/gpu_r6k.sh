#!/bin/bash
# round 6, call k: what the waves of the bucket pass (k_msm_accumulate) wait for
REPO=$PWD; mkdir -p gpurun_out/r6k; cd /tmp && export TMPDIR=/tmp; cd $REPO
for P in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
  O=$REPO/gpurun_out/r6k/pmc_$(echo $P | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O -o run -- python3 tools/profile_msm.py msm 6 > $O.log 2>&1
  echo "== $P"; python3 tools/summarize_pmc.py $O | grep "k_msm_accumulate\|k_msm_stitch \|k_msm_fold" ; rm -rf $O
done | tee gpurun_out/r6k/pmc.txt
