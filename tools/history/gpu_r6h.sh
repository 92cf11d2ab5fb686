#!/bin/bash
# round 6, call h: what the waves of the headline ladder (k_verify_fast<ECDSA_KEYED>) and of the general ladder wait for
REPO=$PWD; mkdir -p gpurun_out/r6h; cd /tmp && export TMPDIR=/tmp; cd $REPO
for G in auto off; do
  for P in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    O=$REPO/gpurun_out/r6h/pmc_${G}_$(echo $P | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O -o run -- python3 bench.py --key-grouping $G --no-extras --no-cpu-baseline --no-pcie --steps 4 --warmup 1 > $O.log 2>&1
    echo "== key grouping $G: $P"; python3 tools/summarize_pmc.py $O | grep "k_verify_fast\|k_generator_part\|k_key_" ; rm -rf $O
  done
done | tee gpurun_out/r6h/pmc.txt
