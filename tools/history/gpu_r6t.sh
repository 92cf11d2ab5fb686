#!/bin/bash
# round 6, call t: k_generator_part with two table entries in flight against one (same box, alternating): the step, the key-set call, the kernel under the counters
REPO=$PWD; mkdir -p gpurun_out/r6t; cd /tmp && export TMPDIR=/tmp; cd $REPO
for i in 1 2 3; do for V in shipped gp1; do
  L=""; [ $V = gp1 ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.gp1.so
  for G in auto keyset5; do
    S2K_LIB=$L timeout 600 python3 bench.py --key-grouping $G --no-extras --no-cpu-baseline --no-pcie --steps 30 --warmup 8 --full > gpurun_out/r6t/b.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/r6t/b.json')); r=d['roofline']
print('$V', '$G', 'ms_per_step', round(d['ms_per_step'],4), 'ladder', round(r['kernel_ms'],4), 'stages', {k: round(v,3) for k,v in r['stages_ms'].items() if v})"
  done
done; done | tee gpurun_out/r6t/ab.txt
for V in shipped gp1; do
  L=""; [ $V = gp1 ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.gp1.so
  O=$REPO/gpurun_out/r6t/pmc_$V
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o run -- python3 bench.py --key-grouping keyset5 --no-extras --no-cpu-baseline --no-pcie --steps 4 --warmup 1 > $O.log 2>&1
  echo "== $V (key-set call: the generator part runs beside the sort only)"; python3 tools/summarize_pmc.py $O | grep "k_generator_part"; rm -rf $O
done | tee gpurun_out/r6t/pmc.txt
