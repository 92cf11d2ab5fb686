#!/bin/bash
# round 6, call g: the 5-bit joint key-set ladder with and without the entry prefetch (same box): bench line, then what its waves wait for (PMC)
REPO=$PWD; mkdir -p gpurun_out/r6g; cd /tmp && export TMPDIR=/tmp; cd $REPO
B="bench.py --key-grouping keyset5 --no-extras --no-cpu-baseline --no-pcie --steps 20 --warmup 5 --full"
for V in default nojp default nojp; do
  L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.so; [ $V = nojp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nojp.so
  S2K_LIB=$L timeout 600 python3 $B > gpurun_out/r6g/bench_$V.json 2> gpurun_out/r6g/bench_$V.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r6g/bench_$V.json')); r=d['roofline']
print('$V', 'ms_per_step', round(d['ms_per_step'],4), 'kernel', r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r.get('frac',0),4), 'clock', round(r.get('shader_clock_mhz',0)), 'frac_at_clock', r.get('frac_at_measured_clock'))"
done | tee gpurun_out/r6g/ab.txt
for V in default nojp; do
  L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.so; [ $V = nojp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nojp.so
  for P in "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE"; do
    O=$REPO/gpurun_out/r6g/pmc_${V}_$(echo $P | cut -d' ' -f1)
    S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O -o run -- python3 bench.py --key-grouping keyset5 --no-extras --no-cpu-baseline --no-pcie --steps 4 --warmup 1 > $O.log 2>&1
    echo "== $V: $P"; python3 tools/summarize_pmc.py $O | grep "k_verify_fast<10>" ; rm -rf $O
  done
done | tee gpurun_out/r6g/pmc.txt
