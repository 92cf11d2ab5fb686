#!/bin/bash
# round 6, call ac: the 4-bit joint key-set ladder with the entry prefetch (142 VGPRs) against the variant without (150), same box; key-set tests first
REPO=$PWD; mkdir -p gpurun_out/r7c
timeout 1200 python -m pytest tests -q -m gpu -k "keyset" -x 2>&1 | tail -2
for i in 1 2 3; do for V in shipped nojp; do
  L=""; [ $V = nojp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nojp.so
  S2K_LIB=$L timeout 600 python3 bench.py --key-grouping keyset --no-extras --no-cpu-baseline --no-pcie --steps 30 --warmup 8 --full > gpurun_out/r7c/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r7c/b.json')); r=d['roofline']
print('$V', 'ms_per_step', round(d['ms_per_step'],4), r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'clock', round(r['shader_clock_mhz']), 'cycles(M)', round(r['kernel_ms']*r['shader_clock_mhz']/1e3,3), 'frac', round(r.get('frac',0),3), 'at clock', round(r.get('frac_at_measured_clock',0),3))"
done; done | tee gpurun_out/r7c/ab.txt
