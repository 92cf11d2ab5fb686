#!/bin/bash
# round 6, call aa: k_generator_part at four waves per SIMD (128 VGPRs, 6 spilled: 2^20 signatures are exactly four rounds of waves) against three (133 VGPRs), shares 50 / 60 %
REPO=$PWD; mkdir -p gpurun_out/r7a
for i in 1 2 3; do for V in shipped gp4 gp4_50; do
  L=""; P=60; [ $V != shipped ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.gp4.so; [ $V = gp4_50 ] && P=50
  S2K_GP_FIRST_PERCENT=$P S2K_LIB=$L timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-pcie --steps 30 --warmup 8 --full > gpurun_out/r7a/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r7a/b.json')); r=d['roofline']
print('$V', 'ms_per_step', round(d['ms_per_step'],4), 'ladder', round(r['kernel_ms'],4), 'clock', round(r['shader_clock_mhz']), 'stages', {k: round(v,3) for k,v in r['stages_ms'].items() if v})"
done; done | tee gpurun_out/r7a/ab.txt
