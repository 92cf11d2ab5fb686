#!/bin/bash
# round 6, call ad: the ladder over per-key tables with the next entry in flight (147 VGPRs: three waves per SIMD) against the shipped form (126 VGPRs: four waves), same box
REPO=$PWD; mkdir -p gpurun_out/r7d
timeout 900 python -m pytest tests/test_gpu_keyed.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do for V in prefetch nokp; do
  L=""; [ $V = nokp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nokp.so
  S2K_LIB=$L timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-pcie --steps 30 --warmup 8 --full > gpurun_out/r7d/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r7d/b.json')); r=d['roofline']
print('$V', 'ms_per_step', round(d['ms_per_step'],4), 'ladder', round(r['kernel_ms'],4), 'clock', round(r['shader_clock_mhz']), 'cycles(M)', round(r['kernel_ms']*r['shader_clock_mhz']/1e3,3))"
done; done | tee gpurun_out/r7d/ab.txt
