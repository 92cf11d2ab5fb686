#!/bin/bash
# round 6, call af: the ladder over per-key tables with the (unconditional) entry prefetch against the form without, alternating; the keyed tests; then the evidence pass of the final tree
REPO=$PWD; mkdir -p gpurun_out/r7f
timeout 900 python -m pytest tests/test_gpu_keyed.py tests/test_gpu_parity.py tests/test_gpu_round3.py -q -m gpu -x -k "not stress" 2>&1 | tail -2
for i in 1 2 3 4; do for V in prefetch nokp; do
  L=""; [ $V = nokp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nokp.so
  S2K_LIB=$L timeout 600 python3 bench.py --no-extras --no-cpu-baseline --no-pcie --steps 30 --warmup 8 --full > gpurun_out/r7f/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r7f/b.json')); r=d['roofline']
print('$V', 'ms_per_step', round(d['ms_per_step'],4), 'ladder', round(r['kernel_ms'],4), 'clock', round(r['shader_clock_mhz']), 'cycles(M)', round(r['kernel_ms']*r['shader_clock_mhz']/1e3,3))"
done; done | tee gpurun_out/r7f/ab.txt
bash tools/collect_profiles_r04.sh r06w > gpurun_out/collect_r06w.log 2>&1; tail -2 gpurun_out/collect_r06w.log
cd /tmp && export TMPDIR=/tmp; cd $REPO
O=$REPO/gpurun_out/side_r06w
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o run -- python3 tools/side_counts.py run > $O.log 2>&1
python3 tools/side_counts.py summarize $O > gpurun_out/profiles_r06w/side_counts.json; rm -rf $O
O=$REPO/gpurun_out/r7f/pmc
for V in prefetch nokp; do
  L=""; [ $V = nokp ] && L=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.nokp.so
  S2K_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -o run -- python3 bench.py --no-extras --no-cpu-baseline --no-pcie --steps 4 --warmup 1 > $O.log 2>&1
  echo "== $V"; python3 tools/summarize_pmc.py $O | grep "k_verify_fast<4>"; rm -rf $O
done | tee gpurun_out/r7f/pmc.txt
