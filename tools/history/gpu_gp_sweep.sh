#!/bin/bash
# share of the generator part launched beside the key chain (the rest runs after k_key_odd): sweep on one box
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
for rep in 1 2; do
for pc in 40 60 75 90 100; do
  S2K_GP_FIRST_PERCENT=$pc timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[gp_first=$pc]', 'ms_per_step=%.3f ladder=%.3f clock=%.0f' % (d['ms_per_step'], r['kernel_ms'], r['shader_clock_mhz']))"
done
done
