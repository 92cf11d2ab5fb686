#!/bin/bash
# wide joint tables with 64-byte entries: tests of the wide layouts, then rates and traffic
timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x -k "keyset or key_sets" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
for rep in 1 2; do
for opt in "keyset5" "keyset6 --keys-log2 13"; do
  timeout 300 $B --key-grouping $opt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$opt]', 'ms_per_step=%.3f kernel_ms=%.3f clock=%.0f at_clock=%.3f' % (d['ms_per_step'], r['kernel_ms'], r['shader_clock_mhz'], r.get('frac_at_measured_clock',0)))"
done
done
