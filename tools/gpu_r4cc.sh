#!/bin/bash
# key-set call: preparation + generator part in two halves on the two streams against one piece on the second stream
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras --key-grouping keyset5"
for rep in 1 2 3; do
for v in 1 0; do
  S2K_KEYSET_SPLIT_FRONT=$v timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[split=$v]', 'ms_per_step=%.3f kernel_ms=%.3f clock=%.0f' % (d['ms_per_step'], r['kernel_ms'], r['shader_clock_mhz']))"
done
done
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x -k "keyset and not schnorr" 2>&1 | tail -2
