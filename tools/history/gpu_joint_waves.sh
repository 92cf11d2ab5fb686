#!/bin/bash
# same-box A/B of the joint-table key-set ladder at three and at four waves per SIMD (S2K_JOINT_WAVES; prebuilt variants)
REPO=$PWD
for rep in 1 2; do
for v in default j3; do
  if [ "$v" = default ]; then unset S2K_LIB; else export S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras --key-grouping keyset 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'ms_per_step=%.3f kernel_ms=%.3f median=%.3f clock=%.0f frac=%.3f build=%s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['shader_clock_mhz'], r['frac'], d['config']['build']))"
done
done
unset S2K_LIB
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x -k "keyset" 2>&1 | tail -2
