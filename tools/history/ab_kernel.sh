#!/bin/bash
# Same-box A/B of compile-time variants of the hot kernel:  bash tools/ab_kernel.sh "<flags A>" "<flags B>" ...
# For each flag set: rebuild (S2K_EXTRA_FLAGS), run the ECDSA random-batch and exceptional-case parity tests,
# then the bench three times without extras; prints ms_per_step, the ladder kernel's HIP-event time and clock.
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
for v in "$@"; do
  S2K_EXTRA_FLAGS="$v" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1 || { echo "[$v] BUILD FAILED"; continue; }
  bash tools/kernel_regs.sh | grep "k_verify_fastILi0" | sed "s/^/[$v] /"
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_hotpath.py -m gpu -q -x -k "random_batches or exceptional or wycheproof_ecdsa or ladder" 2>&1 | tail -1 | sed "s/^/[$v] /"
  for rep in 1 2 3; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'ms_per_step=%.3f kernel_ms=%.3f median=%.3f clock=%.0f' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['shader_clock_mhz']))"
  done
done
