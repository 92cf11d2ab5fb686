#!/bin/bash
# same-box A/B of the generator window width
# whatever variant ran last, leave the DEFAULT build behind (build() also rebuilds when the recorded flags differ)
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
for b in 16 20 22 24 16; do
  S2K_EXTRA_FLAGS="-DS2K_GT_BITS=$b" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generator_table or scalar_base_mult or random_batches or wycheproof_ecdsa" 2>&1 | tail -1
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GT_BITS=$b', d['ms_per_step'])"
  done
done
