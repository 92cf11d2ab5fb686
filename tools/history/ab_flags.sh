#!/bin/bash
# Same-box A/B of compile-time variants: bash tools/ab_flags.sh "<flags A>" "<flags B>" ...
# Rebuilds the engine with each flag set (S2K_EXTRA_FLAGS) and runs the bench twice.
# whatever variant ran last, leave the DEFAULT build behind (build() also rebuilds when the recorded flags differ)
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
for v in "$@"; do
  S2K_EXTRA_FLAGS="$v" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['ms_per_step'])"
  done
done
