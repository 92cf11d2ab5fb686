#!/bin/bash
# A/B of per-signature table layouts on the GPU box: rebuilds engine.o per variant and runs the bench.
# whatever variant ran last, leave the DEFAULT build behind (build() also rebuilds when the recorded flags differ)
trap 'S2K_EXTRA_FLAGS="" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1' EXIT
set -e
for v in "-DS2K_QT_PLANE=8 -DS2K_STRIDE_PAD=1088" "-DS2K_QT_PLANE=16 -DS2K_STRIDE_PAD=1088" "-DS2K_QT_PLANE=16 -DS2K_STRIDE_PAD=0" "-DS2K_QT_PLANE=8 -DS2K_STRIDE_PAD=64" "-DS2K_QT_PLANE=8 -DS2K_STRIDE_PAD=4160" "-DS2K_QT_PLANE=8 -DS2K_STRIDE_PAD=16448"; do
  S2K_EXTRA_FLAGS="$v" python -c "import secp256k1_voi_amd as S; S.build(force=True)" > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_batches or exceptional" 2>&1 | tail -1
