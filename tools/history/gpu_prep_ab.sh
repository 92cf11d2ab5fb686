#!/bin/bash
# signatures sharing one inversion in k_scalar_prep (S2K_PREP_M): 4 / 6 (shipped) / 8, on the grouped step and the key-set call
REPO=$PWD
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
for rep in 1 2; do
for v in ${VARIANTS:-default prep4 prep8}; do
  if [ "$v" = default ]; then unset S2K_LIB; else export S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.$v.so; fi
  for opt in "keyset5" "auto"; do
    timeout 300 $B --key-grouping $opt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v $opt]', 'ms_per_step=%.3f ladder=%.3f' % (d['ms_per_step'], r['kernel_ms']))"
  done
done
done
