#!/bin/bash
# width of the generator windows for key-set callers (where the generator part is on the critical path): 22 (shipped), 24, 26 bits
# prebuilt variants: python -c "import secp256k1_voi_amd as S; S.build_variant('gt24', '-DS2K_GT_BITS=24'); S.build_variant('gt26', '-DS2K_GT_BITS=26')"
REPO=$PWD
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
for rep in 1 2; do
for v in default gt24 gt26; do
  if [ "$v" = default ]; then unset S2K_LIB; else export S2K_LIB=$REPO/secp256k1_voi_amd/libsecp256k1_voi_amd.$v.so; fi
  for opt in "keyset5" "auto"; do
    timeout 300 $B --key-grouping $opt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v $opt]', 'ms_per_step=%.3f ladder=%.3f' % (d['ms_per_step'], r['kernel_ms']), d['config']['build'][:12])"
  done
done
done
