#!/bin/bash
# A/B of the keyed ladder's additions on one box: XYZZ (default) against Jacobian mixed additions (variant "jadd")
export PROBE_MODES=auto
for rep in 1 2; do
for V in default jadd; do
  if [ "$V" != default ]; then export S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.$V.so; else unset S2K_LIB; fi
  echo "== $V"
  python3 tools/keyed_probe.py 20 16,17 2>&1 | grep '"mode"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys 2^%d: %.3f ms  stages %s' % (d['keys_log2'], d['ms'], {k: round(v, 2) for k, v in d['stages_ms'].items()}))"
done
done
