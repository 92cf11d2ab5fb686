#!/bin/bash
export PROBE_MODES=auto
for V in default w3; do
  for P in 1 2; do
    if [ "$V" != default ]; then export S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.$V.so; else unset S2K_LIB; fi
    echo "== $V parts=$P"
    S2K_KEYED_PARTS=$P python3 tools/keyed_probe.py 20 16,17 2>&1 | grep '"mode"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys 2^%d: %.3f ms  stages %s' % (d['keys_log2'], d['ms'], {k: round(v, 2) for k, v in d['stages_ms'].items()}))"
  done
done
