#!/bin/bash
# A/B on one box: the current library against a prebuilt variant libsecp256k1_voi_amd.$1.so (tools/keyed_probe.py,
# 2^20 signatures of 2^16 and 2^17 keys; stage times from the engine's own events)
V=${1:?variant name}
export PROBE_MODES=auto
for rep in 1 2 3; do
for L in default $V; do
  if [ "$L" != default ]; then export S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.$L.so; else unset S2K_LIB; fi
  echo "== $L"
  python3 tools/keyed_probe.py 20 16,17 2>&1 | grep '"mode"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys 2^%d: %.3f ms  stages %s' % (d['keys_log2'], d['ms'], {k: round(v, 2) for k, v in d['stages_ms'].items()}))"
done
done
