#!/bin/bash
# same-box A/B of library variants on the 2^16-key batch: S2K_LIB=<path> per variant, default library last
for rep in 1 2; do
  for lib in ${VARIANTS:-} default; do
    if [ "$lib" = default ]; then unset S2K_LIB; else export S2K_LIB=$PWD/secp256k1_voi_amd/$lib; fi
    PROBE_MODES=auto timeout 300 python tools/keyed_probe.py 20 16 2>/dev/null | tail -1 | cut -c1-200 | sed "s|^|$lib |"
  done
done
