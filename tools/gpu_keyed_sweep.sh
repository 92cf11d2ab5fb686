#!/bin/bash
# same-box sweep of the generator-part split (S2K_GP_FIRST_PERCENT), 2^20 signatures of 2^16 keys
for rep in 1 2; do for pc in 45 60 80 100; do S2K_GP_FIRST_PERCENT=$pc PROBE_MODES=auto timeout 300 python tools/keyed_probe.py 20 16 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/pc=$pc /"; done; done
