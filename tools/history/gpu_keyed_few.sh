#!/bin/bash
# few distinct keys: same-address atomics in k_key_insert
timeout 600 python -m pytest tests/test_gpu_keyed.py -m gpu -q -x 2>&1 | tail -3
PROBE_MODES=off,auto timeout 300 python tools/keyed_probe.py 20 0,2,6,16,20 2>/dev/null | cut -c1-330
