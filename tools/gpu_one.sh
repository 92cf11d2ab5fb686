#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_keyed.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do PROBE_MODES=off,auto timeout 300 python tools/keyed_probe.py 20 0,2,16,20 2>/dev/null | cut -c1-200; done
