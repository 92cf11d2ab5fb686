#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_keyed.py -m gpu -q -x 2>&1 | tail -2
PROBE_MODES=off,auto timeout 300 python tools/keyed_probe.py 20 209716,174763,131072 2>/dev/null | cut -c1-260
