#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -m gpu -k "msm or rlc or bisect" 2>&1 | tail -2
for W in 0 1 2 3 4 6; do echo "== split window $W"; S2K_MSM_SPLIT_WINDOW=$W python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; done
