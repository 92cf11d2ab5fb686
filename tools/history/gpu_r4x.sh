#!/bin/bash
# wide joint tables: tests of all layouts, then their rates
O=gpurun_out/r4x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "keyset" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
