#!/bin/bash
# round 6, call b: the table-swap test on the shipped library (must pass) and on the per-launch variant (must fail at the verdicts)
mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_round6.py -q -m gpu > gpurun_out/r6b/pytest_round6.log 2>&1
echo "round6 rc=$?"; tail -3 gpurun_out/r6b/pytest_round6.log
S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.gtperlaunch.so timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu -k table_swap > gpurun_out/r6b/pytest_variant_per_launch.log 2>&1
echo "variant (expected to fail) rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r6b/pytest_variant_per_launch.log | head -8
timeout 600 python -m pytest tests/test_c_harness.py tests/test_gpu_round5.py -q -m gpu -k "harness or generator_table or config5 or ticket_times" 2>&1 | tail -3
