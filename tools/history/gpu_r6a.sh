#!/bin/bash
# round 6, call a: one generator-table view per call.  The new tests on the shipped library; the table-swap test on the variant
# that reloads the view per launch (the fault of round 5: must FAIL); then every GPU test, smoke(), and a bench line.
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_round6.py -q -m gpu > gpurun_out/r6a/pytest_round6.log 2>&1
echo "round6 rc=$?"; tail -5 gpurun_out/r6a/pytest_round6.log
S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.gtperlaunch.so timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu -k table_swap > gpurun_out/r6a/pytest_variant_per_launch.log 2>&1
echo "variant (expected to fail) rc=$?"; grep -E "AssertionError|passed|failed" gpurun_out/r6a/pytest_variant_per_launch.log | head -5
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r6a/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r6a/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6a/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r6a/bench.json')); print(d['value'], d['ms_per_step'], d.get('msm_2p20'), d.get('schnorr_rlc_2p20'))"
