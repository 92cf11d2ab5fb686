#!/bin/bash
# round 5, call l: the tests the small-batch threshold touched, then the bench line
mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_c_harness.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q -m gpu -k "harness or adaptive or small or config5" > gpurun_out/r5l/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5l/tests.log
timeout 900 python bench.py > gpurun_out/r5l/bench.json 2> gpurun_out/r5l/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r5l/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r5l/bench.json')); print(d['value'], d['ms_per_step'], d.get('batch_sweep'))"
