#!/bin/bash
# round 5, call u: the four-lanes-per-signature ladder (k_verify_quad) - the test that reads the grouped path's statistics through
# RCCL, smoke(), the mid-size probe, the randomised runs, the bench line
mkdir -p gpurun_out/r5u
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round5.py -q -m gpu -k "rccl or smoke or small_batch or two_ranks" 2>&1 | tail -3
timeout 600 python tools/mid_batch_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5u/mid_batch_ab.txt; cat gpurun_out/r5u/mid_batch_ab.txt
timeout 900 python3 tools/stress_keyed.py 60 111 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-100
timeout 900 python3 tools/stress_pipeline.py 60 112 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
timeout 900 python bench.py > gpurun_out/r5u/bench.json 2> gpurun_out/r5u/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r5u/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r5u/bench.json')); print(d['value'], d['ms_per_step'], d.get('batch_sweep'))"
