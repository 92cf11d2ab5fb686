#!/bin/bash
# round 5, call v: the final tree - every GPU test, smoke(), randomised runs through all three families of kernels, the bench line
mkdir -p gpurun_out/r5v
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r5v/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r5v/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
{
echo "tools/gpu_r5v.sh, the final tree (wave / quad / lane per signature):"
echo "== tools/stress_small.py 400 121"; timeout 2400 python3 tools/stress_small.py 400 121 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_keyed.py 80 122"; timeout 2400 python3 tools/stress_keyed.py 80 122 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_pipeline.py 80 123"; timeout 2400 python3 tools/stress_pipeline.py 80 123 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
} | tee gpurun_out/r5v/stress.txt
timeout 900 python bench.py > gpurun_out/r5v/bench.json 2> gpurun_out/r5v/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r5v/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r5v/bench.json')); print(d['value'], d['ms_per_step'], d.get('batch_sweep'))"
