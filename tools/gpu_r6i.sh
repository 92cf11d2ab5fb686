#!/bin/bash
# round 6, call i: every GPU test, smoke(), the randomised runs and the bench line of the tree
T=${1:-r6i}; mkdir -p gpurun_out/$T
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/$T/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/$T/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
{
echo "tools/gpu_r6i.sh:"
echo "== tools/stress_small.py 200 131"; timeout 2400 python3 tools/stress_small.py 200 131 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_keyed.py 60 132"; timeout 2400 python3 tools/stress_keyed.py 60 132 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_pipeline.py 60 133"; timeout 2400 python3 tools/stress_pipeline.py 60 133 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_msm.py 100 134"; timeout 2400 python3 tools/stress_msm.py 100 134 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== S2K_MSM_SPLIT_WINDOW=2 tools/stress_msm.py 60 135"; S2K_MSM_SPLIT_WINDOW=2 timeout 2400 python3 tools/stress_msm.py 60 135 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== tools/stress_rlc.py 40 136"; timeout 2400 python3 tools/stress_rlc.py 40 136 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
} | tee gpurun_out/$T/stress.txt
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/$T/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/$T/bench.json')); print(d['value'], d['ms_per_step'], d.get('small_call'), d.get('dropped'))"
