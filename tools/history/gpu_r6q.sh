#!/bin/bash
# round 6, call q: context lifecycle with the builder joined; the two-part multiscalar flow on sizes that reach it; the bench line in its final form
mkdir -p gpurun_out/r6q
{ /usr/bin/time -f "process wall time %e s" python3 tools/ctx_lifecycle_probe.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6q/ctx_lifecycle.txt
S2K_MSM_SPLIT_WINDOW=2 timeout 1200 python3 tools/stress_msm.py 120 611 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r6q/bench.json 2> gpurun_out/r6q/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6q/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r6q/bench.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'], d.get('dropped'))"
