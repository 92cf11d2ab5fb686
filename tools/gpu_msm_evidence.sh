#!/bin/bash
# the multiscalar profiles (kernel trace, traffic, counters per kernel: collect_msm_profiles.sh) and the bench line of the tree
bash tools/collect_msm_profiles.sh ${1:-r06v} 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -40
mkdir -p gpurun_out/${1:-r06v}_bench
timeout 900 python bench.py > gpurun_out/${1:-r06v}_bench/bench.json 2> gpurun_out/${1:-r06v}_bench/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/${1:-r06v}_bench/bench.json)"
