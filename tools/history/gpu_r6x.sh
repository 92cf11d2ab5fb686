#!/bin/bash
# round 6, call x: the multiscalar profiles and the bench line of the final tree (after the last changes to the bucket pass and k_msm_stitch)
bash tools/collect_msm_profiles.sh r06y 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -40
mkdir -p gpurun_out/r6x
timeout 900 python bench.py > gpurun_out/r6x/bench.json 2> gpurun_out/r6x/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6x/bench.json)"
