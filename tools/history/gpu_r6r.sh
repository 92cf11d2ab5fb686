#!/bin/bash
# round 6, call r: context lifecycle with the builder joined (s2k_ctx_destroy at every stage of the background build; exit with a live context)
mkdir -p gpurun_out/r6r
T0=$(date +%s.%N)
python3 tools/ctx_lifecycle_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6r/ctx_lifecycle.txt
echo "rc of the process ${PIPESTATUS[0]}, wall time $(echo "$(date +%s.%N) - $T0" | bc) s (import torch-free: about 1 s of start-up)" | tee -a gpurun_out/r6r/ctx_lifecycle.txt
