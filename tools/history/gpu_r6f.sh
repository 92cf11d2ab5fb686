#!/bin/bash
# round 6, call f: whole multiscalar call by the number of ranges of the bucket pass (fewer, longer ranges: fewer buckets cut by a border, a cheaper stitch)
mkdir -p gpurun_out/r6f
{
for LN in 262144 229376 196608 163840 131072 98304 262144; do echo "== S2K_MSM_LANES=$LN"; S2K_MSM_LANES=$LN timeout 300 python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids; done
} | tee gpurun_out/r6f/lanes_ab.txt
