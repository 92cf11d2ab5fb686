#!/bin/bash
# round 5, confirmation pass of the final tree: every GPU test, smoke(), randomised long runs (multiscalar with the new and the old
# reduction and the other geometries, grouped verification paths, the streaming boundary), the group on one device with the wide tables in
REPO=$PWD; O=$REPO/gpurun_out/r5j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
{
echo "tools/gpu_r5j.sh on the final tree, one MI355X:"
echo "== tools/stress_msm.py 120 51 (new reduction)"; timeout 1200 python3 tools/stress_msm.py 120 51 2>&1 | grep -v amdgpu.ids | tail -2
echo "== S2K_MSM_OLD_REDUCE=1 tools/stress_msm.py 40 52"; S2K_MSM_OLD_REDUCE=1 timeout 900 python3 tools/stress_msm.py 40 52 2>&1 | grep -v amdgpu.ids | tail -2
echo "== S2K_MSM_WIDE_PAIRS=1 tools/stress_msm.py 40 53"; S2K_MSM_WIDE_PAIRS=1 timeout 900 python3 tools/stress_msm.py 40 53 2>&1 | grep -v amdgpu.ids | tail -2
echo "== S2K_MSM_LANES=4096 tools/stress_msm.py 40 54"; S2K_MSM_LANES=4096 timeout 900 python3 tools/stress_msm.py 40 54 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tools/stress_rlc.py"; timeout 900 python3 tools/stress_rlc.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tools/stress_keyed.py 100 55"; timeout 1200 python3 tools/stress_keyed.py 100 55 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== S2K_KEY_SCALE_WIDE=1 tools/stress_keyed.py 40 56"; S2K_KEY_SCALE_WIDE=1 timeout 1200 python3 tools/stress_keyed.py 40 56 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_pipeline.py 80 57"; timeout 1500 python3 tools/stress_pipeline.py 80 57 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
} | tee $O/stress.txt
timeout 600 python3 tools/group_bench.py --devices 0 --batches 12 2>&1 | tail -1 > $O/group_bench.json; cut -c1-900 $O/group_bench.json
timeout 600 python3 tools/group_bench.py --devices 0 --batches 12 --keyset 2>&1 | tail -1 > $O/group_bench_keyset.json; cut -c1-400 $O/group_bench_keyset.json
timeout 300 python3 tools/msm_time.py 2>&1 | tail -2 | tee $O/msm_time.txt
