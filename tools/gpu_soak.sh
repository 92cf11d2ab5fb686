#!/bin/bash
# long randomised runs with seeds the regression script does not use: every family of kernels against the oracle
T=${1:-soak}; S=${2:-900}; mkdir -p gpurun_out/$T
run() {  # name, seconds, command ...
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  echo "== $name"; timeout $secs "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
  echo "   rc=${PIPESTATUS[0]} $(( $(date +%s) - t0 )) s"
}
{
echo "tools/gpu_soak.sh $T (seeds from $S):"
run "tools/stress_msm.py 2000 $((S+1))" 600 python3 tools/stress_msm.py 2000 $((S+1))
S2K_MSM_SPLIT_WINDOW=2 run "S2K_MSM_SPLIT_WINDOW=2 tools/stress_msm.py 600 $((S+2))" 600 python3 tools/stress_msm.py 600 $((S+2))
run "tools/stress_rlc.py 500 $((S+3))" 600 python3 tools/stress_rlc.py 500 $((S+3))
run "tools/stress_small.py 600 $((S+4))" 900 python3 tools/stress_small.py 600 $((S+4))
run "tools/stress_keyed.py 200 $((S+5))" 900 python3 tools/stress_keyed.py 200 $((S+5))
run "tools/stress_pipeline.py 200 $((S+6))" 900 python3 tools/stress_pipeline.py 200 $((S+6))
} | tee gpurun_out/$T/soak.txt
