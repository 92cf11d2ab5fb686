#!/bin/bash
# default library, then prebuilt variants: MSM parity tests once, timing + kernel stats for each
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -m gpu -k "msm or rlc or bisect" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
for V in default "$@"; do
  echo "== $V"
  if [ "$V" != default ]; then export S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.$V.so; fi
  python3 tools/msm_time.py 2>&1 | grep -v amdgpu.ids
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$V -o run -- python3 tools/profile_msm.py msm 8 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ab_$V/run_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("k_msm") and float(r["AverageNs"]) > 8000: print("   %-30s %8.1f us" % (n, float(r["AverageNs"]) / 1e3))
PY
done
