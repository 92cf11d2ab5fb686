#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/gpu_check.sh): every GPU test, then the bench at the driver's settings with a short digest.
O=gpurun_out/check; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -3 $O/bench_n1.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/check/bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step")}); r=d["roofline"]; print({k:r.get(k) for k in ("kernel_ms","frac","frac_at_measured_clock","shader_clock_mhz_first_wave","shader_clock_mhz_last_round")})
print("stages", r.get("stages_ms")); print("grouping", d.get("key_grouping"))
for k in ("general_path_same_batch","distinct_keys","worst_case_all_fallback","msm_2p20","schnorr_rlc_2p20","pcie_inclusive","extras_error"):
    print(k, d.get(k))
PY
