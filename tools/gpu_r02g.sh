#!/bin/bash
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "exceptional or worst_case or small_r or random_batches or wycheproof_ecdsa or force_complete or structured_fuzz or ecdsa_kats or full_size_properties" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02g/bench_n1.json"))
print({k:d[k] for k in ("value","ms_per_step")}); r=d["roofline"]; print({k:r.get(k) for k in ("kernel_ms","frac","frac_at_measured_clock","mad_issue_frac_at_measured_clock","shader_clock_mhz","traffic","frac_of_2cycle_nominal_peak")})
for k in ("distinct_keys","worst_case_all_fallback","msm_2p20","schnorr_rlc_2p20","pcie_inclusive","extras_error"):
    print(k, d.get(k))
PY
