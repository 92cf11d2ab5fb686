#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 1500 bash tools/collect_profiles_r04.sh r04b > $O/collect.log 2>&1; echo "collect rc=$?"; tail -4 $O/collect.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"].get("frac"), "stale", d["roofline"].get("counts_stale"))
for k in ("general_path_same_batch","distinct_keys","worst_case_all_fallback","worst_case_equal_points","forced_worklist","keyset_resident"):
    v=d.get(k,{})
    print(k, {a:v.get(a) for a in ("ms","ms_with_key_grouping_off","grouping_overhead","on_worklist")}, (v.get("roofline") or {}).get("frac"))
print("pcie", {k:(v if not isinstance(v,dict) else {a:v.get(a) for a in ("ms_per_batch","fraction_of_resident_value","ms_each")}) for k,v in d.get("pcie_inclusive",{}).items() if k in ("ms_each","pinned","pipelined","pipelined_pageable")})
print("encoded", {k:d.get("encoded_2p20",{}).get(k) for k in ("ms_each",)}, d.get("encoded_2p20",{}).get("pipelined",{}).get("ms_per_batch"))
print("msm", d.get("msm_2p20",{}).get("ms"), "rlc", d.get("schnorr_rlc_2p20",{}).get("ms"))
print("cpu", {k:d.get("cpu_baseline",{}).get(k) for k in ("value","threads","speedup_vs_1_thread","cgroup_cpu_quota","host_logical_cpus","cpus_in_affinity_mask")})
print(d.get("extras_error"))
PY
