#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
timeout 300 python tools/boundary_probe.py 20 16 12 > $O/boundary.json 2>/dev/null; echo "probe rc=$?"
timeout 300 python tools/dma_interference_probe.py > $O/dma.txt 2>/dev/null; cat $O/dma.txt
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4m/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"].get("frac"), "stale", d["roofline"].get("counts_stale"))
for k in ("general_path_same_batch","distinct_keys","worst_case_all_fallback","worst_case_equal_points","forced_worklist","keyset_resident"):
    v=d.get(k,{})
    print(k, {a:v.get(a) for a in ("ms","ms_with_key_grouping_off","grouping_overhead","on_worklist")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("kernel_ms"))
p=d.get("pcie_inclusive",{})
print("pcie", p.get("ms_each"), p.get("pinned",{}).get("ms_each"), p.get("pipelined",{}).get("ms_per_batch_each"), p.get("pipelined",{}).get("fraction_of_resident_value"), p.get("pipelined_pageable",{}).get("ms_per_batch"))
print("encoded", d.get("encoded_2p20",{}).get("ms_each"), d.get("encoded_2p20",{}).get("pipelined",{}).get("ms_per_batch_each"))
print("msm", d.get("msm_2p20",{}).get("ms"), "rlc", d.get("schnorr_rlc_2p20",{}).get("ms"))
print(d.get("extras_error"))
b=json.load(open("gpurun_out/r4m/boundary.json"))
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in b.items() if 'ms' in k and 'all' not in k and 'stats' not in k})
PY
