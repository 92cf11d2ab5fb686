#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 2400 bash tools/collect_profiles_r04.sh r04z > $O/collect.log 2>&1; echo "collect rc=$?"; tail -3 $O/collect.log
S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"; tail -c 300 $O/bench.err
timeout 300 python tools/boundary_probe.py 20 16 12 > $O/boundary.json 2>/dev/null; echo "probe rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4z/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"].get("frac"), "stale", d["roofline"].get("counts_stale"), d.get("extras_error"))
p=d.get("pcie_inclusive",{}); print("pipelined_keyset", p.get("pipelined_keyset",{}).get("ms_per_batch_each"))
for k in ("keyset_resident","keyset_resident_joint_tables_4bit","keyset_resident_chunk_tables","distinct_keys","worst_case_ladder_collision","resident_two_contexts"):
    v=d.get(k,{}); print(k, v.get("ms"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("kernel_ms"))
p=d.get("pcie_inclusive",{}); print("pipelined", p.get("pipelined",{}).get("ms_per_batch_each"), p.get("pipelined",{}).get("fraction_of_resident_value"))
PY
