#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
S=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4o/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "steps", d["steps"], d.get("extras_error"))
print("msm", d["msm_2p20"]["ms"], d["msm_2p20"].get("two_calls_in_flight_ms"), d.get("msm_2p22"))
c=d["cpu_baseline"]; print(c["thread_probe"], c["cores"], c["value"], c["speedup_vs_1_thread"], c["cgroup_cpu_quota"])
PY
