#!/bin/bash
# the rewritten parse kernel: encoded tests, kernel time, pipelined encoded rate
O=gpurun_out/r4w; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "encoded or wycheproof or bip0066 or recoverable or group_two or harness" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
bash tools/gpu_parse_prof.sh 2>&1 | grep "parse_encoded\|encoded (DER"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4w/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
e=d.get("encoded_2p20",{}); print("encoded", e.get("ms_each"), e.get("pipelined",{}).get("ms_per_batch_each"))
print("pinned pipelined", d["pcie_inclusive"]["pipelined"]["ms_per_batch_each"])
PY
