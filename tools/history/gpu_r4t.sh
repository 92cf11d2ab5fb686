#!/bin/bash
# key sets through submit / wait: test, then the bench line's pcie_inclusive.pipelined_keyset
O=gpurun_out/r4t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "keyset_submit or adaptive" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4t/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
p=d["pcie_inclusive"]
for k in ("pipelined","pipelined_keyset"):
    print(k, p.get(k,{}).get("ms_per_batch_each"), p.get(k,{}).get("value"))
print("keyset_resident", d.get("keyset_resident",{}).get("ms"))
PY
