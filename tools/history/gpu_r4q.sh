#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x -k "keyset" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4q/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
for k in ("keyset_resident","keyset_resident_chunk_tables"):
    v=d.get(k,{}); print(k, v.get("ms"), v.get("keyset_device_bytes"), v.get("roofline"))
PY
