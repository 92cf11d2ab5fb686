#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_keyed.py tests/test_gpu_round4.py -m gpu -q -x -k "keyset or exceptional or worst" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4n/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
k=d.get("keyset_resident",{})
print("keyset", k.get("ms"), k.get("roofline"))
print(d.get("extras_error"))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras --key-grouping keyset | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keyset main', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('stages_ms'))"
