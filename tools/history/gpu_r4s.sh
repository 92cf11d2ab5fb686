#!/bin/bash
# adaptive key grouping: its test, the grouping tests around it, and the bench line's distinct_keys
O=gpurun_out/r4s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_keyed.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
v=d.get("distinct_keys",{}); print({k:v.get(k) for k in ("ms","ms_with_key_grouping_off","grouping_overhead","ms_with_key_grouping_in_every_call","grouping_overhead_in_every_call","adaptive_calls","ms_rounds","ms_rounds_every_call","ms_rounds_off")})
PY
