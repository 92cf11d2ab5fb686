#!/bin/bash
O=gpurun_out/r4aa; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4aa/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
s=d["schnorr_rlc_2p20"]; print("rlc", s["ms"], s.get("host_buffers")); print("msm", d["msm_2p20"]["ms"], d["msm_2p20"].get("host_buffers"))
PY
