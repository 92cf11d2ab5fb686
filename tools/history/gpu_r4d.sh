#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_keyed.py tests/test_gpu_round4.py tests/test_c_harness.py -m gpu -q -x -k "keyset or keyed or exceptional or submit or group or harness or worst" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q$q.json 2> $O/bench_q$q.err; echo "bench q=$q rc=$?"
done
python3 - <<'PY'
import json
for q in (8,16):
    try:
        d=json.loads(open("gpurun_out/r4d/bench_q%d.json"%q).read().strip().splitlines()[-1])
    except Exception as e:
        print(q, "no line", e); continue
    p=d.get("pcie_inclusive",{})
    print("q",q,"value",round(d["value"]/1e6,1),"ms",round(d["ms_per_step"],3),
          "pinned",[round(x,2) for x in p.get("pinned",{}).get("ms_each",[])],
          "pipelined",[round(x,3) for x in p.get("pipelined",{}).get("ms_per_batch_each",[])], "frac", p.get("pipelined",{}).get("fraction_of_resident_value"),
          "pageable_pl",[round(x,3) for x in p.get("pipelined_pageable",{}).get("ms_per_batch_each",[])],
          "enc", [round(x,2) for x in d.get("encoded_2p20",{}).get("ms_each",[])], "enc_pl", [round(x,3) for x in d.get("encoded_2p20",{}).get("pipelined",{}).get("ms_per_batch_each",[])])
    k=d.get("keyset_resident",{})
    print("   keyset", k.get("ms"), (k.get("roofline") or {}).get("kernel_ms"), (k.get("roofline") or {}).get("frac"), (k.get("roofline") or {}).get("frac_at_measured_clock"), "dk", d.get("distinct_keys",{}).get("grouping_overhead"), d.get("extras_error"))
PY
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/boundary_probe.py 20 16 12 x > $O/boundary_q8.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/boundary_q8.json')); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k and 'all' not in k and 'stats' not in k})"
