#!/bin/bash
# wide joint tables: rates of the 4-, 5- and 6-bit layouts (6 bits at 2^13 keys: 3.6 MiB per key), then the full bench line
O=gpurun_out/r4y; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
for opt in "keyset" "keyset5" "keyset5 --keys-log2 13" "keyset6 --keys-log2 13" "keyset --keys-log2 13"; do
  timeout 300 $B --key-grouping $opt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$opt]', 'ms_per_step=%.3f kernel_ms=%.3f clock=%.0f frac=%.3f at_clock=%.3f instr=%s from %s' % (d['ms_per_step'], r['kernel_ms'], r['shader_clock_mhz'], r.get('frac',0), r.get('frac_at_measured_clock',0), r.get('valu_instr_per_verify'), r.get('counts_from')))"
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4y/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d.get("extras_error"))
for k in ("keyset_resident","keyset_resident_joint_tables_4bit","keyset_resident_chunk_tables"):
    v=d.get(k,{}); r=v.get("roofline",{}); print(k, v.get("ms"), v.get("keyset_build_s"), v.get("keyset_device_bytes"), r.get("kernel_ms"), r.get("frac"), r.get("frac_at_measured_clock"), r.get("shader_clock_mhz"))
print("pipelined_keyset", d["pcie_inclusive"].get("pipelined_keyset",{}).get("ms_per_batch_each"))
PY
