#!/bin/bash
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_last.json 2> gpurun_out/bench_last.err; echo rc=$?
python3 -c "
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d['value'], d['ms_per_step'], d['distinct_keys'], d.get('extras_error'), d['pcie_inclusive']['value'])"
