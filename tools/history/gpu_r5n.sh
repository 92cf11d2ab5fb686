#!/bin/bash
# round 5, call n: all GPU tests of the final tree, then the bench line as the driver runs it
mkdir -p gpurun_out/r5n
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r5n/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r5n/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r5n/bench.json 2> gpurun_out/r5n/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r5n/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r5n/bench.json')); print(d['value'], d['ms_per_step'], d.get('batch_sweep'), d.get('dropped'))"
