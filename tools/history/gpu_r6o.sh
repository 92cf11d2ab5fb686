#!/bin/bash
# round 6, call o: the bench line of the final tree at its defaults
mkdir -p gpurun_out/r6o
timeout 900 python bench.py > gpurun_out/r6o/bench.json 2> gpurun_out/r6o/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6o/bench.json)"; python -c "
import json; d=json.load(open('gpurun_out/r6o/bench.json')); print(d['value'], d['ms_per_step'], d.get('msm_2p20'), d.get('dropped'), d.get('extras_error'))"
