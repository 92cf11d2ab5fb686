#!/bin/bash
# round 6, call z: refresh bench_notes.json (the explanations of the bench line, by path) from a run of the final tree
mkdir -p gpurun_out/r6z
timeout 900 python bench.py --write-notes > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r6z/bench.json)"; cp bench_notes.json gpurun_out/r6z/bench_notes.json; wc -c bench_notes.json
