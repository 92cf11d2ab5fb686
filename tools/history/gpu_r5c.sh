#!/bin/bash
# round 5, third GPU pass: the whole GPU suite on the runtime generator tables, context start-up times, the default bench line
REPO=$PWD; O=$REPO/gpurun_out/r5c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
for b in 0 13 5 1; do timeout 300 python3 tools/ctx_time.py --budget-gib $b 2>&1 | tail -1; done | tee $O/ctx_time.txt
timeout 300 python3 tools/ctx_time.py --gt-bits 22 2>&1 | tail -1 | tee -a $O/ctx_time.txt
timeout 3000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
