#!/bin/bash
# first GPU pass of round 2: all GPU tests, the bench at the driver's settings, the two-rank flow
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
tail -c 3000 $O/bench_n1.json
python bench.py --gpus 2 --oversubscribe --steps 10 --warmup 3 > $O/bench_n2_oversub.json 2> $O/bench_n2.err; echo "bench2 rc=$?"
tail -c 1500 $O/bench_n2_oversub.json; tail -5 $O/bench_n2.err
