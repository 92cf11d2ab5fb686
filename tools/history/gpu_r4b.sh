#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_c_harness.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/boundary_probe.py 20 16 12 > $O/boundary.json 2> $O/boundary.err; echo "boundary rc=$?"; cat $O/boundary.json; tail -3 $O/boundary.err
timeout 1500 bash tools/collect_profiles_r04.sh r04a > $O/collect.log 2>&1; echo "collect rc=$?"; tail -30 $O/collect.log
