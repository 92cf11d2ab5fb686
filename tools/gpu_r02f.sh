#!/bin/bash
O=gpurun_out/r02f; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
bash tools/ab_libs.sh default pack1 default > $O/ab.log 2>&1; grep "^\[" $O/ab.log
bash tools/collect_profiles_r02.sh r02f > $O/collect.log 2>&1; tail -30 $O/collect.log
