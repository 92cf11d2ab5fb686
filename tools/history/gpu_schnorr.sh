#!/bin/bash
# BIP-340 paths: the aggregated whole-batch check and everything that shares its code
O=gpurun_out/schnorr; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "schnorr or rlc or bisect or msm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step']); print('schnorr', d.get('schnorr_rlc_2p20')); print('msm', d.get('msm_2p20')); print(d.get('extras_error'))"
