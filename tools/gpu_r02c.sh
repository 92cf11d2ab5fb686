#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
./tools/valu_rates2 > $O/valu_instruction_rates.txt 2>&1; tail -5 $O/valu_instruction_rates.txt
bash tools/collect_profiles_r02.sh r02c > $O/collect.log 2>&1; tail -40 $O/collect.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "generator_table" 2>&1 | tail -3
