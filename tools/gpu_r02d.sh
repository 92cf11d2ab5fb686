#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
python -m pytest tests/test_gpu_round2.py -m gpu -q -s -k "bisection or rlc" > $O/bisect.log 2>&1; echo "rc=$?"; tail -25 $O/bisect.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "schnorr" 2>&1 | tail -3
./tools/valu_rates2 > $O/valu_instruction_rates.txt 2>&1; tail -9 $O/valu_instruction_rates.txt | cut -c1-200
