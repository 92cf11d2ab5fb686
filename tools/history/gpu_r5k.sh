#!/bin/bash
# round 5, call k: the wave-per-signature ladder for small batches - parity through both ladders, the crossover by size, then all GPU tests
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_keyed.py -x -q -m gpu -k "small or fer or pt29r or swaps or ragged or invalid_keys or chosen or malleable" > gpurun_out/r5k/small_tests.log 2>&1
echo "small tests rc=$?"; tail -5 gpurun_out/r5k/small_tests.log
timeout 600 python tools/small_batch_probe.py > gpurun_out/r5k/small_batch_ab.txt 2>&1
echo "probe rc=$?"; tail -16 gpurun_out/r5k/small_batch_ab.txt
if [ "$1" = full ]; then
  timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r5k/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -8 gpurun_out/r5k/pytest_gpu.log
fi
