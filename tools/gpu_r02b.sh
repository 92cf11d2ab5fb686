#!/bin/bash
O=gpurun_out/r02b; mkdir -p $O
python -m pytest tests/test_gpu_hotpath.py -m gpu -q > $O/hotpath.log 2>&1; echo "hotpath rc=$?"; tail -40 $O/hotpath.log
echo skip-rest
