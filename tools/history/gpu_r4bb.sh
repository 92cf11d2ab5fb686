#!/bin/bash
# long randomised runs of the final tree: boundary + key sets + adaptive grouping, grouped paths, big batches
O=gpurun_out/r4bb; mkdir -p $O
timeout 1500 python tools/stress_pipeline.py 200 20261003 > $O/stress_pipeline.log 2>&1; echo "stress_pipeline rc=$?"; tail -1 $O/stress_pipeline.log | cut -c1-200
ITERS=150 SEED=77 bash tools/gpu_stress.sh
timeout 900 python tools/big_device_batch_check.py > $O/big.log 2>&1; echo "big rc=$?"; grep "ok=" $O/big.log | cut -c1-200
