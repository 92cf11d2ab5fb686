#!/bin/bash
O=gpurun_out/stress; mkdir -p $O
timeout 1200 python tools/stress_keyed.py ${ITERS:-150} ${SEED:-1} > $O/stress.log 2>&1; echo "stress rc=$?"; tail -4 $O/stress.log | cut -c1-300; grep -c " ok" $O/stress.log
