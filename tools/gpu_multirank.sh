#!/bin/bash
# multi-rank flows of bench.py on a 1-GPU box (ranks share the device, gloo collectives)
O=gpurun_out/multirank; mkdir -p $O
timeout 600 python bench.py --gpus 8 --oversubscribe --batch-log2 15 --steps 3 --warmup 1 > $O/self_launch_8.json 2> $O/self_launch_8.err; echo "self-launch 8 ranks rc=$?"
tail -c 700 $O/self_launch_8.json; echo
S2K_BENCH_DEVICE=0 S2K_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --batch-log2 16 --steps 3 --warmup 1 > $O/torchrun_2.json 2> $O/torchrun_2.err; echo "torchrun 2 ranks rc=$?"
grep "^{" $O/torchrun_2.json | tail -c 600; echo
timeout 300 python -m pytest tests/test_c_harness.py -m gpu -q 2>&1 | tail -2
