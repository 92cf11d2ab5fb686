#!/bin/bash
# multi-rank flows of bench.py on a 1-GPU box (ranks share the device, gloo collectives)
O=gpurun_out/multirank; mkdir -p $O
timeout 600 python bench.py --gpus 8 --oversubscribe --batch-log2 15 --steps 3 --warmup 1 > $O/self_launch_8.json 2> $O/self_launch_8.err; echo "self-launch 8 ranks rc=$?"
tail -c 700 $O/self_launch_8.json; echo
S2K_BENCH_DEVICE=0 S2K_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --batch-log2 16 --steps 3 --warmup 1 > $O/torchrun_2.json 2> $O/torchrun_2.err; echo "torchrun 2 ranks rc=$?"
grep "^{" $O/torchrun_2.json | tail -c 600; echo
# config 5's shape per rank (2^21 signatures, 32 per key), two ranks on the one device
timeout 900 python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 > $O/self_launch_2_full.json 2> $O/self_launch_2_full.err; echo "self-launch 2 ranks, 2^21 each rc=$?"
grep "^{" $O/self_launch_2_full.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['config']['workload'][:90], d.get('key_grouping',{}).get('signatures_on_key_tables'))"
