#!/bin/bash
# the gloo all-gather of two ranks on one box: which interface, what payload
O=gpurun_out/two_rank; mkdir -p $O
for v in "X=1" "GLOO_SOCKET_IFNAME=lo"; do
  for b in 16 17 18 21; do
    env $v timeout 600 python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 --batch-log2 $b > $O/p2.json 2> $O/p2.err
    echo "$v 2^$b rc=$? $(grep '^{' $O/p2.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_rank']
print(d['ms_per_step'], 'local', [round(x,1) for x in m['per_rank_local_ms']], 'collective', [round(x,1) for x in m['collective_ms']])")"
  done
done
ip -br addr 2>/dev/null | head; hostname; getent hosts $(hostname) || echo "hostname does not resolve"
