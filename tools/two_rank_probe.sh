#!/bin/bash
# two ranks of bench.py on ONE device (test hook --oversubscribe): what the step costs by batch size, hardware queues and flow
O=gpurun_out/two_rank; mkdir -p $O
one() {  # tag, env..., -- args
  local tag=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 "$@" > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$? $(grep '^{' $O/$tag.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_rank']
print(d['ms_per_step'], 'local', [round(x,1) for x in m['per_rank_local_ms']], 'collective', [round(x,1) for x in m['collective_ms']])")"
}
one q8_2p21 X=1 --
one q4_2p21 GPU_MAX_HW_QUEUES=4 --
one q2_2p21 GPU_MAX_HW_QUEUES=2 --
one q8_2p20 X=1 -- --batch-log2 20
one q8_2p18 X=1 -- --batch-log2 18
one q8_2p21_off X=1 -- --key-grouping off
