#!/bin/bash
# (when this ran the lane-per-entry scaling pass was the default and S2K_KEY_SCALE_OLD=1 selected the lane-per-chunk one; the default is the
# lane-per-chunk pass again since - profiles/r05_key_scale_ab.txt - and S2K_KEY_SCALE_WIDE=1 selects the other)
# round 5, seventh GPU pass: does the ladder's table traffic cost clock?  Same 2^20 signatures under 2^16, 2^12, 2^8, 2^4 keys (old scaling pass):
# ladder time, its shader clock, step time
REPO=$PWD; O=$REPO/gpurun_out/r5g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
for K in 16 12 8 4 16; do
  echo "--- keys 2^$K"; S2K_KEY_SCALE_OLD=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --keys-log2 $K --no-extras --no-pcie --no-cpu-baseline --full 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['ms_per_step'], r['kernel_ms'], r['shader_clock_mhz'], r['stages_ms'])"
done 2>&1 | tee $O/keys_clock.txt
