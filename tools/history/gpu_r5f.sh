#!/bin/bash
# (when this ran the lane-per-entry scaling pass was the default and S2K_KEY_SCALE_OLD=1 selected the lane-per-chunk one; the default is the
# lane-per-chunk pass again since - profiles/r05_key_scale_ab.txt - and S2K_KEY_SCALE_WIDE=1 selects the other)
# round 5, sixth GPU pass: share of the generator part launched beside the key chain (S2K_GP_FIRST_PERCENT), now that the scaling pass is shorter
REPO=$PWD; O=$REPO/gpurun_out/r5f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
for rep in 1 2; do for P in 60 70 80 100; do
  echo "--- first percent $P"; S2K_GP_FIRST_PERCENT=$P timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline --full 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms']['key_tables_ms'])"
done; done 2>&1 | tee $O/gp_sweep.txt
echo "--- old scale, 60"; S2K_KEY_SCALE_OLD=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline --full 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms']['key_tables_ms'])" | tee -a $O/gp_sweep.txt
