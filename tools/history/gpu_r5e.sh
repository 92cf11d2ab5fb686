#!/bin/bash
# (when this ran the lane-per-entry scaling pass was the default and S2K_KEY_SCALE_OLD=1 selected the lane-per-chunk one; the default is the
# lane-per-chunk pass again since - profiles/r05_key_scale_ab.txt - and S2K_KEY_SCALE_WIDE=1 selects the other)
# round 5, fifth GPU pass: first-verdict latency with the gated table build; k_key_scale through LDS vs the lane-per-chunk walk (same box);
# the whole bench line in its compact form (new rows: recover_2p20, schnorr_per_signature_2p20, batch_sweep)
REPO=$PWD; O=$REPO/gpurun_out/r5e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 1200 python3 -m pytest tests/test_gpu_round5.py -q -k "budget or widths or ticket" > $O/pytest_r5.log 2>&1; tail -4 $O/pytest_r5.log
timeout 1200 python3 -m pytest tests/test_gpu_keyed.py tests/test_gpu_round4.py -q -x -k "keyset or keyed or grouping or ragged" > $O/pytest_keyed.log 2>&1; tail -3 $O/pytest_keyed.log
for rep in 1 2; do
  echo "--- scale wide (LDS)"; timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline --full 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"
  echo "--- scale old"; S2K_KEY_SCALE_OLD=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline --full 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"
done 2>&1 | tee $O/scale_ab.txt
timeout 900 python3 bench.py --write-notes > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; tail -c 2500 $O/bench.json; tail -3 $O/bench.err
cp bench_notes.json $O/bench_notes.json 2>/dev/null
