#!/bin/bash
# (when this ran the lane-per-entry scaling pass was the default and S2K_KEY_SCALE_OLD=1 selected the lane-per-chunk one; the default is the
# lane-per-chunk pass again since - profiles/r05_key_scale_ab.txt - and S2K_KEY_SCALE_WIDE=1 selects the other)
# round 5, fourth GPU pass: whole GPU suite; k_key_scale lane-per-entry vs lane-per-chunk (same box); multiscalar tail after the
# column-block and S_w changes
REPO=$PWD; O=$REPO/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for rep in 1 2; do
  echo "--- scale wide"; timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"
  echo "--- scale old"; S2K_KEY_SCALE_OLD=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['stages_ms'])"
done 2>&1 | tee $O/scale_ab.txt
timeout 300 python3 tools/msm_time.py 2>&1 | tail -2 | tee $O/msm_time.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm -o run -- python3 tools/profile_msm.py msm 8 > $O/prof_msm.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5d/prof_msm/**/run_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r5d/prof_msm/run_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:22]:
        print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
    break
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o run -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-pcie --no-cpu-baseline > $O/prof_step.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5d/prof_step/**/run_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r5d/prof_step/run_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:22]:
        print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
    break
PY
