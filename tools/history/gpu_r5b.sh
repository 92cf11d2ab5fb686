#!/bin/bash
# round 5, second GPU pass: row arithmetic unit tests (fixed patterns), fold with quad trees, group placement / stats, 2^24 through eight members
REPO=$PWD; O=$REPO/gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_round5.py -x -q > $O/pytest_round5.log 2>&1; tail -5 $O/pytest_round5.log
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "msm or rlc or group or submit" > $O/pytest_msm_group.log 2>&1; tail -5 $O/pytest_msm_group.log
echo "--- new tail"; timeout 300 python3 tools/msm_time.py 2>&1 | tail -2 | tee $O/msm_time_new.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm -o run -- python3 tools/profile_msm.py msm 8 > $O/prof_msm.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5b/prof_msm/**/run_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r5b/prof_msm/run_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:24]:
        print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
    break
PY
timeout 600 python3 tools/group_bench.py --devices 0 --batches 8 2>&1 | tail -1 | tee $O/group_bench.json | cut -c1-1500
