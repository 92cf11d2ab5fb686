#!/bin/bash
# round 5, first GPU pass: row arithmetic unit tests, the multiscalar tests on the new reduction, timings old vs new tail
REPO=$PWD; O=$REPO/gpurun_out/r5a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_round5.py -x -q > $O/pytest_round5.log 2>&1; tail -5 $O/pytest_round5.log
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "msm or rlc or schnorr or pt29" > $O/pytest_msm.log 2>&1; tail -5 $O/pytest_msm.log
echo "--- new tail"; timeout 300 python3 tools/msm_time.py 2>&1 | tail -2 | tee $O/msm_time_new.txt
echo "--- old tail"; S2K_MSM_OLD_REDUCE=1 timeout 300 python3 tools/msm_time.py 2>&1 | tail -2 | tee $O/msm_time_old.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm -o run -- python3 tools/profile_msm.py msm 8 > $O/prof_msm.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5a/prof_msm/**/run_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r5a/prof_msm/run_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:30]:
        print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
    break
PY
