#!/bin/bash
# round 5, eighth GPU pass: lanes of the bucket pass (ranges twice as long halve the buckets that cross a range border: less stitching)
REPO=$PWD; O=$REPO/gpurun_out/r5h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
for rep in 1 2; do for L in 262144 196608 131072 98304; do
  echo "--- S2K_MSM_LANES=$L"; S2K_MSM_LANES=$L timeout 300 python3 tools/msm_time.py 2>&1 | tail -2
done; done | tee $O/msm_lanes.txt
S2K_MSM_LANES=131072 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_msm -o run -- python3 tools/profile_msm.py msm 8 > $O/prof_msm.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r5h/prof_msm/**/run_kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/r5h/prof_msm/run_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), ("%.1f us" % (float(r['AverageNs'])/1e3)).rjust(12))
    break
PY
