#!/bin/bash
# round 5, evidence pass of the final tree: every GPU test, the counter / trace passes of the bench at the driver's settings
# (tools/collect_profiles_r04.sh), the multiscalar profiles, the side-path counters, the bench line itself (compact and --full),
# context start-up times, the group on one device
REPO=$PWD; O=$REPO/gpurun_out/r5i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 2700 bash tools/collect_profiles_r04.sh r05z > $O/collect.log 2>&1; echo "collect rc=$?"; tail -6 $O/collect.log
timeout 900 bash tools/collect_msm_profiles.sh r05 > $O/collect_msm.log 2>&1; echo "collect msm rc=$?"; tail -30 $O/collect_msm.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/side_pmc -o run -- python3 tools/side_counts.py run > $O/side_run.log 2>&1
python3 tools/side_counts.py summarize $O/side_pmc > $O/side_counts.json; cat $O/side_counts.json
cp $O/side_counts.json profiles/r05_side_counts.json
timeout 1200 python3 bench.py --write-notes > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; tail -c 600 $O/bench.json; tail -2 $O/bench.err
cp bench_notes.json $O/bench_notes.json
timeout 1200 python3 bench.py --full > $O/bench_full.json 2> /dev/null; wc -c $O/bench_full.json
for b in 0 13 5 1; do timeout 300 python3 tools/ctx_time.py --budget-gib $b 2>&1 | tail -1; done | tee $O/ctx_time.txt
timeout 300 python3 tools/ctx_time.py --gt-bits 22 2>&1 | tail -1 | tee -a $O/ctx_time.txt
timeout 600 python3 tools/group_bench.py --devices 0 --batches 8 2>&1 | tail -1 > $O/group_bench.json; cut -c1-600 $O/group_bench.json
