#!/bin/bash
# round 5, evidence pass of the final tree (after the wave-per-signature ladders): every GPU test, smoke(), the counter / trace
# passes of the bench at the driver's settings (tools/collect_profiles_r04.sh), the multiscalar profiles, the side-path counters
# (now with the row kernels), the bench line (compact and --full), context start-up times, the group on one device, randomised long runs
REPO=$PWD; O=$REPO/gpurun_out/r5o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 3000 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 2700 bash tools/collect_profiles_r04.sh r05z > $O/collect.log 2>&1; echo "collect rc=$?"; tail -6 $O/collect.log
timeout 900 bash tools/collect_msm_profiles.sh r05 > $O/collect_msm.log 2>&1; echo "collect msm rc=$?"; tail -30 $O/collect_msm.log
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/side_pmc -o run -- python3 tools/side_counts.py run > $O/side_run.log 2>&1
python3 tools/side_counts.py summarize $O/side_pmc > $O/side_counts.json; cat $O/side_counts.json
rm -rf $O/side_pmc
timeout 1200 python3 bench.py --write-notes > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; tail -c 600 $O/bench.json; tail -2 $O/bench.err
cp bench_notes.json $O/bench_notes.json
timeout 1200 python3 bench.py --full > $O/bench_full.json 2> /dev/null; wc -c $O/bench_full.json
for b in 0 13 5 1; do timeout 300 python3 tools/ctx_time.py --budget-gib $b 2>&1 | tail -1; done | tee $O/ctx_time.txt
timeout 300 python3 tools/ctx_time.py --gt-bits 22 2>&1 | tail -1 | tee -a $O/ctx_time.txt
timeout 600 python3 tools/group_bench.py --devices 0 --batches 8 2>&1 | tail -1 > $O/group_bench.json; cut -c1-600 $O/group_bench.json
{
echo "tools/gpu_r5o.sh on the final tree, one MI355X:"
echo "== tools/stress_small.py 150 62"; timeout 1200 python3 tools/stress_small.py 150 62 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_msm.py 120 51 (new reduction)"; timeout 1200 python3 tools/stress_msm.py 120 51 2>&1 | grep -v amdgpu.ids | tail -2
echo "== S2K_MSM_OLD_REDUCE=1 tools/stress_msm.py 40 52"; S2K_MSM_OLD_REDUCE=1 timeout 900 python3 tools/stress_msm.py 40 52 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tools/stress_rlc.py"; timeout 900 python3 tools/stress_rlc.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== tools/stress_keyed.py 100 55"; timeout 1200 python3 tools/stress_keyed.py 100 55 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== tools/stress_pipeline.py 80 57"; timeout 1500 python3 tools/stress_pipeline.py 80 57 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
} | tee $O/stress.txt
