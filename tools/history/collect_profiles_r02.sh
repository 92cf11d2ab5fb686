#!/bin/bash
# Run on the GPU box:  gpurun -- bash tools/collect_profiles_r02.sh TAG
# rocprofv3 kernel trace + stats of the bench at the DRIVER'S OWN settings (--steps 20 --warmup 5), then
# the counter passes, each in its own run (only --kernel-trace next to --pmc).  Raw output lands in
# gpurun_out/<kind>_TAG; the summaries to commit are written to gpurun_out/profiles_TAG/.
TAG=${1:-r02x}
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out
P=$O/profiles_$TAG
mkdir -p $P
cd $REPO
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- $B > $P/bench_under_kernel_trace.json 2> $O/prof_${TAG}.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o run -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o run -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM --output-format csv -d $O/pmc_$TAG -o run -- $B > $P/bench_under_pmc.json 2> $O/pmc_${TAG}.err
cp $O/prof_$TAG/run_kernel_stats.csv $P/kernel_stats_bench_steps20_warmup5.csv
python3 tools/collect_traffic.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG > $P/hbm_traffic.json
python3 tools/summarize_pmc.py $O/pmc_$TAG > $P/pmc_per_dispatch.txt
python3 tools/summarize_profiles_r02.py $O/prof_$TAG/run_kernel_trace.csv $O/pmc_$TAG $P
# the unprofiled bench right after, same box: the line the profile has to reconcile with
$B > $P/bench_same_box_unprofiled.json 2>/dev/null
tail -c 600 $P/bench_same_box_unprofiled.json; echo
cat $P/kernel_time_summary.json
cat $P/valu_counts.json
