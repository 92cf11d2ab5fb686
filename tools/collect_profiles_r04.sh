#!/bin/bash
# Run on the GPU box:  gpurun -- bash tools/collect_profiles_r04.sh TAG
# As tools/collect_profiles_r02.sh (kernel trace + stats of the bench at the DRIVER'S settings, then the counter passes,
# each in its own run), plus the same counter passes with the key grouping OFF (bench.py --key-grouping off), so that the
# general ladder k_verify_fast<ECDSA> - what a batch without key reuse runs on - has PMC counts of the same tree.
# Summaries to commit land in gpurun_out/profiles_TAG/ (tools/merge_counts_r04.py writes the profiles/ files from them).
TAG=${1:-r04x}
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out
P=$O/profiles_$TAG
mkdir -p $P
cd $REPO
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-extras"
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- $B > $P/bench_under_kernel_trace.json 2> $O/prof_${TAG}.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o run -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o run -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_$TAG -o run -- $B > $P/bench_under_pmc.json 2> $O/pmc_${TAG}.err
cp $O/prof_$TAG/run_kernel_stats.csv $P/kernel_stats_bench_steps20_warmup5.csv
python3 tools/collect_traffic.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG > $P/hbm_traffic.json
python3 tools/summarize_pmc.py $O/pmc_$TAG > $P/pmc_per_dispatch.txt
python3 tools/summarize_profiles_r02.py $O/prof_$TAG/run_kernel_trace.csv $O/pmc_$TAG $P
# the general ladder: same passes, key grouping off
G="$B --key-grouping off"
mkdir -p $P/general
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g_$TAG -o run -- $G > $P/general/bench_under_kernel_trace.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_g_$TAG -o run -- $G > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_g_$TAG -o run -- $G > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_g_$TAG -o run -- $G > $P/general/bench_under_pmc.json 2> /dev/null
cp $O/prof_g_$TAG/run_kernel_stats.csv $P/general/kernel_stats_bench_steps20_warmup5.csv
python3 tools/collect_traffic.py $O/pmc_fetch_g_$TAG $O/pmc_write_g_$TAG > $P/general/hbm_traffic.json
python3 tools/summarize_profiles_r02.py $O/prof_g_$TAG/run_kernel_trace.csv $O/pmc_g_$TAG $P/general
# the key-set ladder (32-chunk tables, no doublings): kernel trace and the instruction counters
K="$B --key-grouping keyset"
mkdir -p $P/keyset
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k_$TAG -o run -- $K > $P/keyset/bench_under_kernel_trace.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_k_$TAG -o run -- $K > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_k_$TAG -o run -- $K > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_k_$TAG -o run -- $K > $P/keyset/bench_under_pmc.json 2> /dev/null
cp $O/prof_k_$TAG/run_kernel_stats.csv $P/keyset/kernel_stats_bench_steps20_warmup5.csv
python3 tools/collect_traffic.py $O/pmc_fetch_k_$TAG $O/pmc_write_k_$TAG > $P/keyset/hbm_traffic.json
python3 tools/summarize_profiles_r02.py $O/prof_k_$TAG/run_kernel_trace.csv $O/pmc_k_$TAG $P/keyset
$K > $P/keyset/bench_same_box_unprofiled.json 2>/dev/null
# the key-set ladder over chunk tables (64 additions): the instruction counters only
C="$B --key-grouping keyset-chunks"
mkdir -p $P/keyset_chunks
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c_$TAG -o run -- $C > $P/keyset_chunks/bench_under_kernel_trace.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c_$TAG -o run -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c_$TAG -o run -- $C > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_c_$TAG -o run -- $C > /dev/null 2>&1
python3 tools/collect_traffic.py $O/pmc_fetch_c_$TAG $O/pmc_write_c_$TAG > $P/keyset_chunks/hbm_traffic.json
python3 tools/summarize_profiles_r02.py $O/prof_c_$TAG/run_kernel_trace.csv $O/pmc_c_$TAG $P/keyset_chunks
# the key-set ladder over 5-bit joint tables (26 additions): kernel trace, counters, traffic
W5="$B --key-grouping keyset5"
mkdir -p $P/keyset5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w_$TAG -o run -- $W5 > $P/keyset5/bench_under_kernel_trace.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_w_$TAG -o run -- $W5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_w_$TAG -o run -- $W5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc_w_$TAG -o run -- $W5 > /dev/null 2>&1
python3 tools/collect_traffic.py $O/pmc_fetch_w_$TAG $O/pmc_write_w_$TAG > $P/keyset5/hbm_traffic.json
python3 tools/summarize_profiles_r02.py $O/prof_w_$TAG/run_kernel_trace.csv $O/pmc_w_$TAG $P/keyset5
$W5 > $P/keyset5/bench_same_box_unprofiled.json 2>/dev/null
# the unprofiled bench right after, same box: the lines the profiles have to reconcile with
$B > $P/bench_same_box_unprofiled.json 2>/dev/null
$G > $P/general/bench_same_box_unprofiled.json 2>/dev/null
python3 tools/isa_count.py > $P/static_counts.json
tail -c 400 $P/bench_same_box_unprofiled.json; echo
tail -c 400 $P/general/bench_same_box_unprofiled.json; echo
cat $P/kernel_time_summary.json | head -40
python3 - <<PY
import json
a=json.load(open("$P/valu_counts.json")); b=json.load(open("$P/general/valu_counts.json"))
print({k: a[k]["valu_instr_per_signature"] for k in a if k.startswith("k_verify_fast")}, {k: b[k]["valu_instr_per_signature"] for k in b if k.startswith("k_verify_fast")})
PY
