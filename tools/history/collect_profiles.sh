#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/collect_profiles.sh TAG): kernel trace + stats of the
# bench command, then the counter passes, each in its own run (no trace domains besides
# --kernel-trace next to --pmc).  Outputs land in gpurun_out/<kind>_TAG.
TAG=${1:-r01x}
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out
mkdir -p $O
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie > $O/prof_${TAG}_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM --output-format csv -d $O/pmc_$TAG -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $O/pmc_${TAG}_bench.log 2>&1
python3 tools/collect_traffic.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG > $O/hbm_traffic_$TAG.json
python3 tools/summarize_pmc.py $O/pmc_$TAG > $O/pmc_$TAG.txt
tail -1 $O/prof_${TAG}_bench.log | cut -c1-250
cat $O/pmc_$TAG.txt | grep -E "k_verify_fast|k_scalar_prep"
cat $O/hbm_traffic_$TAG.json | head -30
