#!/bin/bash
# final evidence pass of a round: profiles at the bench's own settings (kernel trace + PMC + traffic), the VALU
# microbenchmark, and the full check (all GPU tests + bench digest).  gpurun -- bash tools/gpu_final.sh TAG
TAG=${1:-r02z}
O=gpurun_out/final_$TAG; mkdir -p $O
timeout 900 bash tools/collect_profiles_r02.sh $TAG > $O/collect.log 2>&1; tail -5 $O/collect.log
timeout 300 ./tools/valu_rates2 > $O/valu_instruction_rates.txt 2>&1; tail -2 $O/valu_instruction_rates.txt | cut -c1-160
timeout 1500 bash tools/gpu_check.sh > $O/check.log 2>&1; tail -12 $O/check.log
