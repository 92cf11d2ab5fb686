#!/bin/bash
# BIP-340 whole-batch path after a change: its tests, the randomised run, the call time and the kernel trace of one call
REPO=$PWD; O=gpurun_out/rlc_check; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -k "rlc or bisect or schnorr" -x 2>&1 | tail -2
timeout 600 python3 tools/stress_rlc.py 300 ${1:-151} 2>&1 | tail -1
for i in 1 2 3; do timeout 300 python3 tools/msm_time.py 2>&1 | grep "rlc"; done
cd /tmp && export TMPDIR=/tmp; cd $REPO
P=$REPO/$O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P -o run -- python3 tools/profile_msm.py rlc 6 > $P.log 2>&1
python3 tools/msm_timeline.py $P | grep -v "^$" | tail -32; rm -rf $P
