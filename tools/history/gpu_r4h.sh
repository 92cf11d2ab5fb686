#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
for rep in 1 2; do for g in 0 1; do echo "no_gate=$g"; S2K_SUBMIT_NO_GATE=$g python tools/pipeline_trace.py 24 | tail -1; done; done
echo "w3 variant"; for rep in 1 2; do for g in 0 1; do echo "no_gate=$g"; S2K_LIB=$PWD/secp256k1_voi_amd/libsecp256k1_voi_amd.w3.so S2K_SUBMIT_NO_GATE=$g python tools/pipeline_trace.py 24 | tail -1; done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pl -- python3 $GRAFT_REPO_ROOT/tools/pipeline_trace.py 16 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_pipeline_trace.py $GRAFT_REPO_ROOT/$O/prof/pl_kernel_trace.csv | head -8
