#!/bin/bash
REPO=$PWD; O=$REPO/gpurun_out/trace_host; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o run -- python3 tools/host_probe3.py > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt | tail -8
ls $O
