#!/bin/bash
# Run on the GPU box:  gpurun -- bash tools/collect_msm_profiles.sh TAG
# BASELINE configs 3 (2^20-term multiscalar multiplication) and 4 (2^20-signature BIP-340 combination): rocprofv3
# kernel trace + stats, then the counter passes, each in its own run (only --kernel-trace next to --pmc).
# Raw output: gpurun_out/msmprof_TAG/; the summaries to commit: gpurun_out/profiles_TAG/msm_*.json, rlc_*.json.
TAG=${1:-r03x}
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
O=$REPO/gpurun_out/msmprof_$TAG
P=$REPO/gpurun_out/profiles_$TAG
mkdir -p $O $P
cd $REPO
for W in msm rlc; do
  C="python3 tools/profile_msm.py $W 8"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_trace -o run -- $C > $O/${W}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${W}_fetch -o run -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${W}_write -o run -- $C > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/${W}_pmc -o run -- $C > $O/${W}_pmc.log 2>&1
  cp $O/${W}_trace/run_kernel_stats.csv $P/${W}_kernel_stats_2p20.csv
  python3 tools/summarize_msm_profile.py $W $O > $P/${W}_profile_2p20.json
  tail -1 $O/${W}_trace.log
  python3 - <<PY
import json
d = json.load(open("$P/${W}_profile_2p20.json"))
print("$W: call %.3f ms kernels, %.3f ms span" % (d["sum_kernel_ms_per_call"], d["span_ms_per_call"]))
for k, v in d["kernels"].items():
    print("  %-28s %5.1f launches/call %8.3f ms/call  valu %8.1f M  issue %.2f  fetch %7.1f MB write %7.1f MB" % (k, v["launches_per_call"], v["ms_per_call"], (v.get("valu_wave_instr_per_call") or 0) / 1e6, v.get("issue_frac") or 0, (v.get("fetch_bytes_per_call") or 0) / 1e6, (v.get("write_bytes_per_call") or 0) / 1e6))
PY
done
