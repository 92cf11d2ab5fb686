#!/bin/bash
# Round 4, first GPU call: the new boundary tests, the C harness, boundary timings, K = N stage times with a kernel trace.
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_c_harness.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 600 python tools/boundary_probe.py 20 16 12 > $O/boundary.json 2> $O/boundary.err; echo "boundary rc=$?"; cat $O/boundary.json; tail -3 $O/boundary.err
PROBE_MODES=off,auto timeout 300 python tools/keyed_probe.py 20 20,16 2>/dev/null | cut -c1-400 | tee $O/kn_probe.jsonl
cd /tmp && export TMPDIR=/tmp
PROBE_MODES=off,auto timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o kn -- python3 $GRAFT_REPO_ROOT/tools/keyed_probe.py 20 20 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/$O/prof | head; python3 - <<'PY'
import csv,glob,os,collections
root=os.environ.get("GRAFT_REPO_ROOT",".")
for f in glob.glob(root+"/gpurun_out/r4a/prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(r.get("Name","")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"))
PY
