#!/bin/bash
# does the keyed ladder's table traffic cost time?  Same 2^20 signatures under 1, 64 and 2^16 keys
# (per-key tables of 9 KiB: fully cache resident for 1 and 64 keys): duration and FETCH_SIZE of k_verify_fast<ECDSA_KEYED>.
REPO=$PWD; O=$REPO/gpurun_out/lds_evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $REPO
export PROBE_MODES=auto
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o run -- python3 tools/keyed_probe.py 20 0,6,16 > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, json
rows = []
for fn in glob.glob("gpurun_out/lds_evidence/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(fn)) if r["Counter_Name"] == "FETCH_SIZE" and "k_verify_fast<4>" in r["Kernel_Name"].replace("(anonymous namespace)::", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 3
out = {}
for i, keys in enumerate((1, 64, 65536)):
    g = rows[i * per + 3:(i + 1) * per]          # drop the warm-up launches
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in g]
    fe = [float(r["Counter_Value"]) * 1024 * 2 for r in g]   # KiB -> bytes, doubled per the gfx950 note
    out["keys_%d" % keys] = {"launches": len(g), "ladder_ms_under_counters": sum(ms) / len(ms), "fetch_bytes_per_launch": sum(fe) / len(fe)}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/lds_evidence/summary.json", "w"), indent=1)
PY
grep -v amdgpu.ids $O/log.txt | grep '"mode"' | cut -c1-400 | tail -3
