#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/r01_hbm_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128 B request for
wide streaming reads (guide: "reports exactly 1/2"); this kernel's reads are mostly 4 B/lane
gathers for which the counter is uncalibrated, so both the raw and the doubled figure are
kept and bench.py reports the conservative (doubled) one.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(dirname, counter):
    acc = defaultdict(list)
    for fn in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row["Counter_Name"] == counter:
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                if name.startswith("k_key_"):
                    name = name.split("<")[0]
                # ladder instances: <0> general, <4> over per-key tables, <5> general over the ungrouped rest
                name = name.replace("k_verify_fast<4>", "k_verify_fast_keyed").replace("k_verify_fast<5>", "k_verify_fast_left").replace("k_verify_fast<8>", "k_verify_fast_keyset").replace("k_verify_fast<9>", "k_verify_fast_keyset_joint").replace("k_verify_fast<10>", "k_verify_fast_keyset_joint5").replace("k_verify_fast<11>", "k_verify_fast_keyset_joint6")
                name = name.split("<")[0]          # the other template instances (k_verify_fast<0>) share an entry
                acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out[k] = {"fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                  "note": "FETCH_SIZE doubled per the gfx950 correction; upper estimate for non-streaming reads"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
