#!/usr/bin/env python3
"""Per-kernel, per-dispatch averages of the counters in a rocprofv3 --pmc output directory.

    python tools/summarize_pmc.py gpurun_out/pmc_r01d > profiles/r01_d_pmc.txt
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if name.startswith("k_"):
                acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:24s} {c:20s} {sum(v) / len(v):18.0f}   (n={len(v)})")


if __name__ == "__main__":
    main()
