#!/usr/bin/env python3
"""Timeline of the LAST s2k_multi_scalar_mult_device call of a rocprofv3 --kernel-trace run of tools/profile_msm.py:
start and end of every kernel relative to the call's first kernel, and the queue it ran on.

    python3 tools/msm_timeline.py <dir with the kernel trace csv>
"""
import csv
import glob
import sys


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows = [r for r in rows if short(r["Kernel_Name"]).startswith(("k_msm", "k_schnorr_rlc", "k_rlc", "k_key"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
firsts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) in ("k_msm_parse", "k_schnorr_rlc_prep<true>")]
lo = firsts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
end = 0
for r in rows[lo:]:
    b, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    end = max(end, e)
    print("%-32s queue %-3s %9.1f .. %9.1f us  (%7.1f)  grid %s wg %s lds %s" % (short(r["Kernel_Name"]), r.get("Queue_Id", "?"), b, e, e - b, r.get("Grid_Size", "?"),
                                                                            r.get("Workgroup_Size", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))))
print("span %.1f us, sum of kernels %.1f us" % (end, sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[lo:])))
