"""Busy time of the device in a pipelined run from a rocprofv3 kernel trace: union of the kernel intervals over the last
stretch of the run, per-kernel mean durations, idle gaps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]) for r in rows)
# the second repetition: the last 16 ladder launches
lad = [x for x in iv if x[2].startswith("k_verify_fast<4>")]
t_lo, t_hi = lad[-14][0], lad[-2][1]            # 12 batches in steady state
sel = [x for x in iv if x[0] >= t_lo and x[1] <= t_hi]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t_hi - t_lo
per = defaultdict(list)
for s, e, k in sel:
    per[k].append((e - s) / 1e6)
print("window %.3f ms for 12 ladders = %.3f ms per batch; device busy %.1f %%" % (wall / 1e6, wall / 1e6 / 12, 100.0 * busy / wall))
tot = 0
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s calls %3d  mean %.4f ms  sum per batch %.4f" % (k, len(v), sum(v) / len(v), sum(v) / 12))
    tot += sum(v) / 12
print("sum of kernel durations per batch %.3f ms" % tot)
