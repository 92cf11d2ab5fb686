#!/usr/bin/env python3
"""Reconciles a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5` with the bench line.

    python tools/summarize_profiles_r02.py <run_kernel_trace.csv> <pmc dir> <out dir>

Writes kernel_time_summary.json (per kernel: calls, mean / median / min / max ms over ALL launches and
over the TIMED launches only = the last 20 of the hot-path kernels, i.e. without the parity guards and
warm-up steps that include the clock ramp) and valu_counts.json (per-signature VALU instruction counts,
GRBM_GUI_ACTIVE-derived clock) in the format bench.py reads from profiles/.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

TIMED = 20


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("k_key_"):          # k_key_insert<64>, k_key_chain<false>: one instance per run
        n = n.split("<")[0]
    return n


def main():
    trace, pmc_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    dur = defaultdict(list)
    for row in csv.DictReader(open(trace)):
        dur[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    summ = {}
    for k, v in dur.items():
        if not k.startswith("k_"):
            continue
        v.sort()
        ms = [d for _, d in v]
        timed = ms[-TIMED:] if len(ms) >= TIMED else ms

        def stats(x):
            s = sorted(x)
            return {"calls": len(x), "mean_ms": sum(x) / len(x), "median_ms": s[len(s) // 2], "min_ms": s[0], "max_ms": s[-1]}
        summ[k] = {"all": stats(ms), "timed_last_%d" % TIMED: stats(timed)}
    json.dump(summ, open(out + "/kernel_time_summary.json", "w"), indent=1)

    acc = defaultdict(list)
    span = defaultdict(list)
    for fn in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            k = short(row["Kernel_Name"])
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                span[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e9)

    def avg(k, c, last=TIMED):
        v = acc.get((k, c), [])
        v = v[-last:] if len(v) >= last else v
        return sum(v) / len(v) if v else None
    res = {"source": "rocprofv3 --pmc (tools/collect_profiles_r02.sh), averages over the last %d dispatches, 2^20 signatures per dispatch" % TIMED}
    # the ladder kernels: template instance <0> = general, <4> = over per-key tables (all 2^20 lanes live in the bench)
    for inst, key in (("k_verify_fast<0>", "k_verify_fast"), ("k_verify_fast<4>", "k_verify_fast_keyed"), ("k_verify_fast<8>", "k_verify_fast_keyset"), ("k_verify_fast<9>", "k_verify_fast_keyset_joint"),
                      ("k_verify_fast<10>", "k_verify_fast_keyset_joint5"), ("k_verify_fast<11>", "k_verify_fast_keyset_joint6")):
        fast = next((k for k in set(k for k, _ in acc) if k.startswith(inst)), None)
        if not fast:
            continue
        valu, waves, gui = avg(fast, "SQ_INSTS_VALU"), avg(fast, "SQ_WAVES"), avg(fast, "GRBM_GUI_ACTIVE")
        sec = span[fast][-TIMED:]
        sec = sum(sec) / len(sec)
        res[key] = {"valu_instr_per_signature": valu / waves, "waves": waves,
                    "salu_instr_per_wave": avg(fast, "SQ_INSTS_SALU") / waves, "vmem_instr_per_wave": avg(fast, "SQ_INSTS_VMEM") / waves,
                    "grbm_gui_active": gui, "dispatch_seconds_under_pmc": sec,
                    # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles
                    "clock_mhz_from_gui_active": gui / 8 / sec / 1e6}
    # the other kernels of a step: VALU instructions per signature of the batch (wave counts differ per kernel)
    # (a kernel launched several times per step - the generator part runs in two pieces - counts with all its launches:
    # launches per step = its dispatches / the ladder's dispatches, averaged over the last TIMED steps' worth of them)
    steps_total = max((len(acc[(k, "SQ_INSTS_VALU")]) for k in set(k for k, _ in acc) if k.startswith("k_verify_fast<4>") or k.startswith("k_verify_fast<0>") or k.startswith("k_verify_fast<8>") or k.startswith("k_verify_fast<9>")), default=0)
    for k in sorted(set(k for k, _ in acc)):
        if k.startswith("k_verify_fast") or not k.startswith("k_"):
            continue
        nd = len(acc[(k, "SQ_INSTS_VALU")])
        per_step = nd // steps_total if steps_total and nd >= steps_total and nd % steps_total == 0 else 1
        v = avg(k, "SQ_INSTS_VALU", TIMED * per_step)
        if v:
            res[k] = {"valu_instr_per_signature": v * per_step * 64 / (1 << 20), "launches_averaged": min(TIMED * per_step, nd),
                      "launches_per_step": per_step}
    json.dump(res, open(out + "/valu_counts.json", "w"), indent=1)


if __name__ == "__main__":
    main()
