#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 passes tools/collect_msm_profiles.sh makes over tools/profile_msm.py:

    python3 tools/summarize_msm_profile.py msm|rlc <dir with <what>_trace, _fetch, _write, _pmc>  > profile.json

Per kernel and per CALL of the entry point (the first call of the run, which allocates and warms up, is left out):
launches, milliseconds (kernel trace), VALU wave-instructions (SQ_INSTS_VALU), fetched / written bytes
(FETCH_SIZE / WRITE_SIZE, KiB -> bytes; fetch doubled per the gfx950 note of MI355X_MICROARCH.md), and the issue
fraction = wave-instructions x 4 cycles / (1024 SIMDs x kernel time x clock), clock from GRBM_GUI_ACTIVE of the
counter pass (the counter sums the 8 XCDs)."""
import csv
import glob
import json
import sys
from collections import defaultdict, OrderedDict

SKIP = ("k_gen_gtable", "k_point_op", "k_fn_op", "k_scalar_base", "k_hot_prep", "k_affine_finish", "k_verify_fast", "k_point_fallback")


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def rows(d):
    out = []
    for fn in glob.glob(d + "/**/*.csv", recursive=True):
        if fn.endswith("kernel_trace.csv") or fn.endswith("counter_collection.csv"):
            out += list(csv.DictReader(open(fn)))
    return out


def main():
    what, base = sys.argv[1], sys.argv[2]
    calls = 8
    # kernel trace: time order; the marker of a call is the first kernel of the path
    tr = [r for r in rows(f"{base}/{what}_trace") if "Start_Timestamp" in r and "Counter_Name" not in r]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks3 = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr]
    ks3 = [x for x in ks3 if x[0].startswith("k_") and not x[0].startswith(SKIP)]
    ks = [(k, (e - b) / 1e6) for k, b, e in ks3]
    # a call ends with k_msm_final: call i = the kernels after the previous k_msm_final up to and including this one
    ends = [i for i, (k, _) in enumerate(ks) if k.startswith("k_msm_final")]
    ends = ends[-calls:]
    timed = ks[ends[0] + 1:ends[-1] + 1]          # drop the first call (it allocates and warms up)
    ncall = len(ends) - 1
    # wall span of a call: first kernel start to k_msm_final end (kernels of two streams overlap in the BIP-340 flow)
    spans = [(ks3[ends[i + 1]][2] - min(b for _, b, _ in ks3[ends[i] + 1:ends[i + 1] + 1])) / 1e6 for i in range(ncall)]
    ms = defaultdict(float)
    cnt = defaultdict(int)
    order = []
    for k, d in timed:
        if k not in ms:
            order.append(k)
        ms[k] += d
        cnt[k] += 1

    def counter(sub, name):
        acc = defaultdict(list)
        for r in rows(f"{base}/{what}_{sub}"):
            if r.get("Counter_Name") == name:
                acc[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
        res = {}
        for k, v in acc.items():
            v.sort()
            per = cnt.get(k, 0) // max(ncall, 1)
            keep = v[-per * ncall:] if per else v
            res[k] = sum(x for _, x in keep) / max(ncall, 1)
        return res

    valu = counter("pmc", "SQ_INSTS_VALU")
    gui = counter("pmc", "GRBM_GUI_ACTIVE")
    fetch = counter("fetch", "FETCH_SIZE")
    write = counter("write", "WRITE_SIZE")
    # clock: GRBM_GUI_ACTIVE cycles / kernel duration in the counter pass
    pm = [r for r in rows(f"{base}/{what}_pmc") if r.get("Counter_Name") == "GRBM_GUI_ACTIVE"]
    dur_pmc = defaultdict(float)
    gui_sum = defaultdict(float)
    for r in pm:
        k = short(r["Kernel_Name"])
        dur_pmc[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
        gui_sum[k] += float(r["Counter_Value"])
    out = OrderedDict()
    out["what"] = {"msm": "s2k_multi_scalar_mult_device, 2^20 terms (BASELINE config 3)",
                   "rlc": "s2k_schnorr_batch_verify_rlc_device, 2^20 signatures of 2^16 keys (BASELINE config 4)"}[what]
    out["calls_averaged"] = ncall
    out["sum_kernel_ms_per_call"] = sum(ms.values()) / ncall
    out["span_ms_per_call"] = sum(spans) / ncall
    kern = OrderedDict()
    tot_valu = 0.0
    for k in order:
        e = OrderedDict()
        e["launches_per_call"] = cnt[k] / ncall
        e["ms_per_call"] = ms[k] / ncall
        if k in valu:
            e["valu_wave_instr_per_call"] = valu[k]
            tot_valu += valu[k]
            big = dur_pmc.get(k, 0) > 0 and gui_sum.get(k, 0) > 0
            clock = gui_sum[k] / dur_pmc[k] / 8.0 if big else None      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            if clock and clock > 5e8:
                e["clock_ghz_counter_pass"] = clock / 1e9
            ghz = (clock if clock and clock > 1.5e9 else 2.2e9)
            e["issue_frac"] = valu[k] * 4 / (1024 * (ms[k] / ncall) * 1e-3 * ghz)
        if k in fetch:
            e["fetch_bytes_per_call"] = 2 * fetch[k] * 1024
        if k in write:
            e["write_bytes_per_call"] = write[k] * 1024
        kern[k] = e
    out["valu_wave_instr_per_call"] = tot_valu
    out["whole_call_issue_frac_at_2p2ghz"] = tot_valu * 4 / (1024 * out["span_ms_per_call"] * 1e-3 * 2.2e9)
    out["kernels"] = kern
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
