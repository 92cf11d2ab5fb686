#!/usr/bin/env python3
"""Writes the profiles/ files of a round from a tools/collect_profiles_r04.sh run.

    python tools/merge_counts_r04.py gpurun_out/profiles_TAG r04 [letter] [commit the run was made at]

profiles/<round>_valu_counts.json: the PMC counts of both ladders (grouped run: k_verify_fast_keyed and the kernels around
it; grouping-off run: k_verify_fast), the static recount of the SAME library (tools/isa_count.py, run on the box), the
commit and a hash of the device sources they were taken at; <round>_hbm_traffic.json; the kernel-time summaries and
the bench lines of the same box."""
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    letter = sys.argv[3] if len(sys.argv) > 3 else "a"
    prof = os.path.join(ROOT, "profiles")
    grouped = json.load(open(os.path.join(src, "valu_counts.json")))
    general = json.load(open(os.path.join(src, "general", "valu_counts.json")))
    static = json.load(open(os.path.join(src, "static_counts.json")))
    prev = None
    for name in sorted(os.listdir(prof), reverse=True):
        if name.endswith("_valu_counts.json") and not name.startswith(rnd):
            prev = json.load(open(os.path.join(prof, name)))
            break
    out = dict(grouped)
    out["source"] = ("rocprofv3 --pmc of bench.py --steps 20 --warmup 5 (tools/collect_profiles_r04.sh): averages over the timed steps, 2^20 "
                     "signatures of 2^16 keys per dispatch; k_verify_fast (the general ladder) from the same passes with --key-grouping off")
    out["k_verify_fast"] = general["k_verify_fast"]
    for sub in ("keyset", "keyset_chunks", "keyset5"):
        ks_path = os.path.join(src, sub, "valu_counts.json")
        if os.path.exists(ks_path):
            ks = json.load(open(ks_path))
            for kn in ("k_verify_fast_keyset", "k_verify_fast_keyset_joint", "k_verify_fast_keyset_joint5", "k_verify_fast_keyset_joint6"):
                if kn in ks:
                    out[kn] = ks[kn]
    head = sys.argv[4] if len(sys.argv) > 4 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    out["head"] = head
    # whole step: every kernel of a grouped step, VALU instructions per verification
    whole = sum(v["valu_instr_per_signature"] for k, v in grouped.items()
                if isinstance(v, dict) and "valu_instr_per_signature" in v and k.startswith("k_") and
                k not in ("k_gen_gtable", "k_gen_gtable_bases", "k_fn_op", "k_point_op", "k_verify_fast"))
    for key, kname in (("static", "k_verify_fast"), ("static_keyed", "k_verify_fast_keyed"), ("static_keyset", "k_verify_fast_keyset"),
                       ("static_keyset_joint", "k_verify_fast_keyset_joint"), ("static_keyset_joint5", "k_verify_fast_keyset_joint5"),
                       ("static_keyset_joint6", "k_verify_fast_keyset_joint6")):
        if kname not in static:
            continue
        st = dict((prev or {}).get(key, {}))
        st["valu_instr_static"] = static[kname]["valu_instr_static"]
        st["mad_u64_u32_per_verify"] = static[kname]["mad_u64_u32_per_verify"]
        st["valu_per_trip"] = static[kname]["valu_per_trip"]
        st["recount"] = "tools/isa_count.py on the library the counters were read from (disassembly of the shipped code object)"
        out[key] = st
    out["static_keyed"]["valu_instr_whole_step_per_verify"] = round(whole)
    json.dump(out, open(os.path.join(prof, "%s_valu_counts.json" % rnd), "w"), indent=1)
    traffic = json.load(open(os.path.join(src, "hbm_traffic.json")))
    gtraffic = json.load(open(os.path.join(src, "general", "hbm_traffic.json")))
    if "k_verify_fast" in gtraffic:
        traffic["k_verify_fast"] = gtraffic["k_verify_fast"]
    for sub in ("keyset", "keyset_chunks", "keyset5"):
        kt_path = os.path.join(src, sub, "hbm_traffic.json")
        if os.path.exists(kt_path):
            kt = json.load(open(kt_path))
            for kn in ("k_verify_fast_keyset", "k_verify_fast_keyset_joint", "k_verify_fast_keyset_joint5", "k_verify_fast_keyset_joint6"):
                if kn in kt:
                    traffic[kn] = kt[kn]
    json.dump(traffic, open(os.path.join(prof, "%s_hbm_traffic.json" % rnd), "w"), indent=1)
    for a, b in (("kernel_time_summary.json", "%s_%s_kernel_time_summary.json"), ("kernel_stats_bench_steps20_warmup5.csv", "%s_%s_kernel_stats_bench_steps20_warmup5.csv"),
                 ("bench_same_box_unprofiled.json", "%s_%s_bench_same_box_unprofiled.json"), ("pmc_per_dispatch.txt", "%s_%s_pmc_per_dispatch.txt"),
                 ("general/kernel_time_summary.json", "%s_%s_general_kernel_time_summary.json"),
                 ("general/kernel_stats_bench_steps20_warmup5.csv", "%s_%s_general_kernel_stats_bench_steps20_warmup5.csv"),
                 ("general/bench_same_box_unprofiled.json", "%s_%s_general_bench_same_box_unprofiled.json"),
                 ("keyset/kernel_time_summary.json", "%s_%s_keyset_kernel_time_summary.json"),
                 ("keyset/bench_same_box_unprofiled.json", "%s_%s_keyset_bench_same_box_unprofiled.json"),
                 ("keyset5/kernel_time_summary.json", "%s_%s_keyset5_kernel_time_summary.json"),
                 ("keyset5/bench_same_box_unprofiled.json", "%s_%s_keyset5_bench_same_box_unprofiled.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(prof, b % (rnd, letter)))
    print("wrote profiles/%s_valu_counts.json: keyed %.0f, general %.0f, whole step %.0f (head %s)" %
          (rnd, out["k_verify_fast_keyed"]["valu_instr_per_signature"], out["k_verify_fast"]["valu_instr_per_signature"], whole, head))


if __name__ == "__main__":
    main()
