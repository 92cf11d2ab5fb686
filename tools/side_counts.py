#!/usr/bin/env python3
"""VALU instruction counts of the ladders beside the headline: public-key recovery and per-signature
BIP-340 verification at 2^20 items.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d DIR -o run -- python3 tools/side_counts.py run
    python3 tools/side_counts.py summarize DIR [head] > profiles/r05_side_counts.json

`run`: three calls of s2k_ecdsa_recover_batch and of s2k_schnorr_verify_batch on 2^20 synthetic items of 2^16 keys.
`summarize`: SQ_INSTS_VALU x 64 / items per dispatch of k_verify_fast<RECOVER> (template argument 2), <SCHNORR_KEYED> (6) and
<SCHNORR> (1, the general ladder: the run is repeated with the key grouping off), averaged over the dispatches; bench.py prices
the live duration of those kernels with them (recover_2p20.roofline, schnorr_per_signature_2p20.roofline).  The same for the
wave-per-signature ladders of small calls (k_verify_row, k_schnorr_row, k_recover_row; 1024 items per dispatch): there the
figure is WAVE instructions per item - one wave works on one item -, what DESIGN 4d prices their latency with."""
import csv
import glob
import json
import os
import sys

N = 1 << 20
KERNELS = {"k_verify_fast<2>": "k_verify_fast_recover", "k_verify_fast<6>": "k_verify_fast_schnorr_keyed", "k_verify_fast<1>": "k_verify_fast_schnorr",
           "k_verify_fast<(int)2>": "k_verify_fast_recover", "k_verify_fast<(int)6>": "k_verify_fast_schnorr_keyed", "k_verify_fast<(int)1>": "k_verify_fast_schnorr"}


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch, synth_schnorr_batch
    eng = S.Engine(0, wait_tables=True)
    pub, dig, r, s = synth_batch(eng, N, 1 << 16, seed=5)
    rid = np.zeros(N, np.uint8)
    for _ in range(3):
        q, ok = eng.ecdsa_recover_batch(dig, r, s, rid)
    print("recover ok", int(ok.sum()))
    pk, msgs, sig = synth_schnorr_batch(eng, N, 1 << 16, seed=9)
    for mode in (S.KEYS_ADAPTIVE, S.KEYS_OFF):
        eng.set_key_grouping(mode)
        for _ in range(3):
            v = eng.schnorr_verify_batch(pk, msgs, sig)
        print("schnorr valid", int(v.sum()))
    eng.set_key_grouping(S.KEYS_ADAPTIVE)
    m = 1024
    for _ in range(3):
        v = eng.ecdsa_verify_batch(pub[:m], dig[:m], r[:m], s[:m])
        v2 = eng.schnorr_verify_batch(pk[:m], msgs[:m], sig[:m])
        q, ok = eng.ecdsa_recover_batch(dig[:m], r[:m], s[:m], rid[:m])
    print("small calls", int(v.sum()), int(v2.sum()), int(ok.sum()))
    m = 16384                             # four lanes per signature (k_verify_quad / k_schnorr_quad / k_recover_quad)
    for _ in range(3):
        v = eng.ecdsa_verify_batch(pub[:m], dig[:m], r[:m], s[:m])
        v2 = eng.schnorr_verify_batch(pk[:m], msgs[:m], sig[:m])
        q, ok = eng.ecdsa_recover_batch(dig[:m], r[:m], s[:m], rid[:m])
    print("mid-size calls", int(v.sum()), int(v2.sum()), int(ok.sum()))


ROW_KERNELS = ("k_verify_row", "k_schnorr_row", "k_recover_row")
QUAD_KERNELS = ("k_verify_quad", "k_schnorr_quad", "k_recover_quad")


def summarize(d, head):
    acc = {}
    rowk = {}
    quadk = {}
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(" ", "")
            key = KERNELS.get(name)
            if key and row["Counter_Name"] == "SQ_INSTS_VALU":
                acc.setdefault(key, []).append(float(row["Counter_Value"]))
            base = name.split("<")[0]
            if base in QUAD_KERNELS and row["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVES"):
                quadk.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if base in ROW_KERNELS and row["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVES"):
                rowk.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU of tools/side_counts.py run: 2^20 items of 2^16 keys per dispatch, wave-instructions x 64 / items, "
                     "averaged over the dispatches", "items_per_dispatch": N, "head": head}
    for key, vals in sorted(acc.items()):
        out[key] = {"valu_instr_per_item": sum(vals) / len(vals) * 64.0 / N, "dispatches": len(vals)}
    for name, c in sorted(rowk.items()):
        v, w = c.get("SQ_INSTS_VALU", []), c.get("SQ_WAVES", [])
        if v and w:
            # (1024 items per dispatch; five waves per four items since the preparation wave joined the kernels: its instructions
            # - one lane's scalar arithmetic for four signatures - are in the sum)
            out[name] = {"valu_wave_instr_per_item": sum(v) / (1024.0 * len(v)), "waves_per_dispatch": sum(w) / len(w), "dispatches": len(v)}
    for name, c in sorted(quadk.items()):
        v, w = c.get("SQ_INSTS_VALU", []), c.get("SQ_WAVES", [])
        if v and w:     # 16384 items per dispatch, 16 per wave
            out[name] = {"valu_wave_instr_per_item": sum(v) / (16384.0 * len(v)), "valu_wave_instr_per_wave": sum(v) / sum(w),
                         "waves_per_dispatch": sum(w) / len(w), "dispatches": len(v)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        run()
