#!/usr/bin/env python3
"""One synchronous device-resident call by size, wave-per-signature ladder (k_verify_row) against the lane-per-signature
kernels: ms per call (median of 30 after 5), verdicts compared between the two.  -> profiles/r05_small_batch_ab.txt"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import secp256k1_voi_amd as S
import oracle as O
from workload import make_ecdsa_batch


def main():
    O.build()
    orc = O
    eng = S.Engine(0, wait_tables=True)
    top = 1 << 14
    w = make_ecdsa_batch(orc, top, seed=5, n_keys=top, corrupt_every=7, low_s=False)
    dev = {k: torch.from_numpy(np.ascontiguousarray(w[k])).cuda() for k in ("pub", "digest", "r", "s")}
    out = torch.empty(top, dtype=torch.uint8, device="cuda")
    rows = []
    for lg in range(0, 15):
        n = 1 << lg
        res = {}
        for name, rm in (("row", 1 << 20), ("lane", 0)):
            eng.set_small_batch_max(rm)
            ts = []
            for i in range(35):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.ecdsa_verify_batch_device(n, dev["pub"].data_ptr(), dev["digest"].data_ptr(), dev["r"].data_ptr(), dev["s"].data_ptr(), out.data_ptr())
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[name] = (float(np.median(ts[5:])), out[:n].cpu().numpy().copy())
        same = bool(np.array_equal(res["row"][1], res["lane"][1]))
        rows.append({"log2_n": lg, "row_ms": round(res["row"][0], 4), "lane_ms": round(res["lane"][0], 4), "same_verdicts": same, "valid": int(res["row"][1].sum())})
        print(json.dumps(rows[-1]), flush=True)
    # BIP-340 verification and key recovery of the same sizes, host arrays to host results (synchronous calls)
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    pk, msgs, sig = synth_schnorr_batch(eng, 4096, 4096, 9)
    rid = np.zeros(4096, np.uint8)
    hw = {k: np.ascontiguousarray(w[k][:4096]) for k in ("digest", "r", "s")}
    for lg in (0, 6, 10, 11, 12):
        n = 1 << lg
        row = {"log2_n": lg}
        for name, rm in (("row", 1 << 20), ("lane", 0)):
            eng.set_small_batch_max(rm)
            for what, call in (("schnorr", lambda: eng.schnorr_verify_batch(pk[:n], msgs[:n], sig[:n])),
                               ("recover", lambda: eng.ecdsa_recover_batch(hw["digest"][:n], hw["r"][:n], hw["s"][:n], rid[:n])[1])):
                ts = []
                for i in range(25):
                    t0 = time.perf_counter()
                    res = call()
                    ts.append((time.perf_counter() - t0) * 1e3)
                row["%s_%s_ms" % (what, name)] = round(float(np.median(ts[5:])), 4)
                row["%s_%s_ok" % (what, name)] = int(np.asarray(res).sum())
        print(json.dumps(row), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
