#!/usr/bin/env python3
"""One synchronous device-resident ECDSA call by size, 2^11 .. 2^17 signatures: the wave-per-signature kernel (row), the
four-lanes-per-signature kernel (quad) and the lane-per-signature kernels (lane), verdicts compared; keys: every one distinct, and
16 signatures per key.  -> profiles/r05_mid_batch_ab.txt"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def main():
    eng = S.Engine(0, wait_tables=True)
    top = 1 << 17
    for keys_name, nk_of in (("distinct", lambda n: n), ("16_per_key", lambda n: max(1, n // 16))):
        for lg in range(11, 18):
            n = 1 << lg
            pub, dig, r, s = synth_batch(eng, n, nk_of(n), seed=100 + lg)
            r[::7, 3] ^= 1
            dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (pub, dig, r, s)]
            out = torch.empty(n, dtype=torch.uint8, device="cuda")
            res = {}
            for name, rm, qm in (("row", 1 << 20, 0), ("quad", 0, 1 << 20), ("lane", 0, 0)):
                if name == "row" and n > (1 << 14):
                    continue
                eng.set_small_batch_max(rm)
                eng.set_mid_batch_max(qm)
                ts = []
                for i in range(25):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.ecdsa_verify_batch_device(n, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), out.data_ptr())
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                res[name] = (float(np.median(ts[5:])), out.cpu().numpy().copy())
            same = all(np.array_equal(res["lane"][1], v[1]) for v in res.values())
            row = {"keys": keys_name, "log2_n": lg, **{k + "_ms": round(v[0], 4) for k, v in res.items()}, "same_verdicts": bool(same),
                   "valid": int(res["lane"][1].sum())}
            print(json.dumps(row), flush=True)
    # BIP-340 verification and public-key recovery of the same sizes, host arrays to host results
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    for lg in (12, 13, 14, 15):
        n = 1 << lg
        pk, msgs, sig = synth_schnorr_batch(eng, n, n, 9 + lg)
        pub, dig, r, s = synth_batch(eng, n, n, seed=200 + lg)
        rid = np.zeros(n, np.uint8)
        row = {"log2_n": lg}
        keep = {}
        for name, qm in (("quad", 1 << 20), ("lane", 0)):
            eng.set_small_batch_max(0)
            eng.set_mid_batch_max(qm)
            for what, call in (("schnorr", lambda: eng.schnorr_verify_batch(pk, msgs, sig)),
                               ("recover", lambda: eng.ecdsa_recover_batch(dig, r, s, rid))):
                ts = []
                for i in range(15):
                    t0 = time.perf_counter()
                    res = call()
                    ts.append((time.perf_counter() - t0) * 1e3)
                row["%s_%s_ms" % (what, name)] = round(float(np.median(ts[4:])), 4)
                keep[(what, name)] = res
        row["same"] = bool(np.array_equal(keep[("schnorr", "quad")], keep[("schnorr", "lane")]) and
                           np.array_equal(keep[("recover", "quad")][0], keep[("recover", "lane")][0]) and
                           np.array_equal(keep[("recover", "quad")][1], keep[("recover", "lane")][1]))
        row["schnorr_valid"] = int(np.asarray(keep[("schnorr", "quad")]).sum())
        row["recovered"] = int(np.asarray(keep[("recover", "quad")][1]).sum())
        print(json.dumps(row), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
