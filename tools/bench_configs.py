#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations on one GPU (not the bench.py metric):
  config 3: 2^20-term multi-scalar multiplication;  config 4: 2^20 BIP-340 signatures as one MSM;
  plus per-signature BIP-340 verification.  Inputs resident in HBM; HIP-event timing.
    python tools/bench_configs.py [--log2n 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import secp256k1_voi_amd as S


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    a = ap.parse_args()
    n = 1 << a.log2n
    dev = torch.device("cuda", 0)
    eng = S.Engine(0)
    rng = np.random.default_rng(7)
    out = {}

    # ---- MSM: points with known discrete logs ----
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); d[:, 0] &= 0x7F
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x7F
    pts = eng.scalar_base_mult_batch(d)
    dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
    dout = torch.zeros(80, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ms = timed(lambda: eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), dout.data_ptr(), st))
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    tot = sum(int.from_bytes(bytes(x), "big") * int.from_bytes(bytes(y), "big") for x, y in zip(k[:4096], d[:4096])) % N
    chk = eng.multi_scalar_mult(k[:4096], pts[:4096]) == eng.scalar_base_mult_batch([tot.to_bytes(32, "big")])[0].tobytes()
    out["msm"] = {"terms": n, "ms": ms, "terms_per_s": n / (ms * 1e-3), "prefix_check": bool(chk)}

    # ---- BIP-340: a few hundred signers, signatures made with the reference-independent signer ----
    import pyref as R
    base = 256
    keys = [int.from_bytes(rng.bytes(32), "big") % (R.N - 1) + 1 for _ in range(base)]
    trip = []
    for i in range(base):
        m = rng.bytes(32)
        trip.append((R.b32(R.mul(keys[i], R.G)[0]), m, R.schnorr_sign(keys[i], m, bytes(32))))
    idx = np.arange(n) % base
    pk = np.frombuffer(b"".join(t[0] for t in trip), np.uint8).reshape(base, 32)[idx]
    msg = np.frombuffer(b"".join(t[1] for t in trip), np.uint8).reshape(base, 32)[idx]
    sig = np.frombuffer(b"".join(t[2] for t in trip), np.uint8).reshape(base, 64)[idx]
    dpk, dmsg, dsig = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pk, msg, sig))
    dval = torch.zeros(n, dtype=torch.uint8, device=dev)
    lib, h = eng._lib, eng._h
    ms = timed(lambda: lib.s2k_schnorr_verify_batch_device(h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                           0, dval.data_ptr(), st))
    out["schnorr_single"] = {"sigs": n, "ms": ms, "sigs_per_s": n / (ms * 1e-3), "all_valid": bool(dval.all().item())}
    import ctypes
    res = ctypes.c_int(0)
    seed = np.frombuffer(os.urandom(32), np.uint8)
    ms = timed(lambda: lib.s2k_schnorr_batch_verify_rlc_device(h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32,
                                                               dsig.data_ptr(), seed.ctypes.data, ctypes.byref(res), st))
    out["schnorr_rlc_msm"] = {"sigs": n, "ms": ms, "sigs_per_s": n / (ms * 1e-3), "all_valid": bool(res.value)}

    # ---- public-key recovery: ECDSA signatures of the bench generator, ids found by trial ----
    from secp256k1_voi_amd.synth import synth_batch
    pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=11)
    rid = np.zeros(n, dtype=np.uint8)
    rec, ok = eng.ecdsa_recover_batch(dig, r, s, rid)
    rid[(rec[:, 1:] != pub).any(axis=1)] = 1
    dd, dr, ds, did = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (dig, r, s, rid))
    dpub65 = torch.zeros((n, 65), dtype=torch.uint8, device=dev)
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    ms = timed(lambda: lib.s2k_ecdsa_recover_batch_device(h, n, dd.data_ptr(), dr.data_ptr(), ds.data_ptr(), did.data_ptr(), 0,
                                                          dpub65.data_ptr(), dok.data_ptr(), st))
    match = bool((dpub65[:, 1:].cpu().numpy() == pub).all()) and bool(dok.all().item())
    out["ecdsa_recover"] = {"sigs": n, "ms": ms, "sigs_per_s": n / (ms * 1e-3), "all_recovered": match}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
