#!/usr/bin/env python3
"""Throughput of the encoded entry point (SEC1 keys + DER signatures, as secec.PublicKey.Verify takes
them) at 2^20 items, next to the raw host-buffer entry point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

eng = S.Engine(0)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=5)


def der_int(b):
    b = bytes(b).lstrip(b"\0") or b"\0"
    if b[0] & 0x80:
        b = b"\0" + b
    return b"\x02" + bytes([len(b)]) + b


sigs, pubs = [], []
for i in range(n):
    body = der_int(r[i]) + der_int(s[i])
    sigs.append(b"\x30" + bytes([len(body)]) + body)
    pubs.append(b"\x04" + bytes(pub[i]))
digs = [bytes(d) for d in dig]
pb, po = S._concat(pubs)
db, do = S._concat(digs)
sb, so = S._concat(sigs)
out = np.zeros(n, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    eng._check(eng._lib.s2k_ecdsa_verify_encoded_batch(eng._h, n, pb.ctypes.data, po.ctypes.data, db.ctypes.data, do.ctypes.data,
                                                       sb.ctypes.data, so.ctypes.data, 0, 32, 0, out.ctypes.data))
    dt = time.perf_counter() - t0
    assert out.all()
    print(f"encoded (DER + SEC1): {dt * 1e3:.1f} ms, {n / dt:.3e} /s")
t0 = time.perf_counter()
v = eng.ecdsa_verify_batch(pub, dig, r, s)
print(f"raw host buffers:     {(time.perf_counter() - t0) * 1e3:.1f} ms")
