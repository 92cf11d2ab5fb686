#!/usr/bin/env python3
"""One-off check of unusual batch sizes through the host entry point: all-valid synthetic batches
with a seeded corruption mask; the verdict bitmap must equal the complement of the mask."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

eng = S.Engine(0)
for n in (1, 63, 65, 4097, (1 << 20) + 1, 1 << 22):
    pub, dig, r, s = synth_batch(eng, n, min(n, 1 << 12), seed=n)
    rng = np.random.default_rng(n)
    mask = rng.random(n) < 1 / 32
    s2 = s.copy()
    s2[mask, 31] ^= 1
    t0 = time.perf_counter()
    v = eng.ecdsa_verify_batch(pub, dig, r, s2)
    dt = time.perf_counter() - t0
    assert (v == (~mask).astype(np.uint8)).all(), n
    print(f"n={n}: ok, {int(v.sum())} valid, {dt * 1e3:.1f} ms")
