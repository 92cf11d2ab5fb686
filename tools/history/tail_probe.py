#!/usr/bin/env python3
"""How much does the last, partly filled round of k_verify_fast cost?  Times batches that are whole
rounds (multiples of 3 waves x 1024 SIMDs x 64 lanes = 196608) against 2^20."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

eng = S.Engine(0, wait_tables=True)      # (the wide generator tables are built in the background: a measurement waits for them)
dev = torch.device("cuda", 0)
N = 6 * 196608
pub, dig, r, s = synth_batch(eng, N, 1 << 16, seed=3)
d = [torch.from_numpy(x).to(dev) for x in (pub, dig, r, s)]
valid = torch.zeros(N, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for n in (196608 * 5, 1 << 20, 196608 * 6, 196608 * 5, 1 << 20):
    for _ in range(3):
        eng.ecdsa_verify_batch_device(n, *[x.data_ptr() for x in d], valid.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.ecdsa_verify_batch_device(n, *[x.data_ptr() for x in d], valid.data_ptr(), 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"n={n}: {dt * 1e3:.3f} ms, {dt / n * 1e9:.3f} ns per signature")
