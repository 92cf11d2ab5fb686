#!/usr/bin/env python3
"""Two multiscalar calls in flight (two contexts, two host threads, two streams) against one call after the other:
wall time per call.  S2K_PKG_ROOT selects the tree whose package is imported (same-box A/B of two trees)."""
import os, sys, threading, time
ROOT = os.environ.get("S2K_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_msm_terms
dev = torch.device("cuda", 0)
eng = S.Engine(0, wait_tables=True)
m = 1 << 20
k, pts, tot = synth_msm_terms(eng, m, seed=7)
dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
eng_b = S.Engine(0)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
outs = [torch.zeros(80, dtype=torch.uint8, device=dev) for _ in range(2)]


def worker(e_, stream, o, reps):
    for _ in range(reps):
        e_.multi_scalar_mult_device(m, dk.data_ptr(), dp.data_ptr(), o.data_ptr(), stream.cuda_stream)


for e_, s_, o in zip((eng, eng_b), streams, outs):
    worker(e_, s_, o, 2)
torch.cuda.synchronize()
res = {}
for rep in range(3):
    t0 = time.perf_counter()
    worker(eng, streams[0], outs[0], 12)
    torch.cuda.synchronize()
    res.setdefault("one_after_the_other_ms", []).append(round((time.perf_counter() - t0) * 1e3 / 12, 3))
    th = [threading.Thread(target=worker, args=(e_, s_, o, 6)) for e_, s_, o in zip((eng, eng_b), streams, outs)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    res.setdefault("two_in_flight_ms_per_call", []).append(round((time.perf_counter() - t0) * 1e3 / 12, 3))
print(os.path.basename(ROOT) or ROOT, res)
