#!/usr/bin/env python3
"""Multi-scalar multiplication time by size, for the window-width threshold (S2K_MSM_C16_FROM_LOG2)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
eng = S.Engine(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
N = 1 << 20
d = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); d[:, 0] &= 0x7F
k = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); k[:, 0] &= 0x7F
pts = eng.scalar_base_mult_batch(d)
dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
out = torch.zeros(80, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for lg in range(11, 21):
    n = 1 << lg
    f = lambda: eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), out.data_ptr(), st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    res[lg] = round(e0.elapsed_time(e1) / 5, 4)
print(os.environ.get("S2K_MSM_C16_FROM_LOG2", "default"), json.dumps(res))
