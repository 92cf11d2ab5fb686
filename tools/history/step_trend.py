#!/usr/bin/env python3
"""Per-step time over a long run (clock ramp / warm-up behaviour of the box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

eng = S.Engine(0)
dev = torch.device("cuda", 0)
n = 1 << 20
pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=3)
d = [torch.from_numpy(x).to(dev) for x in (pub, dig, r, s)]
valid = torch.zeros(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
time.sleep(2.0)          # let the device go idle first, as at the start of a fresh process
ts = []
for i in range(300):
    t0 = time.perf_counter()
    eng.ecdsa_verify_batch_device(n, *[x.data_ptr() for x in d], valid.data_ptr(), 0, st)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("steps 0-4:", [round(x, 2) for x in ts[:5]])
for a in (5, 10, 20, 50, 100, 200):
    print(f"mean of steps {a}..{a+10}: {sum(ts[a:a+10]) / 10:.3f} ms")
print(f"min {min(ts):.3f}  max {max(ts):.3f}")
