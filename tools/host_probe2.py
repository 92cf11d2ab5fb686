"""Bisecting why the pinned host path is slower inside bench.py than in tools/host_probe.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
eng = S.Engine(0)
n = 1 << 20
arrs = synth_batch(eng, n, n >> 4, seed=3)
pinned = [S.pinned_array(a.shape) for a in arrs]
for d, a in zip(pinned, arrs):
    d[...] = a

def measure(tag):
    eng.ecdsa_verify_batch(*pinned)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); v = eng.ecdsa_verify_batch(*pinned); ts.append((time.perf_counter() - t0) * 1e3)
    assert v.all()
    print("%-40s median %.2f ms" % (tag, sorted(ts)[2]), flush=True)

measure("fresh")
eng.profile(True)
measure("profile on")
eng.profile(False)
measure("profile off again")
dev = torch.device("cuda", 0)
d = [torch.from_numpy(a).to(dev) for a in arrs]
valid = torch.zeros(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    eng.ecdsa_verify_batch_device(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), valid.data_ptr(), 0, st)
torch.cuda.synchronize()
measure("after 30 device-pointer calls")
eng.ecdsa_verify_batch(*arrs)
measure("after a pageable call")
