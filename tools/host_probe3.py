"""Pinned host path after bench-like activity (bisect)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
from secp256k1_voi_amd.sharding import gather_valid_device
eng = S.Engine(0)
n = 1 << 20
arrs = synth_batch(eng, n, 1 << 16, seed=0x5EC9)
pinned = [S.pinned_array(a.shape) for a in arrs]
for d, a in zip(pinned, arrs):
    d[...] = a

def measure(tag, bufs=pinned):
    eng.ecdsa_verify_batch(*bufs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); v = eng.ecdsa_verify_batch(*bufs); ts.append((time.perf_counter() - t0) * 1e3)
    assert v.all()
    print("%-44s median %.2f ms" % (tag, sorted(ts)[2]), flush=True)

measure("pinned, fresh")
measure("pageable x6", arrs)
measure("pinned after pageable x6")
dev = torch.device("cuda", 0)
d = [torch.from_numpy(a).to(dev) for a in arrs]
valid = torch.zeros(n, dtype=torch.uint8, device=dev)
bitmap = torch.zeros(n // 8, dtype=torch.uint8, device=dev)
count = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
eng.profile(True)
for _ in range(30):
    valid.zero_()
    eng.ecdsa_verify_batch_device(n, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), valid.data_ptr(), 0, st)
    gather_valid_device(valid, n, None, eng, bitmap, count)
torch.cuda.synchronize()
pr = eng.profile_read_stages(cap=64)
measure("pinned after 30 bench steps + profile read")
measure("pageable again", arrs)
measure("pinned again")
