import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
    x = torch.zeros(10, device="cuda")
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_schnorr_batch
eng = S.Engine(0)
n = 1 << 20
pk, msgs, sig = synth_schnorr_batch(eng, n, 1 << 16, seed=340)
def pinned(a):
    p = S.pinned_array(a.shape, a.dtype); p[...] = a; return p
h = [pinned(x) for x in (pk, msgs, sig)]
if len(sys.argv) > 2 and sys.argv[2] == "submit":     # create the submit / wait children first, as the bench has by then
    from secp256k1_voi_amd.synth import synth_batch
    b = [np.array(a) for a in synth_batch(eng, 1 << 16, 1 << 10, seed=1)]
    for _ in range(6):
        eng.ecdsa_verify_batch_submit(*b).wait()
ts = []
for i in range(8):
    t0 = time.perf_counter(); assert eng.schnorr_batch_verify_rlc(*h); ts.append((time.perf_counter() - t0) * 1e3)
print(sys.argv[1:], ["%.2f" % t for t in ts])
