import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
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
    print("%-44s median %.2f ms  all %s" % (tag, sorted(ts)[2], " ".join("%.2f" % t for t in ts)), flush=True)

for prof in (False, True, False):
    eng.profile(prof)
    for rep in range(2):
        measure("prof %s pinned" % prof)
        measure("prof %s pageable" % prof, arrs)
    measure("prof %s pinned" % prof)
