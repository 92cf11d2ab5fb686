"""submit / wait from pinned buffers, 3 in flight, for a kernel trace (rocprofv3 --kernel-trace --memory-copy-trace):
python tools/pipeline_trace.py [batches]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

n = 1 << 20
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = S.Engine(0)
base = [np.array(a) for a in synth_batch(eng, n, 1 << 16, seed=1)]
pin = [[S.pinned_array(a.shape) for a in base] for _ in range(3)]
for q in pin:
    for d, a in zip(q, base):
        d[...] = a
outs = [S.pinned_array((n,)) for _ in range(3)]
for rep in range(2):
    tickets = []
    t0 = time.perf_counter()
    for k in range(nb):
        tickets.append(eng.ecdsa_verify_batch_submit(*pin[k % 3], out=outs[k % 3]))
        if len(tickets) >= 3:
            tickets.pop(0).wait()
    for t in tickets:
        t.wait()
    print("rep", rep, "ms per batch", (time.perf_counter() - t0) * 1e3 / nb, flush=True)
