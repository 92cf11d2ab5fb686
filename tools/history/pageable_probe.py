"""Why is submit / wait from PAGEABLE memory slower inside bench.py than in a bare script?  Times the submit call itself (the
host blocks while the runtime stages a pageable copy) and the whole pipeline, with and without torch in the process.
python tools/pageable_probe.py [torch|notorch] [queues]"""
import json
import os
import sys
import time

os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[2] if len(sys.argv) > 2 else "8"
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
with_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
if with_torch:
    import torch
    torch.cuda.init()
    x = torch.zeros(1 << 20, device="cuda")
    torch.cuda.synchronize()
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

n = 1 << 20
eng = S.Engine(0)
base = [np.array(a) for a in synth_batch(eng, n, 1 << 16, seed=1)]
bufs = [base] + [[a.copy() for a in base] for _ in range(2)]
res = {"torch": with_torch, "queues": os.environ["GPU_MAX_HW_QUEUES"]}
for rep in range(3):
    tickets, sub_ms = [], []
    t0 = time.perf_counter()
    for k in range(12):
        t1 = time.perf_counter()
        tickets.append(eng.ecdsa_verify_batch_submit(*bufs[k % 3]))
        sub_ms.append((time.perf_counter() - t1) * 1e3)
        if len(tickets) >= 3:
            tickets.pop(0).wait()
    for t in tickets:
        t.wait()
    res["rep%d" % rep] = {"ms_per_batch": (time.perf_counter() - t0) * 1e3 / 12, "submit_ms": [round(x, 2) for x in sub_ms]}
print(json.dumps(res))
