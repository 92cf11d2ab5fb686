"""A group of one member in a process that holds no other context (synthetic inputs copied from a file-free generator:
the group's own member cannot synthesise, so a temporary engine makes the batch and is closed before the group exists)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boundary_probe import pipelined

n = 1 << 20
eng = S.Engine(0)
base = [np.array(a) for a in synth_batch(eng, n, 1 << 16, seed=1)]
eng.close()
del eng
pin = [[S.pinned_array(a.shape) for a in base] for _ in range(4)]
for q in pin:
    for d, a in zip(q, base):
        d[...] = a
outs = [S.pinned_array((n,)) for _ in range(4)]
g = S.Group([0])
pipelined(lambda k: g.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 8, 4)
ms = sorted(pipelined(lambda k: g.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 16, 4) for _ in range(3))
print(json.dumps({"queues": os.environ["GPU_MAX_HW_QUEUES"], "group_of_one_alone_ms": ms}))
g.close()
