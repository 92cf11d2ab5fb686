"""Why is a group of one member slower than the context's own pipeline?  The same pipelined loop (a) on the main thread,
(b) on another Python thread, (c) through a group of one."""
import json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from boundary_probe import pipelined

n = 1 << 20
eng = S.Engine(0)
base = [np.array(a) for a in synth_batch(eng, n, 1 << 16, seed=1)]
pin = [[S.pinned_array(a.shape) for a in base] for _ in range(4)]
for q in pin:
    for d, a in zip(q, base):
        d[...] = a
outs = [S.pinned_array((n,)) for _ in range(4)]
res = {}

def direct(e):
    pipelined(lambda k: e.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 8, 4)
    return sorted(pipelined(lambda k: e.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 16, 4) for _ in range(3))[1]

res["direct_main_thread"] = direct(eng)
box = {}
def worker():
    e2 = S.Engine(0)
    box["v"] = direct(e2)
    e2.close()
t = threading.Thread(target=worker); t.start(); t.join()
res["direct_other_thread_own_context"] = box["v"]
g = S.Group([0])
pipelined(lambda k: g.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 8, 4)
res["group_of_one"] = sorted(pipelined(lambda k: g.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), 16, 4) for _ in range(3))[1]
res["group_member_stats"] = g.member_stats()
g.close()
res["direct_main_thread_again"] = direct(eng)
print(json.dumps(res))
