#!/usr/bin/env python3
"""How long s2k_ctx_destroy of the LAST context of a device takes while the background build of the wide generator tables
is at its various stages (the builder is cancelled and joined, never detached), and what a process pays that exits with
a context alive (the atexit handler).  One line per case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

e0 = S.Engine(0, gt_bits=20)
pub, dig, r, s = (np.array(x) for x in synth_batch(e0, 2048, 64, seed=5))
e0.close()
for delay in (0.0, 0.2, 0.6, 1.0, 1.5, 2.5, 4.0, 6.0):
    t0 = time.time()
    e = S.Engine(0)
    t1 = time.time()
    assert e.ecdsa_verify_batch(pub, dig, r, s).all()
    t2 = time.time()
    time.sleep(delay)
    info = e.gt_info()
    t3 = time.time()
    e.close()
    t4 = time.time()
    print("destroy %.1f s after the first call: create %.3f s, first verdicts %.3f s, tables then %d bits (building %s, %s), s2k_ctx_destroy %.3f s"
          % (delay, t1 - t0, t2 - t0, info["bits"], info["building"], info["note"], t4 - t3), flush=True)
e = S.Engine(0)
assert e.ecdsa_verify_batch(pub, dig, r, s).all()
t5 = time.time()
print("exit with a live context 0 s after its first call (atexit joins the builder): see the wall time of the process", flush=True)
