"""Host-buffer entry point: ms per 2^20 batch from pageable memory, best and median of 7 calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
eng = S.Engine(0)
n = 1 << 20
pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=3)
for mode, name in ((S.KEYS_OFF, "off"), (S.KEYS_AUTO, "auto")):
    eng.set_key_grouping(mode)
    eng.ecdsa_verify_batch(pub, dig, r, s)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); v = eng.ecdsa_verify_batch(pub, dig, r, s); ts.append((time.perf_counter() - t0) * 1e3)
    assert v.all()
    print(name, "best %.2f median %.2f ms" % (min(ts), sorted(ts)[3]), flush=True)
