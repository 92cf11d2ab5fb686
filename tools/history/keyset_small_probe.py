import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
eng = S.Engine(0, wait_tables=True)
pub, dig, r, s = synth_batch(eng, 16384, 256, seed=5)
keys, inv = np.unique(pub, axis=0, return_inverse=True)
ks = eng.keyset_create(keys, S.KEYSET_JOINT)
kidx = inv.reshape(-1).astype(np.uint32)
for n in (1, 64, 1024, 4096, 16384):
    ts = []
    for i in range(30):
        t0 = time.perf_counter(); v = eng.ecdsa_verify_batch_keyset(ks, kidx[:n], dig[:n], r[:n], s[:n]); ts.append((time.perf_counter() - t0) * 1e3)
    print("keyset call of", n, "ms", round(float(np.median(ts[6:])), 4), int(v.sum()))
