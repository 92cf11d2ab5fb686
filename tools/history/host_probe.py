"""Host-buffer entry point s2k_ecdsa_verify_batch: ms per 2^20 batch from pageable, registered and pinned (s2k_host_alloc)
memory, key grouping off and on; best and median of 7 calls.  Verdicts compared across the three kinds of memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
eng = S.Engine(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
arrs = synth_batch(eng, n, max(n >> 4, 1), seed=3)
arrs[2][5, 7] ^= 1                          # one bad signature: verdicts are not all ones
pinned = [S.pinned_array(a.shape) for a in arrs]
for d, a in zip(pinned, arrs):
    d[...] = a
registered = [S.page_aligned_array(a.shape) for a in arrs]   # whole pages (s2k_host_register refuses heap blocks)
for d, a in zip(registered, arrs):
    d[...] = a
ref = None
for mode, name in ((S.KEYS_OFF, "grouping off"), (S.KEYS_AUTO, "grouping auto")):
    eng.set_key_grouping(mode)
    for kind, bufs in (("pageable", arrs), ("registered", registered), ("pinned", pinned)):
        if kind == "registered":
            for a in registered:
                S.host_register(a)
        eng.ecdsa_verify_batch(*bufs)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); v = eng.ecdsa_verify_batch(*bufs); ts.append((time.perf_counter() - t0) * 1e3)
        if kind == "registered":
            for a in registered:
                S.host_unregister(a)
        if ref is None:
            ref = v
        assert np.array_equal(v, ref) and int(v.sum()) == n - 1
        print("%-14s %-10s best %.2f median %.2f ms" % (name, kind, min(ts), sorted(ts)[3]), flush=True)
