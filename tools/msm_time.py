#!/usr/bin/env python3
"""Times s2k_multi_scalar_mult_device (2^20 terms) and s2k_schnorr_batch_verify_rlc_device (2^20 signatures of 2^16
keys) with HIP events: median of 10 calls after 3 warm-up calls, result checked."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_msm_terms, synth_schnorr_batch

eng = S.Engine(0, wait_tables=True)      # (the wide generator tables are built in the background: a measurement waits for them)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)


def timed(f, reps=10, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


k, pts, tot = synth_msm_terms(eng, n, seed=7)
dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
out = torch.zeros(80, dtype=torch.uint8, device=dev)
med, best = timed(lambda: eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), out.data_ptr(), st))
got = bytes(out[:65].cpu().numpy())
exp = eng.scalar_base_mult_batch(np.frombuffer(int(tot).to_bytes(32, "big"), np.uint8).reshape(1, 32))[0].tobytes()
print("msm  median %.3f ms  best %.3f ms  ok=%s" % (med, best, got == exp))
pk, msgs, sig = synth_schnorr_batch(eng, n, max(n >> 4, 1), seed=340)
dpk, dmsg, dsig = (torch.from_numpy(x).to(dev) for x in (pk, msgs, sig))
res = ctypes.c_int(0)
seed = np.frombuffer(os.urandom(32), np.uint8)
f = lambda: eng._lib.s2k_schnorr_batch_verify_rlc_device(eng._h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                         seed.ctypes.data, ctypes.byref(res), st)
med, best = timed(f)
print("rlc  median %.3f ms  best %.3f ms  accept=%d" % (med, best, res.value))
