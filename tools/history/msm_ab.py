#!/usr/bin/env python3
"""2^20-term multiscalar multiplication (checked against the known discrete log) and 2^20-signature BIP-340
combination: time of each, for same-box A/B of library variants (S2K_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_msm_terms, synth_schnorr_batch
eng = S.Engine(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = 1 << 20
k, pts, tot = synth_msm_terms(eng, n, seed=7)
dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
out = torch.zeros(80, dtype=torch.uint8, device=dev)
ms = [round(timed(lambda: eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), out.data_ptr(), st), 10), 4) for _ in range(3)]
ok = out[:65].cpu().numpy().tobytes() == eng.scalar_base_mult_batch([tot.to_bytes(32, "big")])[0].tobytes()
pk, msgs, sig = synth_schnorr_batch(eng, n, 1 << 16, seed=340)
dpk, dmsg, dsig = (torch.from_numpy(x).to(dev) for x in (pk, msgs, sig))
res = ctypes.c_int(0)
seed = np.frombuffer(os.urandom(32), np.uint8)
rl = [round(timed(lambda: eng._lib.s2k_schnorr_batch_verify_rlc_device(eng._h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(), seed.ctypes.data, ctypes.byref(res), st), 5), 4) for _ in range(3)]
print(os.path.basename(os.environ.get("S2K_LIB", "default")), "msm_ms", ms, "check", ok, "rlc_ms", rl, "accept", res.value)
