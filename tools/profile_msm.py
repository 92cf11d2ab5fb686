#!/usr/bin/env python3
"""2^20-term multiscalar multiplication and 2^20-signature BIP-340 combination, a few calls each, for
rocprofv3 --kernel-trace --stats (per-kernel breakdown of BASELINE configs 3 and 4)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_msm_terms, synth_schnorr_batch
eng = S.Engine(0)
dev = torch.device("cuda", 0)
n = 1 << 20
k, pts, tot = synth_msm_terms(eng, n, seed=7)
dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
out = torch.zeros(80, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), out.data_ptr(), st)
torch.cuda.synchronize()
pk, msgs, sig = synth_schnorr_batch(eng, n, 1 << 16, seed=340)
dpk, dmsg, dsig = (torch.from_numpy(x).to(dev) for x in (pk, msgs, sig))
res = ctypes.c_int(0)
seed = np.frombuffer(os.urandom(32), np.uint8)
for _ in range(6):
    eng._lib.s2k_schnorr_batch_verify_rlc_device(eng._h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                 seed.ctypes.data, ctypes.byref(res), st)
torch.cuda.synchronize()
print("ok", res.value)
