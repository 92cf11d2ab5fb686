#!/usr/bin/env python3
"""2^20-term multiscalar multiplication (BASELINE config 3) or 2^20-signature BIP-340 combination (config 4),
a few calls, for rocprofv3 (--kernel-trace --stats, or one --pmc pass): per-kernel breakdown.

    python3 tools/profile_msm.py msm|rlc [calls] [log2 n]
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_msm_terms, synth_schnorr_batch

what = sys.argv[1] if len(sys.argv) > 1 else "msm"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 20)
eng = S.Engine(0, wait_tables=True)      # (the wide generator tables are built in the background: a measurement waits for them)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
if what == "msm":
    k, pts, tot = synth_msm_terms(eng, n, seed=7)
    dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
    out = torch.zeros(80, dtype=torch.uint8, device=dev)
    for _ in range(calls):
        eng.multi_scalar_mult_device(n, dk.data_ptr(), dp.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    got = bytes(out[:65].cpu().numpy())
    exp = eng.scalar_base_mult_batch(np.frombuffer(int(tot).to_bytes(32, "big"), np.uint8).reshape(1, 32))[0].tobytes()
    print("msm ok", got == exp)
else:
    pk, msgs, sig = synth_schnorr_batch(eng, n, 1 << 16, seed=340)
    dpk, dmsg, dsig = (torch.from_numpy(x).to(dev) for x in (pk, msgs, sig))
    res = ctypes.c_int(0)
    seed = np.frombuffer(os.urandom(32), np.uint8)
    for _ in range(calls):
        eng._lib.s2k_schnorr_batch_verify_rlc_device(eng._h, n, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                     seed.ctypes.data, ctypes.byref(res), st)
    torch.cuda.synchronize()
    print("rlc ok", res.value)
