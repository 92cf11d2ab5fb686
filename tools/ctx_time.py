#!/usr/bin/env python3
"""From nothing to the first verdict: s2k_ctx_create (narrow generator tables), a small verification, then the wait for the
wide tables the background thread builds.  Prints ONE JSON line.

    python tools/ctx_time.py [--budget-gib G] [--gt-bits B]

--budget-gib: s2k_set_generator_table_budget (which width the automatic choice lands on under a memory limit);
--gt-bits: a context with tables of exactly that width (s2k_ctx_create_ex), built synchronously."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import secp256k1_voi_amd as S

ap = argparse.ArgumentParser()
ap.add_argument("--budget-gib", type=float, default=0.0)
ap.add_argument("--gt-bits", type=int, default=0)
a = ap.parse_args()
lib = S.load_library()
assert S.device_count() >= 1                      # (the runtime is initialised here, outside the clock)
if a.budget_gib:
    lib.s2k_set_generator_table_budget(int(a.budget_gib * (1 << 30)))
import pyref as R
rng = np.random.default_rng(5)
keys = [(7 + i, R.mul(7 + i, R.G)) for i in range(4)]
n = 64
pub, dig, rr, ss = [], [], [], []
for i in range(n):
    d, Q = keys[i % 4]
    h = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    r, s = R.ecdsa_sign(d, h, 12345 + i)
    if i % 5 == 0:
        s = (s + 1) % R.N
    pub.append(R.b32(Q[0]) + R.b32(Q[1])); dig.append(h); rr.append(R.b32(r)); ss.append(R.b32(s))
exp = [0 if i % 5 == 0 else 1 for i in range(n)]
t0 = time.perf_counter()
e = S.Engine(0, gt_bits=a.gt_bits)
t1 = time.perf_counter()
got = e.ecdsa_verify_batch(pub, dig, rr, ss)
t2 = time.perf_counter()
first = e.gt_info()
bits_after = e.gt_wait()
t3 = time.perf_counter()
got2 = e.ecdsa_verify_batch(pub, dig, rr, ss)
after = e.gt_info()
print(json.dumps({"ctx_create_s": round(t1 - t0, 4), "create_to_first_verdict_s": round(t2 - t0, 4), "gt_bits_first_call": first["bits"],
                  "gt_bits_target": first["target_bits"], "wide_tables_ready_after_s": round(t3 - t0, 3), "gt_bits_after": bits_after,
                  "gt_bytes_after": after["bytes"], "note": after["note"], "verdicts_ok": list(got) == exp and list(got2) == exp,
                  "budget_gib": a.budget_gib, "gt_bits_asked": a.gt_bits}))
