#!/bin/bash
# round 5, call q: small synchronous host calls without DMA transfers (ctx_small_block) - parity, randomised run, host-to-host times
# with and without (S2K_SMALL_CALLS_DMA=1: the transfers as before)
mkdir -p gpurun_out/r5q
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_c_harness.py -x -q -m gpu -k "small_batch or recover or smoke or harness or schnorr or wycheproof or kats or random_batches" 2>&1 | tail -3
timeout 900 python3 tools/stress_small.py 200 82 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
python3 - > gpurun_out/r5q/host_calls.txt 2>&1 <<'PY'
import os, sys, time, json, subprocess
code = r'''
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch, synth_schnorr_batch
eng = S.Engine(0, wait_tables=True)
pub, dig, r, s = synth_batch(eng, 4096, 4096, seed=5)
pk, msgs, sig = synth_schnorr_batch(eng, 4096, 4096, 9)
rid = np.zeros(4096, np.uint8)
out = {}
for n in (1, 64, 256, 1024, 2048, 4096):
    row = {}
    for what, call in (("ecdsa", lambda: eng.ecdsa_verify_batch(pub[:n], dig[:n], r[:n], s[:n])),
                       ("schnorr", lambda: eng.schnorr_verify_batch(pk[:n], msgs[:n], sig[:n])),
                       ("recover", lambda: eng.ecdsa_recover_batch(dig[:n], r[:n], s[:n], rid[:n])[1])):
        ts = []
        for i in range(40):
            t0 = time.perf_counter(); res = call(); ts.append((time.perf_counter() - t0) * 1e3)
        row[what + "_ms"] = round(float(np.median(ts[8:])), 4)
        row[what + "_ok"] = int(np.asarray(res).sum())
    out[n] = row
print(json.dumps(out))
'''
for knob in ("0", "1"):
    env = dict(os.environ, S2K_SMALL_CALLS_DMA=knob)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("S2K_SMALL_CALLS_DMA=%s" % knob, line[-1] if line else p.stderr[-400:])
PY
cat gpurun_out/r5q/host_calls.txt
