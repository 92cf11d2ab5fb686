"""Stage timings of the verification call with key grouping off / on, by signatures per key.
python tools/keyed_probe.py [batch_log2] -> one JSON line per (keys, mode)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    keys_logs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 14, 16, 17, 18, 20]
    n = 1 << lg
    dev = torch.device("cuda", 0)
    eng = S.Engine(0)
    if os.environ.get("S2K_GP_FIRST_PERCENT"):
        print(json.dumps({"gp_first_percent": int(os.environ["S2K_GP_FIRST_PERCENT"])}), flush=True)
    for kl in keys_logs:
        nkeys = 1 << kl if kl <= 30 else kl          # (values above 30: the number of keys itself)
        if nkeys > n:
            continue
        inp = [torch.from_numpy(x).to(dev) for x in synth_batch(eng, n, nkeys, seed=5)]
        out = torch.empty(n, dtype=torch.uint8, device=dev)
        ref = None
        modes = ((S.KEYS_OFF, "off"), (S.KEYS_AUTO, "auto"), (S.KEYS_ALWAYS, "always"))
        if os.environ.get("PROBE_MODES"):
            modes = [m for m in modes if m[1] in os.environ["PROBE_MODES"].split(",")]
        for mode, name in modes:
            eng.set_key_grouping(mode)
            for _ in range(3):
                eng.ecdsa_verify_batch_device(n, *(x.data_ptr() for x in inp), out.data_ptr())
            torch.cuda.synchronize()
            eng.profile(True)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            reps = 10
            t0.record()
            for _ in range(reps):
                eng.ecdsa_verify_batch_device(n, *(x.data_ptr() for x in inp), out.data_ptr())
            t1.record()
            torch.cuda.synchronize()
            p = eng.profile_read_stages()
            eng.profile(False)
            st = eng.key_grouping_stats()
            v = out.cpu().numpy()
            if ref is None:
                ref = v
            rec = {"n_log2": lg, "keys_log2": kl, "mode": name, "ms": t0.elapsed_time(t1) / reps,
                   "stages_ms": {k: round(p[k] / p["calls"], 4) for k in ("prep_ms", "group_ms", "fast_ms", "left_ms", "fallback_ms")},
                   "mhz": round(p["shader_mhz"]), "stats": st, "all_valid": bool(v.all()), "same_as_off": bool(np.array_equal(v, ref))}
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
