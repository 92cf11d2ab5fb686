#!/usr/bin/env python3
"""bench.py — secp256k1 ECDSA verifications/s at batch 2^20 per GPU (BASELINE.json metric).

One "step" = one pass of the hot path (s2k_ecdsa_verify_batch_device + the valid-bitmap
pack) over one batch of synthetic signatures per GPU, inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-log2 B] [--keys-log2 16]

N = 1: BASELINE config 2, 2^20 verifications on one MI355X.
N > 1: BASELINE config 5 shape, 2^21 verifications per GPU (2^24 over 8), one process per GPU
(torch.distributed, backend nccl = RCCL); every rank verifies its own contiguous shard (no
data-path collective) and per step the ranks all-gather the packed valid bitmap, each shard
with its rank's valid count behind it (one collective).  Started without a launcher (`python bench.py --gpus N`, no
RANK/WORLD_SIZE in the environment) this script starts its N rank processes itself, as fresh
children, before anything in the parent touches a GPU; started by torch.distributed.run it
is one of the ranks.  `--oversubscribe` (test hook for a 1-GPU box) puts rank r on device
r mod device_count and uses gloo for the collectives.

Rank 0 prints ONE JSON line.  Extra keys: "roofline" (the dominant kernel against the
integer-VALU issue roofline that bounds this path, live HIP-event kernel time and effective
shader clock; the HBM roofline the contract names as a sub-object), "cpu_baseline" (the CPU
oracle = port of the reference algorithm, timed on this box's host cores on a bounded sample;
the reference itself is Go and there is no Go toolchain here), and at N = 1 the other
BASELINE configurations, each guarded by a full-result check: "msm_2p20" (config 3),
"schnorr_rlc_2p20" (config 4), plus "distinct_keys" (K = N), "worst_case_all_fallback" and
"pcie_inclusive".  None of them is `value`.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import time

# The verification call overlaps two streams (grouping / tables beside preparation / generator part).  The HIP runtime
# spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues, four by default, and streams that share a queue
# run one after the other: with a process group in the process (RCCL brings streams of its own) the engine's two
# streams ended up on one queue and the step grew from 5.00 to 5.28 ms; with eight queues it is 5.03.
# Read by the runtime when it initialises, so set before anything touches the GPU; the package sets the same default.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_PER_VERIFY = 160 + 1          # r, s, digest (32 each) + pubkey (64) in, 1 byte out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4                     # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9
# Full-rate VALU issue: one wave64 instruction per 4 cycles per SIMD (16 lanes/clk).  Settled on the
# hardware with tools/valu_rates.hip (16 independent accumulators, 2/4/8 waves per SIMD):
# v_fma_f32 / v_fma_f64 / v_mul_lo_u32 / 64-bit shifts saturate at 1 per 4.3 cycles, v_mad_u64_u32 at
# 1 per 5.3, VOP2 add/and/sub/mov at 1 per 2.6 (profiles/r02_valu_instruction_rates.txt).
VALU_PEAK_LANE_OPS = SIMDS * 16 * PEAK_CLOCK_HZ


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-log2", type=int, default=None, help="signatures per GPU (default 20 at N=1, 21 at N>1)")
    ap.add_argument("--keys-log2", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe inclusive) measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip configs 3/4, K=N and the worst case (e.g. under a profiler)")
    ap.add_argument("--key-grouping", choices=("auto", "off", "keyset", "keyset-chunks", "keyset5", "keyset6"), default="auto",
                    help="off: every signature through the general ladder; keyset: the batch against a key set built before the "
                         "timed region (both for profiling those kernels under the driver's settings; not the headline)")
    ap.add_argument("--full", action="store_true", help="print the line with its explanations (as rounds 1-4 did) instead of the compact form")
    ap.add_argument("--write-notes", action="store_true", help="refresh bench_notes.json (the explanations, by path) from this run")
    ap.add_argument("--rank-timeout", type=float, default=600.0,
                    help="launcher (--gpus N without torchrun): seconds every rank has to report ready (process group, context, wide tables) "
                         "before the launch is abandoned and the missing ranks are named")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="test hook: allow more ranks than devices (rank r -> device r mod count, gloo collectives)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# launcher: N rank processes, started before this process has touched a GPU
# ---------------------------------------------------------------------------------------------
def launch(args):
    import torch                       # import only; device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    if ndev < args.gpus:
        if not args.oversubscribe:
            print(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible "
                  "(use --oversubscribe to share devices in a test)", file=sys.stderr)
            return 2
        env["S2K_DIST_BACKEND"] = "gloo"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(args.gpus),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    # Every rank is a fresh child of this process (which never touches a GPU).  The children's stderr comes through here line by
    # line, so that the launcher knows each rank's last words and when it said "ready" (process group up, context created, wide
    # tables built).  First rank to exit non-zero, or --rank-timeout seconds without every rank ready: the siblings are ended -
    # they would otherwise sit in init_process_group / the first all-gather until the collective's own timeout - and the
    # launcher says WHICH rank, with what code and what it printed last.
    import threading
    procs, last_line, ready = [], {}, set()
    lock = threading.Lock()

    def pump(rank, pipe):
        for raw in iter(pipe.readline, b""):
            text = raw.decode("utf-8", "replace").rstrip("\n")
            with lock:
                if text.strip():
                    last_line[rank] = text
                if text.startswith("bench.py: rank %d ready" % rank):
                    ready.add(rank)
            sys.stderr.write(text + "\n")
            sys.stderr.flush()
        pipe.close()

    for rank in range(args.gpus):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank), S2K_BENCH_LAUNCHED="1")
        if ndev and ndev < args.gpus:
            e["S2K_BENCH_DEVICE"] = str(rank % ndev)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e, stderr=subprocess.PIPE)
        threading.Thread(target=pump, args=(rank, p.stderr), daemon=True).start()
        procs.append(p)

    def end_all(but=None):
        for r, p in enumerate(procs):
            if r != but and p.poll() is None:
                p.terminate()
        t_end = time.time() + 5.0
        for r, p in enumerate(procs):
            if r != but and p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()

    t0 = time.time()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if failed:
            r, c = failed[0]
            time.sleep(0.2)                          # (its last lines are still on their way through the pipe)
            with lock:
                words = last_line.get(r, "(nothing on stderr)")
            print("bench.py: rank %d exited with code %d; its last stderr line: %s" % (r, c, words), file=sys.stderr)
            end_all(but=r)
            print("bench.py: ended the other %d rank(s)" % (len(procs) - 1), file=sys.stderr)
            rc = c if c > 0 else 1
            break
        if all(c == 0 for c in codes):
            break
        with lock:
            n_ready = len(ready)
            missing = [r for r in range(args.gpus) if r not in ready]
        if n_ready < args.gpus and time.time() - t0 > args.rank_timeout:
            with lock:
                words = {r: last_line.get(r, "(nothing on stderr)") for r in missing}
            print("bench.py: after %.0f s rank(s) %s have not reported ready; last stderr lines: %s" % (args.rank_timeout, missing, words), file=sys.stderr)
            end_all()
            rc = 124
            break
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------
def cpu_baseline(pub, digest, r, s, budget_s=15.0):
    """Time the CPU oracle (port of the reference algorithm) on a bounded prefix, with the
    thread count (<= host cores) that gives the best rate on a short probe."""
    import oracle
    oracle.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def rate(m, th):
        t0 = time.perf_counter()
        out = oracle.ecdsa_verify_batch(pub[:m], digest[:m], r[:m], s[:m], nthreads=th)
        dt = time.perf_counter() - t0
        assert out.all()
        return m / dt

    m1 = min(1024, r.shape[0])
    single = rate(m1, 1)
    probes = {1: single}           # (probes of at least 8192 signatures: a cgroup's CPU quota lets short bursts run faster than it sustains)
    th = 2
    while th <= cores:
        probes[th] = rate(min(r.shape[0], max(8192, 1024 * th)), th)
        th *= 2
    if cores not in (1,) and cores & (cores - 1):
        probes[cores] = rate(min(r.shape[0], max(8192, 1024 * cores)), cores)
    # the SMALLEST thread count within 3 % of the best probe: on a box whose cgroup gives the process 16 CPUs, 64 threads are
    # no faster than 16, and the figure to quote as `cores` is 16
    top = max(probes.values())
    best_th = min(t for t, v in probes.items() if v >= 0.97 * top)
    best = probes[best_th]
    m = int(min(r.shape[0], max(1024, best * budget_s)))
    value = rate(m, best_th)

    # BASELINE config 1: 1024 DER-encoded signatures through the single-signature entry point
    # (secec.PublicKey.Verify, parse cost included), one thread
    def der_int(b):
        b = bytes(b).lstrip(b"\0") or b"\0"
        if b[0] & 0x80:
            b = b"\0" + b
        return b"\x02" + bytes([len(b)]) + b

    t0 = time.perf_counter()
    for i in range(m1):
        body = der_int(r[i]) + der_int(s[i])
        ok = oracle.ecdsa_verify_asn1(b"\x04" + bytes(pub[i]), bytes(digest[i]), b"\x30" + bytes([len(body)]) + body)
        assert ok == 1
    config1 = m1 / (time.perf_counter() - t0)
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / 100000.0)):
        try:
            with open(path) as f:
                quota = parse(f.read().strip())
            break
        except Exception:
            continue
    # `cores`: the CPUs the sample could actually run on - the worker threads, but no more than the cgroup's CPU quota or the
    # affinity mask gives the process (round 4 printed 64 for 64 threads on a 16-CPU quota)
    cpus = min([best_th, cores] + ([max(1, int(quota + 0.5))] if quota else []))
    return {"value": value, "unit": "verifications/s", "cores": cpus, "threads": best_th, "kind": "port",
            "single_thread_value": single, "speedup_vs_1_thread": value / single,
            "host_logical_cpus": os.cpu_count(), "cpus_in_affinity_mask": cores, "cgroup_cpu_quota": quota,
            "best_probe_value": top,
            "threads_note": "`threads` = worker threads the oracle ran on (static split of the sample); `cores` = the CPUs they had: "
                            "min(threads, affinity mask, cgroup CPU quota); the speed-up over one thread is what the box gave them",
            "config1_der_single_thread": {"value": config1, "unit": "verifications/s", "sample": f"{m1} DER signatures, "
                                          "single-signature verify incl. parsing"},
            "thread_probe": {str(t): round(v) for t, v in sorted(probes.items())},
            "sample": f"first {m} signatures of the rank-0 batch, {best_th} threads (the smallest count within 3 % of the best "
                      f"of 1..{cores}, static split); single-thread rate {single:.0f}/s",
            "reference_toolchain": "go: " + ("present" if shutil.which("go") else "absent - reference Go path not timed")}


def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def committed_counts():
    """PMC-derived per-signature figures of the path's kernels (profiles/, newest round first)."""
    for name in ("r06_valu_counts.json", "r05_valu_counts.json", "r04_valu_counts.json", "r03_valu_counts.json", "r02_valu_counts.json", "r01_valu_counts.json"):
        d = load_profile_json(name)
        if d:
            return d, name
    return None, None


def recount_shipped_binary(lib_path, counts):
    """Static trip-weighted VALU counts of the two ladder kernels from the code object of the library that is LOADED
    (tools/isa_count.py: disassembly, loops from backward branches) against the committed figures: `counts_stale` says the
    kernels changed since the counters were read."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_count
        live = isa_count.static_counts(lib_path)
    except Exception as e:                      # no llvm-objdump on the box: say so, do not guess
        return {"counts_stale": None, "recount_error": repr(e)[:200]}
    out = {"counts_head": (counts or {}).get("head"), "static_recount": {k: v["valu_instr_static"] for k, v in live.items()}}
    stale = False
    for kname, skey in (("k_verify_fast", "static"), ("k_verify_fast_keyed", "static_keyed")):
        ref = (counts or {}).get(skey, {}).get("valu_instr_static")
        pmc = (counts or {}).get(kname, {}).get("valu_instr_per_signature")
        got = live[kname]["valu_instr_static"]
        for want in (ref, pmc):
            if want is None or abs(got - want) > 0.005 * want:
                stale = True
    out["counts_stale"] = stale
    return out


def committed_traffic(kernel="k_verify_fast"):
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        d = load_profile_json(name)
        if d and kernel in d:
            return d[kernel]["hbm_bytes_per_launch"], name
    return None, None


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if xs else 0.0



# ---------------------------------------------------------------------------------------------
# The printed line.  The driver keeps an 8 KB tail of stdout: everything this file measures has to fit in it (VERDICT r04
# next #7).  The explanations that used to ride along in the line ("note", "*_def", "check", "method", "sample" ...: 5.6 KB of
# the 18.5 KB of round 4) live in bench_notes.json next to this file, keyed by the same paths (--write-notes refreshes it
# from a run; --full prints the line as it was); numbers are rounded to six significant digits; per-repetition lists are
# dropped; and if the line is still too long, keys are dropped from the least to the most important and named in `dropped`.
# ---------------------------------------------------------------------------------------------
PROSE_KEYS = ("note", "check", "method", "sample", "threads_note", "peak_def", "frac_def", "reference_toolchain", "mode_def",
              "counts_note", "def")
LIST_KEYS = ("ms_each", "ms_per_batch_each", "thread_probe", "per_call_ms", "ms_rounds", "ms_rounds_every_call", "ms_rounds_off", "static_recount")
NESTED_ROOFLINE_DROPS = ("bound", "unit", "peak", "hbm", "counts_head")     # constants the top-level roofline states once
LAST_KEYS = ("distinct_keys", "general_path_same_batch", "keyset_resident", "pcie_inclusive", "batch_sweep", "small_call", "roofline", "cpu_baseline")
DROP_ORDER = ("worst_case_equal_points", "worst_case_ladder_collision", "forced_worklist", "worst_case_all_fallback", "resident_two_contexts",
              "keyset_resident_chunk_tables", "keyset_resident_joint_tables_4bit", "msm_2p22", "encoded_2p20", "key_grouping")
NESTED_DROP_ORDER = (("cpu_baseline", "best_probe_value"), ("cpu_baseline", "host_logical_cpus"), ("cpu_baseline", "cpus_in_affinity_mask"),
                     ("cpu_baseline", "speedup_vs_1_thread"), ("roofline", "shader_clock_mhz_first_wave"), ("roofline", "shader_clock_mhz_last_round"),
                     ("roofline", "kernel_ms_median"), ("roofline", "mad_u64_u32_per_s"), ("small_call", "crossovers", "lane_ms"),
                     ("schnorr_rlc_2p20", "locate_stats"), ("msm_2p20", "host_buffers"), ("schnorr_rlc_2p20", "host_buffers"))
LINE_BUDGET = 7700


def _sig6(x):
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.6g" % x)


def _strip(obj, path, notes):
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            pth = path + "." + k if path else k
            if isinstance(v, str) and (k in PROSE_KEYS or k.endswith("_def") or k.endswith("_note")):
                notes[pth] = v
                continue
            if k in LIST_KEYS:
                notes.setdefault("_dropped_lists", []).append(pth)
                continue
            if isinstance(v, str) and len(v) > 72 and path and not path.startswith("config"):   # any other long text
                notes[pth] = v
                continue
            if path and k == "unit" and v == "verifications/s":      # (the top level says it)
                continue
            if path and path.endswith("roofline") is False and k == "roofline" and isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk not in NESTED_ROOFLINE_DROPS}
            out[k] = _strip(v, pth, notes)
        return out
    if isinstance(obj, list):
        return [_strip(v, path, notes) for v in obj]
    return _sig6(obj)


def compact_line(line, args):
    if getattr(args, "full", False):
        return json.dumps(line)
    notes = {}
    slim = _strip(line, "", notes)
    head = [k for k in slim if k not in LAST_KEYS]
    ordered = {k: slim[k] for k in head}
    for k in LAST_KEYS:                      # what the record must not lose comes LAST: a tail cut takes the front
        if k in slim:
            ordered[k] = slim[k]
    dropped = []
    text = json.dumps(ordered, separators=(",", ":"))
    for k in DROP_ORDER:
        if len(text) <= LINE_BUDGET:
            break
        if k in ordered:
            dropped.append(k)
            del ordered[k]
            ordered["dropped"] = dropped
            text = json.dumps(ordered, separators=(",", ":"))
    # the contract's cpu_baseline carries `unit` and `sample`: both stay, the sample in its short form (the long one: the notes)
    if isinstance(ordered.get("cpu_baseline"), dict) and isinstance(line.get("cpu_baseline"), dict) and "value" in line["cpu_baseline"]:
        cb = dict(ordered["cpu_baseline"])
        cb["unit"] = "verifications/s"
        cb["sample"] = (line["cpu_baseline"].get("sample") or "").split(" (")[0][:72]
        ordered["cpu_baseline"] = cb
    text = json.dumps(ordered, separators=(",", ":"))
    for path in NESTED_DROP_ORDER:          # numbers of the least use, one by one, while the line is over the budget
        if len(text) <= LINE_BUDGET:
            break
        node = ordered
        for k in path[:-1]:
            node = node.get(k) if isinstance(node, dict) else None
        if isinstance(node, dict) and path[-1] in node:
            del node[path[-1]]
            dropped.append(".".join(path))
            ordered["dropped"] = dropped
            text = json.dumps(ordered, separators=(",", ":"))
    ordered["notes"] = "bench_notes.json"
    text = json.dumps(ordered, separators=(",", ":"))
    if getattr(args, "write_notes", False):
        with open(os.path.join(ROOT, "bench_notes.json"), "w") as f:
            json.dump({"what": "the explanations of bench.py's line, keyed by the path of the object they describe; written by "
                               "`python bench.py --write-notes` (the line itself carries numbers only, so that it fits the 8 KB the driver keeps)",
                       "notes": notes}, f, indent=1, sort_keys=True)
            f.write("\n")
    return text


def batch_sweep(eng, pub, digest, r, s, cpu):
    """The reference caller's own shape on the GPU (BASELINE config 1 is a 1024-signature secec.Verify loop, ecdsa.go:171): one
    synchronous s2k_ecdsa_verify_batch call from page-locked host memory to host verdicts, by batch size - what a shim that
    collects signatures into batches pays per call, and from which size on a call beats the host's CPUs (the timed
    cpu_baseline of this run)."""
    import numpy as np
    from secp256k1_voi_amd import pinned_array
    import numpy as np
    n_base = pub.shape[0]
    rows = {}
    cpu_rate = (cpu or {}).get("value")
    cpu_1 = (cpu or {}).get("single_thread_value")
    cross_all, cross_one = None, None
    for lg in range(6, 23):
        n = 1 << lg
        arrs = [pinned_array((n, a.shape[1])) for a in (pub, digest, r, s)]
        for d_, src in zip(arrs, (pub, digest, r, s)):
            for o in range(0, n, n_base):                  # (sizes above the batch of the headline: the batch repeated)
                d_[o:o + n_base] = src[:min(n_base, n - o)]
        eng.ecdsa_verify_batch(*arrs)
        reps = 9 if lg <= 16 else 5
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            v = eng.ecdsa_verify_batch(*arrs)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert int(v.sum()) == n, "batch sweep: a batch of valid signatures did not verify"
        ms = median(ts)
        rate = n / (ms * 1e-3)
        rows[lg] = ms
        if cpu_rate and cross_all is None and rate > cpu_rate:
            cross_all = n
        if cpu_1 and cross_one is None and rate > cpu_1:
            cross_one = n
        del arrs
    # the other two per-signature entry points at the reference's own size (1024 per call: BASELINE config 1's shape)
    small = {}
    try:
        from secp256k1_voi_amd.synth import synth_schnorr_batch
        m = 1024
        pk, msgs, sig = synth_schnorr_batch(eng, m, m, 77)
        rid = np.zeros(m, np.uint8)
        for key, call in (("schnorr_1024_ms", lambda: eng.schnorr_verify_batch(pk, msgs, sig)),
                          ("recover_1024_ms", lambda: eng.ecdsa_recover_batch(digest[:m], r[:m], s[:m], rid)[1])):
            ts = []
            for _ in range(12):
                t0 = time.perf_counter()
                res = call()
                ts.append((time.perf_counter() - t0) * 1e3)
            assert key != "schnorr_1024_ms" or bool(np.asarray(res).all()), "BIP-340 call of 1024 signatures rejected a valid signature"
            small[key] = median(ts[3:])
    except AssertionError:
        raise
    except Exception as e:   # noqa: BLE001 - a side measurement
        small["small_calls_error"] = "%s: %s" % (type(e).__name__, e)
    return {"log2_n": sorted(rows), "ms": [rows[k] for k in sorted(rows)], "n_from_which_a_call_beats_all_host_threads": cross_all,
            "n_from_which_a_call_beats_one_host_thread": cross_one, **small,
            "note": "ms[i] = one synchronous s2k_ecdsa_verify_batch call of 2^log2_n[i] signatures, page-locked host memory to host verdicts, median of 5-9 calls per size (signatures/s = 2^log2_n / ms); "
                    "up to 3072 signatures the wave-per-signature ladders run (k_verify_row / k_schnorr_row / k_recover_row, DESIGN 4d) and the bytes "
                    "move without DMA transfers (a page-locked block the kernels read and write in place), up to 2^15 four lanes per signature (k_verify_quad), above it the kernels of the headline; "
                    "schnorr_1024_ms / recover_1024_ms: one s2k_schnorr_verify_batch / s2k_ecdsa_recover_batch call of 1024 "
                    "items from pageable host arrays; the streaming entry points (pcie_inclusive.pipelined) hide launch and transfer latencies from 2^17 per batch on"}


def small_call(eng, S, torch, d_pub, d_dig, d_r, d_s, sweep):
    """The small-call ladders in the record: a 1024-signature call's times, the row kernel against the issue
    roofline (wave instructions per item from the committed PMC pass, kernel time from HIP events of THIS run), and the
    crossovers row -> quad -> lane measured on THIS box (device-resident calls, each ladder forced in turn) beside the defaults
    the library ships (engine_internal.h: S2K_ROW_MAX_DEFAULT / S2K_QUAD_MAX_DEFAULT)."""
    row_def, quad_def = 3072, 32768
    cfg = eng._lib.s2k_build_config().decode()
    for tok in cfg.split():
        if tok.startswith("ROW_MAX="):
            row_def = int(tok.split("=")[1])
        if tok.startswith("QUAD_MAX="):
            quad_def = int(tok.split("=")[1])
    st = torch.cuda.current_stream().cuda_stream
    d_valid = torch.zeros(1 << 17, dtype=torch.uint8, device=d_pub.device)

    def timed(n, reps=9):
        ts = []
        for _ in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.ecdsa_verify_batch_device(n, d_pub.data_ptr(), d_dig.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_valid.data_ptr(), 0, st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        assert int(d_valid[:n].sum().item()) == n, "small call: a batch of valid signatures did not verify"
        return median(ts[2:])

    sizes = [1024, 2048, 3072, 4096, 8192, 16384, 32768, 49152, 65536, 131072]
    ms = {"row": [], "quad": [], "lane": []}
    try:
        for n in sizes:
            for kind, (rmax, qmax) in (("row", (1 << 30, 0)), ("quad", (0, 1 << 30)), ("lane", (0, 0))):
                if kind == "row" and n > 16384:
                    ms[kind].append(None)              # (a wave per signature beyond this: milliseconds, and not in question)
                    continue
                eng.set_small_batch_max(rmax)
                eng.set_mid_batch_max(qmax)
                ms[kind].append(round(timed(n, 5), 3))       # (three decimals: the line has 8 KB)
        # one 1024-signature call through the row kernel, HIP events around the kernel
        eng.set_small_batch_max(row_def)
        eng.set_mid_batch_max(quad_def)
        eng.profile(True)
        for _ in range(20):
            eng.ecdsa_verify_batch_device(1024, d_pub.data_ptr(), d_dig.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_valid.data_ptr(), 0, st)
        pr = eng.profile_read_stages(cap=32)
        eng.profile(False)
    finally:
        eng.set_small_batch_max(row_def)
        eng.set_mid_batch_max(quad_def)
    kernel_ms = median(pr["fast_each"]) if pr["fast_each"] else None

    def crossover(a, b):                                   # smallest size from which ladder b is the faster one and stays so
        best = None
        for i in range(len(sizes) - 1, -1, -1):
            if ms[a][i] is None:
                best = sizes[i] if best is None else best
                continue
            if ms[b][i] is not None and ms[b][i] < ms[a][i]:
                best = sizes[i]
            else:
                break
        return best

    x_rq, x_ql = crossover("row", "quad"), crossover("quad", "lane")
    out = {"n": 1024, "ecdsa_ms": (sweep or {}).get("ms", [None] * 5)[4] if (sweep or {}).get("log2_n", [None] * 5)[4] == 10 else None,
           "schnorr_ms": (sweep or {}).get("schnorr_1024_ms"), "recover_ms": (sweep or {}).get("recover_1024_ms"),
           "device_resident_ecdsa_ms": ms["row"][0],
           "crossovers": {"sizes": sizes, "row_ms": ms["row"], "quad_ms": ms["quad"], "lane_ms": ms["lane"],
                          "row_to_quad_measured": x_rq, "quad_to_lane_measured": x_ql, "row_max_default": row_def, "quad_max_default": quad_def},
           "note": "ecdsa_ms / schnorr_ms / recover_ms: one synchronous host call of 1024 items (batch_sweep); crossovers: device-resident "
                   "calls with each ladder forced (s2k_ctx_set_small_batch_max / _mid_batch_max), median of 5; *_measured = the smallest size "
                   "from which the next ladder is faster on this box, beside the shipped defaults (a quad ladder takes sizes above "
                   "row_max up to quad_max)"}
    side = load_profile_json("r06_side_counts.json") or load_profile_json("r05_side_counts.json")
    wi = ((side or {}).get("k_verify_row") or {}).get("valu_wave_instr_per_item")
    if wi and kernel_ms:
        peak_wave_instr = SIMDS * PEAK_CLOCK_HZ / 4.0      # one wave64 VALU instruction per 4 cycles per SIMD
        out["roofline"] = {"kernel": "k_verify_row", "kernel_ms": kernel_ms, "wave_instr_per_item": wi,
                           "frac": wi * 1024 / (kernel_ms * 1e-3) / peak_wave_instr, "waves": 1280,
                           "frac_def": "VALU wave-instructions per signature (PMC, profiles/r05_side_counts.json: the signature's wave and its share of "
                                       "the preparation wave) x 1024 / kernel time (HIP events) / (1024 SIMDs x 2.4 GHz / 4 cycles); 1280 waves on "
                                       "1024 SIMDs: the call is one wave deep, what is not issue is the latency of one wave's ladder"}
    off = []
    for name, meas, dflt in (("row_max", x_rq, row_def), ("quad_max", x_ql, quad_def)):
        if meas and (meas > 2 * dflt or dflt > 2 * meas):
            off.append("%s: measured crossover %d against the default %d" % (name, meas, dflt))
    if off:
        out["crossovers_off_by_more_than_2x"] = off
    return out


# ---------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    import torch

    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.sharding import gather_valid_device
    from secp256k1_voi_amd.synth import synth_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}", file=sys.stderr)
        return 2
    # test hook (not used by the driver): several ranks on one GPU with gloo collectives
    shared_device = "S2K_BENCH_DEVICE" in os.environ
    if shared_device:
        local_rank = int(os.environ["S2K_BENCH_DEVICE"])
    backend = os.environ.get("S2K_DIST_BACKEND", "nccl")
    if os.environ.get("S2K_BENCH_TEST_FAIL_RANK") == str(rank) and world > 1:   # test hook (tests/test_bench_cpu.py): this rank dies before it joins the group
        print("bench.py: rank %d: S2K_BENCH_TEST_FAIL_RANK (test hook): leaving before init_process_group" % rank, file=sys.stderr, flush=True)
        return 3
    if str(rank) in os.environ.get("S2K_BENCH_TEST_HANG_RANK", "").split(",") and world > 1:   # test hook: this rank never gets ready
        print("bench.py: rank %d: S2K_BENCH_TEST_HANG_RANK (test hook): sleeping" % rank, file=sys.stderr, flush=True)
        time.sleep(3600)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # test hook: S2K_FORCE_COLLECTIVES=1 runs a single rank as an RCCL group of one, so that the step's two collectives go
    # through RCCL on a one-GPU box (secp256k1_voi_amd/sharding.py); the driver never sets it
    forced_group = world == 1 and os.environ.get("S2K_FORCE_COLLECTIVES") == "1"
    if world > 1 or forced_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced_group:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    batch_log2 = args.batch_log2 if args.batch_log2 is not None else (20 if world == 1 else 21)
    n = 1 << batch_log2
    n_keys = min(n, 1 << args.keys_log2)
    # the thread that feeds the device runs on the CPUs next to it (csrc/topology.cpp; no-op on one NUMA node)
    numa_node = int(S.load_library().s2k_device_numa_node(local_rank))
    numa_cpus = int(S.load_library().s2k_bind_thread_to_node(numa_node))
    t_c0 = time.perf_counter()
    eng = S.Engine(local_rank)                     # usable at once on narrow generator tables (s2k_ctx_create)
    ctx_create_s = time.perf_counter() - t_c0
    gt_first = eng.gt_info()
    eng.gt_wait()                                  # the timed steps run on the tables a long-lived context has: the wide ones
    gt_wait_s = time.perf_counter() - t_c0
    gt_now = eng.gt_info()
    if world > 1:                                  # (the launcher's watchdog waits for this line from every rank)
        print("bench.py: rank %d ready (device %d, %d-bit generator tables%s, NUMA node %d, %.1f s)"
              % (rank, local_rank, gt_now["bits"], "" if gt_now["bits"] == gt_now["target_bits"] or not gt_now["target_bits"] else " of %d wanted" % gt_now["target_bits"],
                 numa_node, time.perf_counter() - t_c0), file=sys.stderr, flush=True)
    if args.key_grouping == "off":
        eng.set_key_grouping(S.KEYS_OFF)
    pub, digest, r, s = synth_batch(eng, n, n_keys, seed=0x5EC9 + rank)
    d_pub, d_dig, d_r, d_s = (torch.from_numpy(x).to(dev) for x in (pub, digest, r, s))
    d_valid = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_bitmap = torch.zeros(n // 8, dtype=torch.uint8, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gather_scratch = {}

    main_keyset = None
    if args.key_grouping in KEYSET_OPTIONS:
        ks_keys, ks_inv = np.unique(pub, axis=0, return_inverse=True)
        main_keyset = (eng.keyset_create(ks_keys, KEYSET_OPTIONS[args.key_grouping][0]),
                       torch.from_numpy(ks_inv.reshape(-1).astype(np.uint32).view(np.int32)).to(dev))

    def step(inputs=None):
        p, d, rr, ss = inputs or (d_pub, d_dig, d_r, d_s)
        # verdicts are cleared first, so a step that silently did nothing cannot pass the count check
        d_valid.zero_()
        if main_keyset is not None:
            eng.ecdsa_verify_batch_keyset_device(main_keyset[0], n, main_keyset[1].data_ptr(), d.data_ptr(), rr.data_ptr(), ss.data_ptr(),
                                                 d_valid.data_ptr(), 0, st)
        else:
            eng.ecdsa_verify_batch_device(n, p.data_ptr(), d.data_ptr(), rr.data_ptr(), ss.data_ptr(),
                                          d_valid.data_ptr(), 0, st)
        # the only collective of the path: one all-gather of the valid bitmap shards, each with its rank's count behind it
        return gather_valid_device(d_valid, n * world, dist, engine=eng, bitmap=d_bitmap, count=d_count, scratch=gather_scratch)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity guards (untimed) -------------------------------------------------------------
    # (1) every synthetic signature is valid, on every rank
    bitmap, cnt = step()
    sync()
    assert int(cnt.item()) == n * world and bool((bitmap == 0xFF).all().item()), "synthetic batch did not verify"
    # (2) a seeded subset corrupted on every rank (one bit of s flipped): the gathered bitmap of the
    #     whole job must equal the expected pattern, on every rank
    def corrupt_mask(rk):
        i = np.arange(n, dtype=np.uint64)
        return ((i * np.uint64(2654435761) + np.uint64(rk * 7919 + 13)) % np.uint64(61)) == 0
    mine = torch.from_numpy(np.nonzero(corrupt_mask(rank))[0]).to(dev)
    d_s[mine, 31] ^= 1
    bitmap, cnt = step()
    sync()
    expect = np.concatenate([np.packbits(~corrupt_mask(rk), bitorder="little") for rk in range(world)])
    assert np.array_equal(bitmap.cpu().numpy(), expect), "gathered valid bitmap differs from the corrupted pattern"
    assert int(cnt.item()) == int(sum((~corrupt_mask(rk)).sum() for rk in range(world)))
    d_s[mine, 31] ^= 1

    # ---- W untimed warm-up steps, then exactly K timed steps ----------------------------------
    # (only the mandated barrier + synchronize in between: host-side work in that gap lets the
    # device clock down and the first timed steps pay for the ramp)
    cnts = torch.zeros(args.steps, dtype=torch.int64, device=dev)
    for _ in range(args.warmup):
        step()
    eng.profile(True)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        bitmap, cnt = step()
        cnts[k:k + 1].copy_(cnt)
    sync()
    dt = time.perf_counter() - t0
    prof = eng.profile_read_stages(cap=max(args.steps, 1))
    eng.profile(False)
    grouping = eng.key_grouping_stats()                  # of the last timed step
    assert prof["calls"] == min(args.steps, 1024)
    assert bool((cnts == n * world).all().item()), "a timed step lost verdicts"
    assert bitmap.numel() == n * world // 8 and bool((bitmap == 0xFF).all().item())

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    multi = None
    if dist is not None:
        # what a first run on N devices needs to tell a straggler from a slow collective: every rank's own time for the K
        # timed steps, and (a short untimed pass behind them) its step split at the collective by events on the stream:
        # local = verification + bitmap packing, collective = the all-gather of the bitmap shards
        own = torch.tensor([dt], dtype=torch.float64, device=t.device)
        every = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(every, own)
        k2 = max(1, min(args.steps, 10))
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(k2)]
        sync()
        for e in evs:
            e[0].record()
            d_valid.zero_()
            eng.ecdsa_verify_batch_device(n, d_pub.data_ptr(), d_dig.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_valid.data_ptr(), 0, st)
            gather_valid_device(d_valid, n * world, dist, engine=eng, bitmap=d_bitmap, count=d_count, scratch=gather_scratch, events=(e[1], e[2]))
            e[3].record()
        sync()
        split = torch.tensor([sum(e[0].elapsed_time(e[1]) for e in evs) / k2, sum(e[1].elapsed_time(e[2]) for e in evs) / k2,
                              sum(e[0].elapsed_time(e[3]) for e in evs) / k2], dtype=torch.float64, device=t.device)
        splits = torch.zeros(3 * world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(splits, split)
        splits = splits.view(world, 3).cpu().tolist()
        nodes = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(nodes, torch.tensor([float(numa_node)], dtype=torch.float64, device=t.device))
        gtb = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(gtb, torch.tensor([float(gt_now["bits"])], dtype=torch.float64, device=t.device))
        multi = {"per_rank_ms": [x * 1e3 / args.steps for x in every.cpu().tolist()],
                 "per_rank_numa_node": [int(x) for x in nodes.cpu().tolist()],
                 "per_rank_gt_bits": [int(x) for x in gtb.cpu().tolist()],
                 "per_rank_local_ms": [x[0] for x in splits], "collective_ms": [x[1] for x in splits],
                 "per_rank_step_ms_diagnostic_pass": [x[2] for x in splits],
                 "value_without_collective": n * world / (max(x[0] for x in splits) * 1e-3),
                 "per_gpu_value_without_collective": n / (max(x[0] for x in splits) * 1e-3),
                 "group_world_size": dist.get_world_size(), "backend": dist.get_backend(),
                 "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None),
                 "note": "per_rank_ms: each rank's own wall time per step over the K timed steps (value uses the maximum); "
                         "per_rank_local_ms / collective_ms: HIP events on the rank's stream around verification + packing and around "
                         "the all-gather, %d untimed steps behind the timed region; value_without_collective: all ranks' signatures "
                         "over the slowest rank's local time - what N independent single-GPU runs would give in this launch" % k2}
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    rc = 0
    if rank == 0:
        fast_ms = prof["fast_ms"] / prof["calls"]         # dominant kernel, HIP events on its stream
        fast_med = median(prof["fast_each"])
        calls = prof["calls"]
        keyed = grouping["keyed"] * 2 >= n                # which ladder is the dominant kernel of this run
        if main_keyset is not None:
            keyed = True
        clock_hz = prof["shader_mhz"] * 1e6
        value = n * world * args.steps / dt
        counts, counts_src = committed_counts()
        kname = KEYSET_OPTIONS[args.key_grouping][1] if main_keyset is not None else ("k_verify_fast_keyed" if keyed else "k_verify_fast")
        traffic, traffic_src = committed_traffic(kname)
        stages = {"grouping_by_key_ms": prof["group_ms"] / calls, "key_tables_ms": None, "ladder_ms": fast_ms,
                  "general_ladder_ms": prof["left_ms"] / calls, "complete_worklist_ms": prof["fallback_ms"] / calls}
        if keyed:
            # stage [0] is the grouping, stage [1] the per-key tables; the scalar preparation and the generator
            # part u1*G run beside both on a second stream (secp256k1_voi_amd.h: s2k_ctx_profile_read_stages)
            stages["grouping_by_key_ms"], stages["key_tables_ms"] = prof["prep_ms"] / calls, prof["group_ms"] / calls
        else:
            stages["scalar_prep_ms"] = prof["prep_ms"] / calls
            stages.pop("grouping_by_key_ms")
            stages.pop("key_tables_ms")
        roof = {"bound": "valu", "kernel": KEYSET_OPTIONS[args.key_grouping][2] if main_keyset is not None else ("k_verify_fast<ECDSA_KEYED>" if keyed else "k_verify_fast<ECDSA>"),
                "kernel_ms": fast_ms, "kernel_ms_median": fast_med, "stages_ms": stages,
                "shader_clock_mhz": prof["shader_mhz"], "shader_clock_mhz_first_wave": prof["shader_mhz_first_wave"],
                "shader_clock_mhz_last_round": prof["shader_mhz_last_round"], "unit": "Tlane-op/s", "peak": VALU_PEAK_LANE_OPS / 1e12,
                "peak_def": "256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz: one wave64 VALU instruction per 4 cycles per SIMD"}
        counts_all = counts
        if counts and kname not in counts:
            # a ladder the committed profile has no PMC entry for (yet): priced with the static recount of the loaded library
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import isa_count
                sc_ = isa_count.static_counts(S.LIB_PATH)[kname]
                counts = {kname: {"valu_instr_per_signature": sc_["valu_instr_static"]},
                          (KEYSET_OPTIONS[args.key_grouping][3] if main_keyset is not None else "static"): sc_}
                counts_src = "(none: static recount of the loaded library, tools/isa_count.py)"
            except Exception:
                counts = None
        if counts:
            ipv = counts[kname]["valu_instr_per_signature"]
            st_ = counts.get(KEYSET_OPTIONS[args.key_grouping][3] if main_keyset is not None else ("static_keyed" if keyed else "static"), {})
            lane_ops = ipv * n / (fast_ms * 1e-3)
            roof.update({"achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS,
                         "valu_instr_per_verify": ipv, "counts_from": "profiles/" + counts_src,
                         "frac_def": "VALU instructions per verification (PMC) x verifications/s of the kernel / peak; every VALU "
                                     "instruction of this kernel is priced at the 4-cycle issue interval measured for its mix "
                                     "(profiles/r02_valu_instruction_rates.txt)"})
            if clock_hz > 0:
                # the same fraction against the clock the kernel actually ran at: SIMD issue cycles used / available
                wave_instr = ipv * n / 64.0
                cyc_avail = SIMDS * fast_ms * 1e-3 * clock_hz
                roof["frac_at_measured_clock"] = wave_instr * 4.0 / cyc_avail
                if "mad_u64_u32_per_verify" in st_:
                    roof["mad_issue_frac_at_measured_clock"] = st_["mad_u64_u32_per_verify"] * n / 64.0 * 4.0 / cyc_avail
                if "valu_instr_whole_step_per_verify" in st_:
                    # every kernel of the step (grouping, tables, preparation, generator part, ladder) against the
                    # issue slots of the whole step
                    roof["whole_step_issue_frac_at_measured_clock"] = (st_["valu_instr_whole_step_per_verify"] * n / 64.0 * 4.0 /
                                                                       (SIMDS * dt / args.steps * clock_hz))
                    roof["valu_instr_whole_step_per_verify"] = st_["valu_instr_whole_step_per_verify"]
            if "fp_products_per_verify" in st_:
                roof["fp_products_per_verify"] = st_["fp_products_per_verify"]
            if "mad_u64_u32_per_verify" in st_:
                roof.update({"mad_u64_u32_per_verify": st_["mad_u64_u32_per_verify"],
                             "mad_u64_u32_per_s": st_["mad_u64_u32_per_verify"] * n / (fast_ms * 1e-3)})
            # the guide's nominal VALU rate (one wave64 instruction per 2 cycles per SIMD) is reached by pure
            # VOP2 / f32 streams only; against it the kernel sits at half the figure above
            roof["frac_of_2cycle_nominal_peak"] = lane_ops / (2 * VALU_PEAK_LANE_OPS)
        achieved_gbs = BYTES_PER_VERIFY * n / (fast_ms * 1e-3) / 1e9
        roof["traffic"] = traffic
        roof["hbm"] = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved_gbs / HBM_PEAK_GBS, "bytes_per_verify": BYTES_PER_VERIFY,
                       "traffic_from": ("profiles/" + traffic_src) if traffic_src else None,
                       "note": "algorithmic bytes / kernel time; the path is VALU-bound, this fraction is ~1e-3 by construction"}
        line = {
            "metric": "secp256k1 ECDSA verifications/sec at batch=2^%d per GPU" % batch_log2,
            "value": value, "unit": "verifications/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "2^%d ECDSA verifies (u1*G+u2*P per lane) per GPU%s, %d distinct keys per GPU, all valid, low-s"
                                   % (batch_log2, " = 2^%d in total (BASELINE config 5 shape)" % (batch_log2 + world.bit_length() - 1)
                                      if world > 1 else " (BASELINE config 2)", n_keys),
                       "parallelism": "shard%d%s%s" % (world, " (ranks share devices, gloo: test hook)" if shared_device else "",
                                                         " (RCCL group of one: test hook)" if forced_group else ""),
                       "inputs": "resident in HBM", "collective": "one all-gather per step: bitmap shard + valid count of every rank",
                       "build": eng._lib.s2k_build_config().decode(),
                       "gt_bits": gt_now["bits"], "gt_bits_at_first_call": gt_first["bits"], "ctx_create_s": ctx_create_s,
                       "wide_tables_ready_s": gt_wait_s, "gt_bytes": gt_now["bytes"], "numa_node": numa_node, "cpus_bound": numa_cpus},
            "roofline": roof,
            "key_grouping": {"mode": "adaptive (s2k_ctx_set_key_grouping default: looks for repeated keys until two large batches in a row have none)" if args.key_grouping == "auto" else "%s (--key-grouping %s)" % (args.key_grouping, args.key_grouping), "signatures_on_key_tables": grouping["keyed"],
                             "tables_built_per_step": grouping["tables"], "signatures_on_general_ladder": grouping["general"],
                             "note": "signatures are grouped by public key inside every step; keys with >= 4 signatures get a "
                                     "precomputed table (built inside the step) and their signatures a 12-doubling ladder"},
        }
        if multi is not None:
            line["multi_rank"] = multi
        roof.update(recount_shipped_binary(S.LIB_PATH, counts_all))
        extras = world == 1 and not args.no_extras
        if extras:
            try:
                line.update(extra_measurements(eng, dev, n, n_keys, step, sync, st, args, host_pub=pub, resident=(d_dig, d_r, d_s)))
            except AssertionError as e:      # a failed guard must be visible, and must fail the run
                line["extras_error"] = str(e) or "assertion failed"
                rc = 1
            except Exception as e:           # an infrastructure error in a side measurement: the line is printed all the same
                line["extras_error"] = "%s: %s" % (type(e).__name__, e)
                rc = 1
        # host-buffer entry point (chunked H2D overlapped with the kernels, D2H); reported, never `value`.
        # One untimed call first: it creates the context's staging buffers and streams.
        if not args.no_pcie and world == 1:
            try:
                eng.ecdsa_verify_batch(pub, digest, r, s)
                host_ms = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    hv = eng.ecdsa_verify_batch(pub, digest, r, s)
                    host_ms.append((time.perf_counter() - t1) * 1e3)
                    assert int(hv.sum()) == n
                line["pcie_inclusive"] = {"value": n / (median(host_ms) * 1e-3), "unit": "verifications/s", "ms_each": host_ms,
                                          "note": "s2k_ecdsa_verify_batch from pageable host buffers, one 2^%d batch per call, "
                                                  "median of 3 calls after one that creates the staging buffers" % batch_log2}
                # the same from page-locked buffers (s2k_host_alloc): asynchronous copies, one grouped call whose table phase
                # overlaps the transfer of the digests and signatures
                from secp256k1_voi_amd import pinned_array
                pinned = [pinned_array(a.shape) for a in (pub, digest, r, s)]
                for dst, src in zip(pinned, (pub, digest, r, s)):
                    dst[...] = src
                eng.ecdsa_verify_batch(*pinned)
                pin_ms = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    hv = eng.ecdsa_verify_batch(*pinned)
                    pin_ms.append((time.perf_counter() - t1) * 1e3)
                    assert int(hv.sum()) == n
                line["pcie_inclusive"]["pinned"] = {"value": n / (median(pin_ms) * 1e-3), "unit": "verifications/s", "ms_each": pin_ms,
                                                    "note": "the same call from s2k_host_alloc buffers; 160 MiB over PCIe take 3.0 ms on their "
                                                            "own (55 GB/s), the keys' 64 MiB of it are exposed"}
                # submit / wait: four batches in flight on the context's child contexts (two lanes), a new one submitted whenever one
                # is done - what a caller that streams batches gets (s2k_ecdsa_verify_batch_submit / s2k_wait); the verdict
                # arrays are page-locked too
                def pipelined(submit, nb, depth=4, lead=4):
                    # steady state: `lead` batches fill the pipeline (the first transfer has nothing to hide behind), the clock
                    # runs from the completion of batch `lead` to the completion of batch `lead + nb`
                    tickets, done, t_0 = [], 0, None
                    for k in range(nb + lead):
                        tickets.append(submit(k))
                        if len(tickets) >= depth:
                            assert int(tickets.pop(0).wait().sum()) == n
                            done += 1
                            if done == lead:
                                t_0 = time.perf_counter()
                    for tk in tickets:
                        assert int(tk.wait().sum()) == n
                        done += 1
                        if done == lead:
                            t_0 = time.perf_counter()
                    return (time.perf_counter() - t_0) * 1e3 / (done - lead)
                pin3 = [pinned] + [[pinned_array(a.shape) for a in pinned] for _ in range(3)]
                for q in pin3[1:]:
                    for dst, src in zip(q, pinned):
                        dst[...] = src
                outs3 = [pinned_array((n,)) for _ in range(4)]
                pipelined(lambda k: eng.ecdsa_verify_batch_submit(*pin3[k % 4], out=outs3[k % 4]), 8)
                pl_ms = [pipelined(lambda k: eng.ecdsa_verify_batch_submit(*pin3[k % 4], out=outs3[k % 4]), 12) for _ in range(3)]
                line["pcie_inclusive"]["pipelined"] = {"value": n / (median(pl_ms) * 1e-3), "unit": "verifications/s", "ms_per_batch": median(pl_ms),
                                                       "ms_per_batch_each": pl_ms, "batches": 12, "in_flight": 4,
                                                       "fraction_of_resident_value": (n / (median(pl_ms) * 1e-3)) / value,
                                                       "note": "s2k_ecdsa_verify_batch_submit / s2k_wait from page-locked buffers, batches of 2^%d, "
                                                               "four in flight (two lanes of two), host bytes to host verdicts; steady state: 4 batches "
                                                               "fill the pipeline, then 12 are timed completion to completion" % batch_log2}
                # the same through a key set (s2k_ecdsa_verify_batch_keyset_submit): the keys named by index, tables built once
                if n_keys < n and not args.no_extras:
                    from secp256k1_voi_amd import KEYSET_AUTO
                    ks_keys, ks_inv = np.unique(pub, axis=0, return_inverse=True)
                    ks = eng.keyset_create(ks_keys, KEYSET_AUTO)       # (5-bit joint tables where the device has the room)
                    kx3 = []
                    for q in pin3:
                        kx = pinned_array((n,), np.uint32)
                        kx[...] = ks_inv.reshape(-1).astype(np.uint32)
                        kx3.append(kx)
                    sub_ks = lambda k: eng.ecdsa_verify_batch_keyset_submit(ks, kx3[k % 4], pin3[k % 4][1], pin3[k % 4][2], pin3[k % 4][3], out=outs3[k % 4])
                    pipelined(sub_ks, 8)
                    pk_ms = [pipelined(sub_ks, 12) for _ in range(3)]
                    line["pcie_inclusive"]["pipelined_keyset"] = {
                        "value": n / (median(pk_ms) * 1e-3), "unit": "verifications/s", "ms_per_batch": median(pk_ms), "ms_per_batch_each": pk_ms,
                        "keys": int(len(ks)), "keyset_device_bytes": ks.device_bytes(), "keyset_layout": ks.layout(), "batches": 12, "in_flight": 4,
                        "note": "s2k_ecdsa_verify_batch_keyset_submit / s2k_wait: key indices, digests and signatures (100 bytes per "
                                "signature) from page-locked buffers to host verdicts, joint tables of the keys (s2k_keyset_create's choice: keyset_layout 3 = 5-bit digits) built once (not timed); "
                                "never `value`"}
                    ks.close()
                    del kx3
                pg3 = [(pub, digest, r, s)] + [tuple(a.copy() for a in (pub, digest, r, s)) for _ in range(3)]
                pipelined(lambda k: eng.ecdsa_verify_batch_submit(*pg3[k % 4]), 4)
                pg_ms = [pipelined(lambda k: eng.ecdsa_verify_batch_submit(*pg3[k % 4]), 12) for _ in range(3)]
                line["pcie_inclusive"]["pipelined_pageable"] = {"value": n / (median(pg_ms) * 1e-3), "unit": "verifications/s",
                                                                "ms_per_batch": median(pg_ms), "ms_per_batch_each": pg_ms,
                                                                "note": "the same from pageable memory: the runtime stages the copies and submit "
                                                                        "blocks while it does - 3.1 ms in a bare process (5.2 ms per batch, "
                                                                        "tools/boundary_probe.py), but this process has torch in it, where the "
                                                                        "staging copy waits for the kernels in flight (8.6 ms per submit: "
                                                                        "profiles/r04_pageable_submit_with_and_without_torch.txt)"}
                del pinned, pin3, outs3, pg3
                # the encoded boundary (SEC1 keys + DER signatures, what secec.PublicKey.Verify takes): bytes parsed on the device
                if not args.no_extras:
                    line["encoded_2p%d" % batch_log2] = encoded_measurement(eng, pub, digest, r, s)
                eng.wait_all()
            except AssertionError as e:     # a failed guard of a host-path measurement: visible, and fails the run
                line["pcie_inclusive_error"] = str(e) or "assertion failed"
                rc = 1
            except Exception as e:          # an infrastructure error there: the line is printed all the same
                line["pcie_inclusive_error"] = "%s: %s" % (type(e).__name__, e)
                rc = 1
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(pub, digest, r, s)
            except Exception as e:          # (the checker could not be built or run on this box: the line is printed all the same)
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
                rc = 1
        if not args.no_extras and world == 1:
            try:
                line["batch_sweep"] = batch_sweep(eng, pub, digest, r, s, line.get("cpu_baseline"))
            except AssertionError as e:
                line["batch_sweep"] = {"error": str(e) or "assertion failed"}
                rc = 1
            except Exception as e:
                line["batch_sweep"] = {"error": "%s: %s" % (type(e).__name__, e)}
                rc = 1
        if not args.no_extras and world == 1:
            try:
                line["small_call"] = small_call(eng, S, torch, d_pub, d_dig, d_r, d_s, line.get("batch_sweep"))
                if line["small_call"].get("crossovers_off_by_more_than_2x"):
                    line["config"]["small_call_thresholds"] = line["small_call"]["crossovers_off_by_more_than_2x"]
            except AssertionError as e:
                line["small_call"] = {"error": str(e) or "assertion failed"}
                rc = 1
            except Exception as e:
                line["small_call"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(compact_line(line, args), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def encoded_measurement(eng, pub, digest, r, s, reps=3):
    """s2k_ecdsa_verify_encoded_batch on the batch re-encoded as uncompressed SEC1 keys and DER signatures (host bytes to
    verdicts; the encoding itself is outside the timed region)."""
    import numpy as np
    from secp256k1_voi_amd import _concat
    n = r.shape[0]

    def der_int(b):
        b = bytes(b).lstrip(b"\0") or b"\0"
        if b[0] & 0x80:
            b = b"\0" + b
        return b"\x02" + bytes([len(b)]) + b
    sigs, pubs = [], []
    for i in range(n):
        body = der_int(r[i]) + der_int(s[i])
        sigs.append(b"\x30" + bytes([len(body)]) + body)
        pubs.append(b"\x04" + bytes(pub[i]))
    pb, po = _concat(pubs)
    db, do = _concat([bytes(d) for d in digest])
    sb, so = _concat(sigs)
    out = np.zeros(n, dtype=np.uint8)
    ms = []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        eng._check(eng._lib.s2k_ecdsa_verify_encoded_batch(eng._h, n, pb.ctypes.data, po.ctypes.data, db.ctypes.data, do.ctypes.data,
                                                           sb.ctypes.data, so.ctypes.data, 0, 32, 0, out.ctypes.data))
        if rep:
            ms.append((time.perf_counter() - t0) * 1e3)
        assert int(out.sum()) == n
    # (page-locked copies of the blobs: from pageable memory a submit blocks while the runtime stages the copy - 3.1 ms in a
    # bare process, but 8.6 ms in one that has torch in it, where the staging copy waits for the kernels in flight:
    # profiles/r04_pageable_submit_with_and_without_torch.txt)
    from secp256k1_voi_amd import pinned_array
    pblobs = []
    for blob, offs in ((pb, po), (db, do), (sb, so)):
        qb, qo = pinned_array(blob.shape), pinned_array(offs.shape, np.uint64)
        qb[...] = blob
        qo[...] = offs
        pblobs.append((qb, qo))

    def pipelined(nb, depth=4, lead=4):
        tickets, done, t_0 = [], 0, None
        for k in range(nb + lead):
            tickets.append(eng.ecdsa_verify_encoded_batch_submit(*pblobs, digest_len=32))
            if len(tickets) >= depth:
                assert int(tickets.pop(0).wait().sum()) == n
                done += 1
                if done == lead:
                    t_0 = time.perf_counter()
        for tk in tickets:
            assert int(tk.wait().sum()) == n
            done += 1
            if done == lead:
                t_0 = time.perf_counter()
        return (time.perf_counter() - t_0) * 1e3 / (done - lead)
    pipelined(4)
    pl = [pipelined(12) for _ in range(3)]
    return {"value": n / (median(ms) * 1e-3), "unit": "verifications/s", "ms_each": ms, "bytes_per_item": (len(pb) + len(db) + len(sb)) / n,
            "pipelined": {"value": n / (median(pl) * 1e-3), "unit": "verifications/s", "ms_per_batch": median(pl), "ms_per_batch_each": pl,
                          "note": "s2k_ecdsa_verify_encoded_batch_submit / s2k_wait, four in flight, page-locked host memory; steady state "
                                  "(4 batches fill the pipeline, 12 timed completion to completion)"},
            "note": "s2k_ecdsa_verify_encoded_batch: 65-byte SEC1 keys, 32-byte digests, DER signatures from pageable host memory; "
                    "strict DER parsing and key decoding on the device, then the batch verifier"}


def general_roofline(kernel_ms, shader_mhz, n):
    """`roofline` of the general ladder k_verify_fast<ECDSA> (what a batch without key reuse runs on): PMC count x signatures
    / live kernel time against the 4-cycle issue peak."""
    counts, src = committed_counts()
    roof = {"bound": "valu", "kernel": "k_verify_fast<ECDSA>", "kernel_ms": kernel_ms, "shader_clock_mhz": shader_mhz,
            "unit": "Tlane-op/s", "peak": VALU_PEAK_LANE_OPS / 1e12}
    if counts and "k_verify_fast" in counts:
        ipv = counts["k_verify_fast"]["valu_instr_per_signature"]
        lane_ops = ipv * n / (kernel_ms * 1e-3)
        roof.update({"achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS, "valu_instr_per_verify": ipv,
                     "counts_from": "profiles/" + src})
        if shader_mhz > 0:
            roof["frac_at_measured_clock"] = ipv * n / 64.0 * 4.0 / (SIMDS * kernel_ms * 1e-3 * shader_mhz * 1e6)
        st_ = counts.get("static", {})
        if "mad_u64_u32_per_verify" in st_:
            roof["mad_u64_u32_per_verify"] = st_["mad_u64_u32_per_verify"]
    return roof


# --key-grouping value -> (s2k_keyset_create_ex layout, name of the ladder in the committed counts, kernel, its static entry)
KEYSET_OPTIONS = {"keyset-chunks": (1, "k_verify_fast_keyset", "k_verify_fast<ECDSA_KEYSET>", "static_keyset"),
                  "keyset": (2, "k_verify_fast_keyset_joint", "k_verify_fast<ECDSA_KEYSET_JOINT>", "static_keyset_joint"),
                  "keyset5": (3, "k_verify_fast_keyset_joint5", "k_verify_fast<ECDSA_KEYSET_JOINT5>", "static_keyset_joint5"),
                  "keyset6": (4, "k_verify_fast_keyset_joint6", "k_verify_fast<ECDSA_KEYSET_JOINT6>", "static_keyset_joint6")}


def keyset_roofline(eng, kernel_ms, shader_mhz, n, option):
    """`roofline` of k_verify_fast<ECDSA_KEYSET> (64 table additions, no doubling) or <ECDSA_KEYSET_JOINT> (32): PMC count when
    the committed profile has one (bench.py --key-grouping keyset / keyset-chunks under the counters), else the static recount
    of the loaded library."""
    import secp256k1_voi_amd as S
    counts, src = committed_counts()
    kname, klabel = KEYSET_OPTIONS[option][1], KEYSET_OPTIONS[option][2]
    roof = {"bound": "valu", "kernel": klabel, "kernel_ms": kernel_ms,
            "shader_clock_mhz": shader_mhz, "unit": "Tlane-op/s", "peak": VALU_PEAK_LANE_OPS / 1e12}
    ipv = None
    if counts and kname in counts:
        ipv, roof["counts_from"] = counts[kname]["valu_instr_per_signature"], "profiles/" + src
    else:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import isa_count
            ipv = isa_count.static_counts(S.LIB_PATH)[kname]["valu_instr_static"]
            roof["counts_from"] = "static recount of the loaded library (tools/isa_count.py)"
        except Exception as e:
            roof["recount_error"] = repr(e)[:200]
    if ipv and kernel_ms > 0:
        lane_ops = ipv * n / (kernel_ms * 1e-3)
        roof.update({"achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS, "valu_instr_per_verify": ipv})
        if shader_mhz > 0:
            roof["frac_at_measured_clock"] = ipv * n / 64.0 * 4.0 / (SIMDS * kernel_ms * 1e-3 * shader_mhz * 1e6)
    return roof


def side_roofline(count_key, kernel, prof, n):
    """`roofline` object of an entry point beside the headline: its ladder's live duration (the engine's stage events,
    s2k_ctx_profile) and the ladder's VALU instructions per item from the committed counter pass (profiles/r05_side_counts.json,
    tools/collect_side_counts.sh), against the 4-cycle issue peak - as the headline's."""
    calls = max(prof["calls"], 1)
    kernel_ms = prof["fast_ms"] / calls
    roof = {"bound": "valu", "kernel": kernel, "kernel_ms": kernel_ms, "unit": "Tlane-op/s", "peak": VALU_PEAK_LANE_OPS / 1e12,
            "stages_ms": {"front_ms": (prof["prep_ms"] + prof["group_ms"]) / calls, "ladder_ms": kernel_ms,
                          "left_and_inversions_ms": prof["left_ms"] / calls, "worklist_ms": prof["fallback_ms"] / calls},
            "shader_clock_mhz": prof["shader_mhz"]}
    counts = load_profile_json("r05_side_counts.json") or {}
    c = counts.get(count_key)
    if c and kernel_ms > 0:
        ipv = c["valu_instr_per_item"]
        lane_ops = ipv * n / (kernel_ms * 1e-3)
        roof.update({"valu_instr_per_item": ipv, "achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS,
                     "counts_from": "profiles/r05_side_counts.json"})
        if prof["shader_mhz"] > 0:
            roof["frac_at_measured_clock"] = ipv * n / 64.0 * 4.0 / (SIMDS * kernel_ms * 1e-3 * prof["shader_mhz"] * 1e6)
    return roof


def multiscalar_roofline(eng, which, call_ms, stages, units, dominant, dominant_stage):
    """`roofline` object of BASELINE config 3 / 4: the dominant kernel's VALU instructions (PMC, committed profile of the
    same entry point at the same size, tools/collect_msm_profiles.sh) over its live duration (HIP events on the call's
    stream, s2k_ctx_profile_msm) against the 4-cycle issue peak; the whole call the same way; algorithmic work per
    unit; fetched + written bytes of the dominant kernel next to the algorithmic ones."""
    prof = None
    src = None
    for name in ("r06_%s_profile_2p20.json" % which, "r05_%s_profile_2p20.json" % which, "r03_%s_profile_2p20.json" % which):
        prof = load_profile_json(name)
        if prof:
            src = name
            break
    k = stages["calls"]
    st_ms = {a: stages[a] / k for a in ("front_ms", "sort_ms", "bucket_pass_ms", "reduce_ms", "tail_ms")}
    kernel_ms = st_ms[dominant_stage]
    roof = {"bound": "valu", "kernel": dominant, "kernel_ms": kernel_ms, "stages_ms": st_ms, "unit": "Tlane-op/s",
            "peak": VALU_PEAK_LANE_OPS / 1e12, "peak_def": "256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz: one wave64 VALU instruction per 4 cycles per SIMD"}
    if prof and dominant not in prof["kernels"]:       # (a kernel that became a template since: k_msm_accumulate -> k_msm_accumulate<0>)
        alias = [k for k in prof["kernels"] if k.split("<")[0] == dominant.split("<")[0]]
        if alias:
            prof["kernels"][dominant] = prof["kernels"][alias[0]]
    if prof and dominant in prof["kernels"]:
        kk = prof["kernels"][dominant]
        wave_instr = kk["valu_wave_instr_per_call"]
        lane_ops = wave_instr * 64 / (kernel_ms * 1e-3)
        total = prof["valu_wave_instr_per_call"]
        roof.update({"achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS,
                     "valu_wave_instr_dominant_kernel": wave_instr, "valu_wave_instr_whole_call": total,
                     "whole_call_frac": total * 64 / (call_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                     "counts_from": "profiles/" + src,
                     "frac_def": "VALU wave-instructions of the kernel (PMC, committed profile of this entry point at this size) x 64 / "
                                 "its live duration / peak; whole_call_frac: all kernels of the call over the call's duration",
                     "traffic": (kk.get("fetch_bytes_per_call") or 0) + (kk.get("write_bytes_per_call") or 0)})
    roof.update(units)
    return roof


def extra_measurements(eng, dev, n, n_keys, step, sync, st, args, host_pub=None, resident=None):
    """The other BASELINE configurations and side figures on one GPU; every number is guarded by a
    check of the full result."""
    import ctypes

    import numpy as np
    import torch

    from secp256k1_voi_amd.synth import (N_ORDER, synth_all_fallback_batch, synth_batch, synth_msm_terms,
                                         synth_schnorr_batch)
    out = {}
    lib, h = eng._lib, eng._h

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    d_valid = torch.zeros(n, dtype=torch.uint8, device=dev)

    def verify_on(inputs):
        p, d, rr, ss = inputs
        eng.ecdsa_verify_batch_device(n, p.data_ptr(), d.data_ptr(), rr.data_ptr(), ss.data_ptr(), d_valid.data_ptr(), 0, st)

    # ---- key set: the same batch with the per-key tables built ONCE, outside the timed region (s2k_keyset_*).  What a
    # caller with a stable key set (a validator set, say) pays per batch; never `value`: the headline builds its tables
    # inside every step. ----
    if host_pub is not None and resident is not None and n_keys < n:
        keys, inv = np.unique(host_pub, axis=0, return_inverse=True)
        d_kidx = torch.from_numpy(inv.astype(np.uint32).view(np.int32)).to(dev)
        dd, dr, ds = resident
        import time as _time
        layouts = {"keyset5": "5-bit joint tables (S2K_KEYSET_JOINT5): 26 digit positions, one table addition each",
                   "keyset": "joint tables (S2K_KEYSET_JOINT): one table addition per 4-bit digit position, 32 per signature",
                   "keyset-chunks": "chunk tables (S2K_KEYSET_CHUNKS): 64 table additions per signature"}
        for option, key in (("keyset5", "keyset_resident"), ("keyset", "keyset_resident_joint_tables_4bit"), ("keyset-chunks", "keyset_resident_chunk_tables")):
            t_b = _time.perf_counter()
            try:
                ks = eng.keyset_create(keys, KEYSET_OPTIONS[option][0])
            except Exception as e:               # (no memory for this layout's tables: the others are measured all the same)
                out[key] = {"error": "%s: %s" % (type(e).__name__, e), "layout": layouts[option]}
                continue
            build_s = _time.perf_counter() - t_b

            def with_keyset():
                eng.ecdsa_verify_batch_keyset_device(ks, n, d_kidx.data_ptr(), dd.data_ptr(), dr.data_ptr(), ds.data_ptr(), d_valid.data_ptr(), 0, st)
            d_valid.zero_()
            with_keyset()
            eng.profile(True)
            ms = timed(with_keyset, 10)
            prk = eng.profile_read_stages(cap=16)
            eng.profile(False)
            assert int(d_valid.sum().item()) == n, "key-set verification did not accept the synthetic batch"
            out[key] = {"keys": int(len(ks)), "ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s",
                        "layout": layouts[option], "keyset_device_bytes": ks.device_bytes(), "keyset_build_s": build_s,
                        "roofline": keyset_roofline(eng, prk["fast_ms"] / max(prk["calls"], 1), prk["shader_mhz"], n, option),
                        "note": "s2k_ecdsa_verify_batch_keyset_device: tables of the %d keys built once by s2k_keyset_create_ex "
                                "(not timed: keyset_build_s, host clock around the call); per call: scalar preparation, generator part, "
                                "sort by key index, ladder" % len(ks)}
            ks.close()
        del d_kidx

    # ---- two contexts on two streams taking resident batches alternately: one batch's grouping and tables beside the other's
    # ladder.  What a device-resident caller with two batches at hand gets; `value` stays the single-stream figure ----
    if resident is not None and host_pub is not None:
        import secp256k1_voi_amd as S2
        eng_b = S2.Engine(dev.index or 0)
        streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        valid2 = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_pub2 = torch.from_numpy(host_pub).to(dev)
        dd, dr, ds = resident

        def two_ctx(reps):
            for k_ in range(reps):
                j = k_ & 1
                with torch.cuda.stream(streams2[j]):
                    (eng, eng_b)[j].ecdsa_verify_batch_device(n, d_pub2.data_ptr(), dd.data_ptr(), dr.data_ptr(), ds.data_ptr(),
                                                              valid2[j].data_ptr(), 0, streams2[j].cuda_stream)
        for s_ in streams2:
            s_.wait_stream(torch.cuda.current_stream())
        two_ctx(4)
        torch.cuda.synchronize()
        reps2 = 20
        t_0 = time.perf_counter()
        two_ctx(reps2)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t_0) * 1e3 / reps2
        assert all(int(v.sum().item()) == n for v in valid2), "two-context verification lost verdicts"
        out["resident_two_contexts"] = {"ms": ms2, "value": n / (ms2 * 1e-3), "unit": "verifications/s",
                                        "note": "two contexts on two streams, %d resident batches alternately (no bitmap exchange): the "
                                                "overlap the submit / wait lanes are built on; never `value`" % reps2}
        eng_b.close()
        del d_pub2, valid2

    # ---- the same batch with key grouping off: every signature as if its key were new (the reference's way) ----
    from secp256k1_voi_amd import KEYS_ADAPTIVE, KEYS_AUTO, KEYS_OFF
    if n_keys < n:
        eng.set_key_grouping(KEYS_OFF)
        try:
            step()
            eng.profile(True)
            d_valid.zero_()
            ms = timed(step, 5)
            pr = eng.profile_read_stages(cap=8)
            _, cnt = step()
            assert int(cnt.item()) == n, "general path lost verdicts"
        finally:
            eng.profile(False)
            eng.set_key_grouping(KEYS_ADAPTIVE)
        out["general_path_same_batch"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s",
                                          "kernel": "k_verify_fast<ECDSA>", "kernel_ms": pr["fast_ms"] / pr["calls"],
                                          "shader_clock_mhz": pr["shader_mhz"],
                                          "note": "s2k_ctx_set_key_grouping(S2K_KEYS_OFF): per-signature table and 128 doublings "
                                                  "for every signature, step includes the same bitmap exchange"}
        general_ms = pr["fast_ms"] / pr["calls"]
        out["general_path_same_batch"]["roofline"] = general_roofline(general_ms, pr["shader_mhz"], n)

    # ---- K = N: every signature under its own key (SURVEY 8d "also report K = N").  The grouping finds nothing to share
    # and must cost (next to) nothing.  Three settings measured ALTERNATELY, five rounds of five calls each, medians - the
    # clock drifts by more than the difference between two consecutive blocks of calls: the context's default
    # (S2K_KEYS_ADAPTIVE: after two observed batches without a repeated key it verifies fifteen batches without looking,
    # looks again in one, ...), S2K_KEYS_AUTO (looks in every call) and S2K_KEYS_OFF ----
    if n_keys < n:
        inp = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(eng, n, n, seed=0xD157))
        ms_adp, ms_auto, ms_off, kern = [], [], [], None
        try:
            eng.set_key_grouping(KEYS_ADAPTIVE)
            eng.key_grouping_adaptive(reset=True)
            ad0 = eng.key_grouping_adaptive()
            for _ in range(3):               # what a caller's first batches of such a workload are: looked at, and noted
                verify_on(inp)
                torch.cuda.synchronize()
            for rnd_ in range(5):
                eng.set_key_grouping(KEYS_ADAPTIVE)
                d_valid.zero_()
                ms_adp.append(timed(lambda: verify_on(inp), 5))
                assert int(d_valid.sum().item()) == n, "K = N batch did not verify (adaptive)"
                eng.set_key_grouping(KEYS_AUTO)
                d_valid.zero_()
                ms_auto.append(timed(lambda: verify_on(inp), 5))
                assert int(d_valid.sum().item()) == n, "K = N batch did not verify"
                gs = eng.key_grouping_stats()
                assert gs["keyed"] == 0 and gs["general"] == n, gs
                eng.set_key_grouping(KEYS_OFF)
                d_valid.zero_()
                if rnd_ == 4:
                    eng.profile(True)
                ms_off.append(timed(lambda: verify_on(inp), 5))
                assert int(d_valid.sum().item()) == n, "K = N batch did not verify (grouping off)"
            pr2 = eng.profile_read_stages(cap=8)
            kern = (pr2["fast_ms"] / pr2["calls"], pr2["shader_mhz"])
            ad1 = eng.key_grouping_adaptive()
        finally:
            eng.profile(False)
            eng.set_key_grouping(KEYS_ADAPTIVE)
            eng.key_grouping_adaptive(reset=True)
        ms, ms_auto_med, ms_off_med = median(ms_adp), median(ms_auto), median(ms_off)
        out["distinct_keys"] = {"keys": n, "ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s",
                                "ms_with_key_grouping_off": ms_off_med, "grouping_overhead": ms / ms_off_med - 1.0,
                                "ms_with_key_grouping_in_every_call": ms_auto_med,
                                "grouping_overhead_in_every_call": ms_auto_med / ms_off_med - 1.0,
                                "adaptive_calls": {"verified_without_looking": ad1["skipped"] - ad0["skipped"],
                                                   "looked_again": ad1["probes"] - ad0["probes"],
                                                   "of": 3 + 5 * 6},
                                "ms_rounds": ms_adp, "ms_rounds_every_call": ms_auto, "ms_rounds_off": ms_off,
                                "roofline": general_roofline(kern[0], kern[1], n),
                                "note": "every signature under its own key: the grouping finds nothing to share and all "
                                        "signatures take the general ladder.  `ms`: the context's default setting "
                                        "(S2K_KEYS_ADAPTIVE) after three such batches, the calls with the grouping in every "
                                        "call (S2K_KEYS_AUTO) and off measured alternately with it (5 x 5 calls each, medians); "
                                        "roofline: k_verify_fast<ECDSA> of the grouping-off calls"}
        del inp

    # ---- adversarial inputs.  What a key owner can force for GIVEN digests are the two exceptional cases of the final
    # addition: u1 G = -u2 Q (r = -e/d, R = infinity) and u1 G = u2 Q (r = e/d, R = 2 u1 G).  Until round 3 the first sent
    # every lane to the complete-formula worklist (3.4 x a step); now it is decided in the ladder kernel, the second by
    # a short form of the worklist kernel.  `forced_worklist`: the all-valid batch with every lane pushed through the
    # complete kernel by a diagnostic flag - what a batch of mid-ladder exceptional cases would cost if one could be
    # made (DESIGN.md section 4 argues it cannot). ----
    from secp256k1_voi_amd import FORCE_WORKLIST
    from secp256k1_voi_amd.synth import synth_equal_points_batch
    inp = tuple(torch.from_numpy(x).to(dev) for x in synth_all_fallback_batch(eng, n, n_keys, seed=0xBAD))
    d_valid.fill_(1)
    ms = timed(lambda: verify_on(inp), 3)
    assert int(d_valid.sum().item()) == 0, "R = infinity must reject"
    gs = eng.key_grouping_stats()
    out["worst_case_all_fallback"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s", "on_worklist": gs["complete"],
                                      "note": "2^%d signatures built so that u1*G + u2*Q = infinity (r = -e/d): the final addition of "
                                              "every lane is exceptional; decided in the ladder kernel since round 4 (was: every lane "
                                              "re-done by the complete-formula kernel); all verdicts 0 (checked)" % (n.bit_length() - 1)}
    del inp
    inp = tuple(torch.from_numpy(x).to(dev) for x in synth_equal_points_batch(eng, n, n_keys, seed=0xBAD2))
    d_valid.fill_(1)
    ms = timed(lambda: verify_on(inp), 3)
    assert int(d_valid.sum().item()) == 0, "2 u1 G with a random r must reject"
    gs = eng.key_grouping_stats()
    out["worst_case_equal_points"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s", "on_worklist": gs["complete"],
                                      "note": "u1*G = u2*Q in every lane (r = e/d): tagged worklist entries, R = 2 u1 G from the "
                                              "generator tables with complete formulas; all verdicts 0 (checked)"}
    del inp
    # one crafted value of u2 = r/s (no key needed: any r, s = r / u2) made the LAST table addition of the keyed ladder add a
    # point to itself with the plain odd split (every lane on the worklist, 16.6 ms); the split now takes the other lattice
    # vector for it.  Kept in the line as the regression guard.
    from secp256k1_voi_amd.synth import synth_ladder_collision_batch
    inp = tuple(torch.from_numpy(x).to(dev) for x in synth_ladder_collision_batch(eng, n, n_keys, seed=0xBAD3))
    d_valid.fill_(1)
    ms = timed(lambda: verify_on(inp), 2)
    assert int(d_valid.sum().item()) == 0, "crafted u2 with random r must reject"
    gs = eng.key_grouping_stats()
    out["worst_case_ladder_collision"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s", "on_worklist": gs["complete"],
                                          "note": "u2 = r/s = -26 * 16^28 * lambda in every lane: with the plain odd GLV split the keyed ladder's "
                                                  "last table addition was P + P and every lane went to the complete-formula kernel "
                                                  "(16.6 ms); the split avoids it since round 4; all verdicts 0 (checked)"}
    del inp
    if resident is not None:
        dd, dr, ds = resident
        d_pub_all = torch.from_numpy(host_pub).to(dev)
        d_valid.zero_()

        def forced():
            eng.ecdsa_verify_batch_device(n, d_pub_all.data_ptr(), dd.data_ptr(), dr.data_ptr(), ds.data_ptr(), d_valid.data_ptr(), FORCE_WORKLIST, st)
        ms = timed(forced, 2)
        assert int(d_valid.sum().item()) == n, "forced worklist lost verdicts"
        out["forced_worklist"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "verifications/s",
                                  "note": "S2K_ECDSA_FORCE_WORKLIST on the all-valid batch: ladder, then every lane again through "
                                          "the complete-formula kernel (diagnostic; no input is known that does this)"}
        del d_pub_all

    # ---- config 3: 2^20-term multi-scalar multiplication, points with known discrete logs ----
    m = 1 << 20
    k, pts, tot = synth_msm_terms(eng, m, seed=7)
    dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
    dout = torch.zeros(80, dtype=torch.uint8, device=dev)
    eng.multi_scalar_mult_device(m, dk.data_ptr(), dp.data_ptr(), dout.data_ptr(), st)      # (allocates the workspace)
    eng.profile_msm(True)
    ms = timed(lambda: eng.multi_scalar_mult_device(m, dk.data_ptr(), dp.data_ptr(), dout.data_ptr(), st), 5)
    stages = eng.profile_read_msm()
    want = eng.scalar_base_mult_batch([tot.to_bytes(32, "big")])[0].tobytes()
    assert dout[:65].cpu().numpy().tobytes() == want, "2^20-term MSM differs from (sum k_i d_i) G"
    out["msm_2p20"] = {"terms": m, "ms": ms, "terms_per_s": m / (ms * 1e-3),
                       "method": "Pippenger over signed 16-bit windows of the endomorphism-split scalars: two-level counting sort of the "
                                 "(window, digit) keys assembled in LDS, then a bucket pass over EQUAL RANGES OF THE SORTED LIST (one lane per "
                                 "range, incomplete XYZZ additions in registers, pieces stitched per bucket) - LDS is used by the sort, the "
                                 "bucket accumulation itself runs in registers over the sorted list (DESIGN.md 7a); BASELINE config 3 says "
                                 "'LDS bucket accumulation'",
                       "check": "full sum == (sum k_i d_i mod n) * G (big-int on the host, base mult on the device)",
                       "roofline": multiscalar_roofline(eng, "msm", ms, stages, {
                           # every input is two 128-bit terms (endomorphism), 8 signed 16-bit windows each: one bucket
                           # addition per non-zero digit (complete mixed addition: 11 field products), + one per bucket
                           # crossing a range border and the reduction's ~3 complete additions per bucket (12 products)
                           "point_adds_per_term": 2 * 8 * (1 - 2.0 ** -16) + (9 * 32768 * 4.0) / m,
                           "fp_products_per_term": 2 * 8 * 11 + (9 * 32768 * 4.0 * 12) / m + 4,
                           "algorithmic_bytes_per_term": 32 + 65,
                           "hbm": {"bound": "hbm", "achieved": (32 + 65) * m / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": (32 + 65) * m / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}, "k_msm_accumulate", "bucket_pass_ms")}
    # the fixed tail (bucket reduction, tree, 112-doubling Horner chain: 0.39 of 1.68 ms) amortises with the size of the call and
    # with a second call in flight.  (a) 2^22 terms: the same points under the scalars k, 3k, 5k, 7k mod n
    # (other digits, so no point meets itself in a bucket), expected sum 16 * (sum k_i d_i) G
    from secp256k1_voi_amd import OP_MUL
    ks4 = [k]
    for c in (3, 5, 7):
        ks4.append(eng.fn_op_batch(OP_MUL, k, np.tile(np.frombuffer(int(c).to_bytes(32, "big"), np.uint8), (m, 1)))[0])
    dk4 = torch.from_numpy(np.concatenate(ks4)).to(dev)
    dp4 = dp.repeat(4, 1)
    del ks4
    eng.multi_scalar_mult_device(4 * m, dk4.data_ptr(), dp4.data_ptr(), dout.data_ptr(), st)
    ms4 = timed(lambda: eng.multi_scalar_mult_device(4 * m, dk4.data_ptr(), dp4.data_ptr(), dout.data_ptr(), st), 3)
    want4 = eng.scalar_base_mult_batch([(16 * tot % N_ORDER).to_bytes(32, "big")])[0].tobytes()
    assert dout[:65].cpu().numpy().tobytes() == want4, "2^22-term MSM differs from 16 (sum k_i d_i) G"
    out["msm_2p22"] = {"terms": 4 * m, "ms": ms4, "terms_per_s": 4 * m / (ms4 * 1e-3),
                       "check": "full sum == 16 * (sum k_i d_i mod n) * G: the 2^20 points under k, 3k, 5k, 7k"}
    del dk4, dp4
    # (b) two calls in flight: two contexts, two host threads (the device entry point synchronises its stream to read the
    # status word), two streams, against one call after the other on one of them; both contexts warmed, three rounds, medians
    # (tools/two_calls_probe.py is the same measurement alone in a process: 1.20 against 1.36 ms per call - one call's tail
    # hides behind the other's bucket pass)
    import threading

    import secp256k1_voi_amd as S
    eng_b = S.Engine(dev.index or 0)
    streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs2 = [torch.zeros(80, dtype=torch.uint8, device=dev) for _ in range(2)]

    def msm_worker(e_, stream, o, reps):
        for _ in range(reps):
            e_.multi_scalar_mult_device(m, dk.data_ptr(), dp.data_ptr(), o.data_ptr(), stream.cuda_stream)
    for e_, s_, o in zip((eng, eng_b), streams2, outs2):               # (allocates the second workspace, warms both)
        msm_worker(e_, s_, o, 2)
    torch.cuda.synchronize()
    reps2, serial, both = 6, [], []
    for _ in range(3):
        t_0 = time.perf_counter()
        msm_worker(eng, streams2[0], outs2[0], 2 * reps2)
        torch.cuda.synchronize()
        serial.append((time.perf_counter() - t_0) * 1e3 / (2 * reps2))
        th = [threading.Thread(target=msm_worker, args=(e_, s_, o, reps2)) for e_, s_, o in zip((eng, eng_b), streams2, outs2)]
        t_0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize()
        both.append((time.perf_counter() - t_0) * 1e3 / (2 * reps2))
    assert all(o[:65].cpu().numpy().tobytes() == want for o in outs2), "MSM in two contexts differs"
    out["msm_2p20"]["two_calls_in_flight_ms"] = median(both)
    out["msm_2p20"]["one_call_after_the_other_ms"] = median(serial)
    out["msm_2p20"]["two_calls_in_flight_note"] = ("wall time PER CALL: two contexts on two host threads and streams, %d calls each, against %d calls "
                                                   "one after the other on one context (host time included in both), median of three rounds" % (reps2, 2 * reps2))
    del dk, dp

    # (c) from HOST memory: the synchronous host-pointer call pays transfer and kernels in series; two verifiers (two contexts,
    # two host threads) taking whole batches alternately hide one's transfer behind the other's kernels - the library lets the
    # phases of such calls take turns per device (msm.hip: phase_locks), so the two do not fall into lock step
    from secp256k1_voi_amd import pinned_array

    def host_two_verifiers(call_a, call_b, reps=6):
        def loop(fn):
            for _ in range(reps):
                fn()
        call_a(); call_b()
        each = []
        for _ in range(reps):                    # (median: a synchronous call now and then takes a scheduling hiccup of milliseconds)
            t_0 = time.perf_counter()
            call_a()
            each.append((time.perf_counter() - t_0) * 1e3)
        one = median(each)
        th = [threading.Thread(target=loop, args=(f,)) for f in (call_a, call_b)]
        t_0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        return one, (time.perf_counter() - t_0) * 1e3 / (2 * reps)
    hk, hp = [], []
    for _ in range(2):
        a_, b_ = pinned_array(k.shape), pinned_array(pts.shape)
        a_[...] = k
        b_[...] = pts
        hk.append(a_)
        hp.append(b_)

    def host_msm(e_, j):
        assert e_.multi_scalar_mult(hk[j], hp[j]) == want, "MSM from host memory differs"
    one_ms, two_ms = host_two_verifiers(lambda: host_msm(eng, 0), lambda: host_msm(eng_b, 1))
    out["msm_2p20"]["host_buffers"] = {"one_verifier_ms": one_ms, "two_verifiers_ms_per_call": two_ms, "terms_per_s_two_verifiers": m / (two_ms * 1e-3),
                                       "note": "s2k_multi_scalar_mult from page-locked host memory (97 bytes per term over PCIe): one verifier, "
                                               "and two (two contexts, two host threads) taking whole batches alternately; never `value`"}
    del hk, hp

    # ---- config 4: 2^20 BIP-340 signatures as one random-linear-combination MSM ----
    # ---- public-key recovery (RecoverPublicKey, secec/ecdsa.go:244-282) over the resident batch of the headline: the recovered
    # keys must be the signers' ----
    if resident is not None and host_pub is not None:
        dd, dr, ds = resident
        d_rid = torch.zeros(n, dtype=torch.uint8, device=dev)
        d_q = torch.zeros((n, 65), dtype=torch.uint8, device=dev)
        d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        rec = lambda: eng._check(lib.s2k_ecdsa_recover_batch_device(h, n, dd.data_ptr(), dr.data_ptr(), ds.data_ptr(), d_rid.data_ptr(), 0,
                                                                    d_q.data_ptr(), d_ok.data_ptr(), st))
        # the recovery id of a synthetic signature is not kept by the generator: both parities are tried, the right one
        # gives the signer's key (ids 2 and 3, x(R) >= n, do not occur in random data)
        rec()
        torch.cuda.synchronize()
        want = torch.from_numpy(host_pub).to(dev)
        hit0 = (d_q[:, 1:] == want).all(dim=1) & (d_ok == 1)
        d_rid[~hit0] = 1
        ms_rec = timed(rec, 5)
        assert bool(((d_q[:, 1:] == want).all(dim=1) & (d_ok == 1)).all().item()), "recovery did not return the signers' keys"
        eng.profile(True)
        for _ in range(5):
            rec()
        torch.cuda.synchronize()
        pr = eng.profile_read_stages(cap=8)
        eng.profile(False)
        out["recover_2p20"] = {"sigs": n, "ms": ms_rec, "sigs_per_s": n / (ms_rec * 1e-3),
                               "roofline": side_roofline("k_verify_fast_recover", "k_verify_fast<RECOVER>", pr, n),
                               "note": "s2k_ecdsa_recover_batch_device: r, s, digest and recovery id resident in HBM -> 65-byte keys; every "
                                       "recovered key compared with the signer's; never `value`"}
        del d_rid, d_q, d_ok, want
    pk, msgs, sig = synth_schnorr_batch(eng, m, min(m, 1 << 16), seed=340)
    dpk, dmsg, dsig = (torch.from_numpy(x).to(dev) for x in (pk, msgs, sig))
    res = ctypes.c_int(0)
    seed = np.frombuffer(os.urandom(32), np.uint8)

    def rlc():
        rc_ = lib.s2k_schnorr_batch_verify_rlc_device(h, m, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                      seed.ctypes.data, ctypes.byref(res), st)
        assert rc_ == 0
    rlc()
    eng.profile_read_msm()
    ms = timed(rlc, 5)
    stages = eng.profile_read_msm()
    eng.profile_msm(False)
    assert res.value == 1, "valid BIP-340 batch rejected"
    bad = int(np.random.default_rng(5).integers(0, m))
    dsig[bad, 63] ^= 1                       # one bad signature anywhere must reject the batch
    rlc()
    assert res.value == 0, "BIP-340 batch with one bad signature accepted"
    dsig[bad, 63] ^= 1
    dval = torch.zeros(m, dtype=torch.uint8, device=dev)
    single = lambda: lib.s2k_schnorr_verify_batch_device(h, m, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(), 0, dval.data_ptr(), st)
    ms_single = timed(single, 3)
    assert int(dval.sum().item()) == m
    # the same call with the engine's stage events on: its ladder as a roofline row
    eng.profile(True)
    for _ in range(5):
        single()
    torch.cuda.synchronize()
    pr = eng.profile_read_stages(cap=8)
    eng.profile(False)
    out["schnorr_per_signature_2p20"] = {
        "sigs": m, "ms": ms_single, "sigs_per_s": m / (ms_single * 1e-3), "keys": int(min(m, 1 << 16)),
        "roofline": side_roofline("k_verify_fast_schnorr_keyed", "k_verify_fast<SCHNORR_KEYED>", pr, m),
        "note": "s2k_schnorr_verify_batch_device (SchnorrPublicKey.Verify per signature, secec/bitcoin/schnorr.go:221-253): 2^20 "
                "signatures of 2^16 x-only keys resident in HBM, keys grouped and lifted once per call; never `value`"}
    # locating one bad signature by bisection on the kept terms (s2k_schnorr_verify_batch_bisect_device)
    stats = (ctypes.c_uint32 * 4)()
    dsig[bad, 63] ^= 1

    def locate():
        rc_ = lib.s2k_schnorr_verify_batch_bisect_device(h, m, dpk.data_ptr(), dmsg.data_ptr(), None, 32, dsig.data_ptr(),
                                                         seed.ctypes.data, dval.data_ptr(), stats, st)
        assert rc_ == 0
    ms_locate = timed(locate, 3)
    assert int(dval.sum().item()) == m - 1 and int(dval[bad].item()) == 0, "bisection did not single out the bad signature"
    dsig[bad, 63] ^= 1
    # per-signature verification over a key set: the batch's x-only keys lifted once and kept as joint tables
    ks_entry = None
    try:
        xs, inv = np.unique(pk, axis=0, return_inverse=True)
        pts65, okd = eng.point_decode_batch(np.concatenate([np.full((len(xs), 1), 2, np.uint8), xs], axis=1), 33)
        assert bool(okd.all())
        ks = eng.keyset_create(np.ascontiguousarray(pts65[:, 1:]))
        dkidx = torch.from_numpy(inv.reshape(-1).astype(np.uint32).view(np.int32)).to(dev)
        dval.zero_()
        ms_ks = timed(lambda: eng.schnorr_verify_batch_keyset_device(ks, m, dkidx.data_ptr(), dmsg.data_ptr(), 32, dsig.data_ptr(),
                                                                     dval.data_ptr(), st), 5)
        assert int(dval.sum().item()) == m, "BIP-340 verification over the key set did not accept the synthetic batch"
        ks_entry = {"ms": ms_ks, "sigs_per_s": m / (ms_ks * 1e-3), "keys": int(len(ks)), "keyset_layout": ks.layout(),
                    "keyset_device_bytes": ks.device_bytes(),
                    "note": "s2k_schnorr_verify_batch_keyset_device: per-signature verdicts, the keys' tables built once (not timed)"}
        ks.close()
        del dkidx
    except AssertionError:
        raise
    except Exception as e:
        ks_entry = {"error": "%s: %s" % (type(e).__name__, e)}
    # from host memory, one verifier and two (as for config 3)
    hb = []
    for _ in range(2):
        q = [pinned_array(a_.shape) for a_ in (pk, msgs, sig)]
        for dst_, src_ in zip(q, (pk, msgs, sig)):
            dst_[...] = src_
        hb.append(q)

    def host_rlc(e_, j):
        assert e_.schnorr_batch_verify_rlc(*hb[j]), "BIP-340 batch from host memory rejected"
    try:
        one_ms, two_ms = host_two_verifiers(lambda: host_rlc(eng, 0), lambda: host_rlc(eng_b, 1))
        host_entry = {"one_verifier_ms": one_ms, "two_verifiers_ms_per_call": two_ms, "sigs_per_s_two_verifiers": m / (two_ms * 1e-3),
                      "note": "s2k_schnorr_batch_verify_rlc from page-locked host memory (128 bytes per signature over PCIe): one verifier, "
                              "and two (two contexts, two host threads) taking whole batches alternately; never `value`"}
    finally:
        eng_b.close()
    del hb
    out["schnorr_rlc_2p20"] = {"sigs": m, "ms": ms, "sigs_per_s": m / (ms * 1e-3),
                               "host_buffers": host_entry,
                               "per_signature_verify_ms": ms_single,
                               "per_signature_verify_over_key_set": ks_entry,
                               "locate_one_bad_signature_ms": ms_locate,
                               "locate_stats": {"sub_combinations": int(stats[0]), "verified_one_by_one": int(stats[1]),
                                                "levels": int(stats[2])},
                               "check": "accepts the valid batch, rejects it with one flipped bit at index %d; "
                                        "per-signature verification accepts all" % bad,
                               "roofline": multiscalar_roofline(eng, "rlc", ms, stages, {
                                   # per signature: one square root (lift_x of r: 253 squarings + 13 products), the
                                   # challenge hash (2 SHA-256 compressions) and the coefficient (1), 3 scalar products; one
                                   # 128-bit term (8 bucket additions); per distinct key one more square root and two terms
                                   "fp_products_per_sig": 266 + 8 * 11 + (16 * 11 + 266) / 16.0,
                                   "point_adds_per_sig": 8 + 16 / 16.0,
                                   "algorithmic_bytes_per_sig": 32 + 32 + 64,
                                   "hbm": {"bound": "hbm", "achieved": 128 * m / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": 128 * m / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}, "k_schnorr_rlc_prep<true>", "front_ms")}
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
