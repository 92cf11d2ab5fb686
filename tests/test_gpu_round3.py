"""Round-3 GPU tests: pinned / registered / pageable host buffers give the same verdicts; the memory query; the
sharded entry points (verification, multi-scalar multiplication, BIP-340 batch) with the real engine in two ranks
sharing one device; the bench's two-rank flow at the real per-rank size; the regression check of a compiler
miscompile that the product code avoids by shape.  Needs a real MI355X.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pyref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b32 = R.b32


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    return S.Engine(0)


@pytest.mark.parametrize("n", [300, 70000, 300001])
def test_host_buffers_pinned_registered_pageable(eng, oracle, n):
    """s2k_ecdsa_verify_batch from page-locked memory (s2k_host_alloc), from registered memory (s2k_host_register) and
    from pageable memory: the pinned forms take the single grouped call whose signature data arrives piece by piece
    behind the keys; all three must give the oracle's verdicts (valid and damaged signatures, repeated and lone keys),
    with the grouping on and off."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    arrs = [np.array(a) for a in synth_batch(eng, n, max(n // 9, 3), seed=300 + n)]
    rng = np.random.default_rng(n)
    for i in range(0, n, 7):                      # every seventh item damaged somewhere (key, digest, r or s)
        a = arrs[int(rng.integers(0, 4))]
        a[i, int(rng.integers(0, a.shape[1]))] ^= 1 << int(rng.integers(0, 8))
    m = min(n, 4096)
    exp_head = oracle.ecdsa_verify_batch(*(a[:m] for a in arrs), nthreads=os.cpu_count() or 1)
    pinned = [S.pinned_array(a.shape) for a in arrs]
    for d, a in zip(pinned, arrs):
        d[...] = a
    registered = [S.page_aligned_array(a.shape) for a in arrs]   # buffers that own their pages (see s2k_host_register)
    for d, a in zip(registered, arrs):
        d[...] = a
    for a in registered:
        S.host_register(a)
    try:
        for mode in (S.KEYS_AUTO, S.KEYS_OFF):
            eng.set_key_grouping(mode)
            ref = eng.ecdsa_verify_batch(*arrs)
            assert np.array_equal(ref[:m], exp_head)
            assert 0 < int(ref.sum()) < n
            assert np.array_equal(eng.ecdsa_verify_batch(*pinned), ref)
            assert np.array_equal(eng.ecdsa_verify_batch(*registered), ref)
    finally:
        eng.set_key_grouping(S.KEYS_AUTO)
        for a in registered:
            S.host_unregister(a)


def test_device_bytes_counts_grouping_buffers(eng):
    """s2k_ctx_device_bytes: what a context holds for a batch of n - generator tables + per-signature workspace, and with
    the grouping on the grouping arrays and the table buffer (s2k_ecdsa_workspace_bytes alone understated it)."""
    import secp256k1_voi_amd as S
    n = 1 << 20
    ws = eng.workspace_bytes(n)
    eng.set_key_grouping(S.KEYS_OFF)
    off = eng.device_bytes(n)
    eng.set_key_grouping(S.KEYS_AUTO)
    auto = eng.device_bytes(n)
    assert off >= ws + (3 << 30)
    assert auto - off >= (n // 6) * 72 * 128          # the per-key table buffer alone
    assert eng.device_bytes(100) == eng.workspace_bytes(100) + off - ws   # below 256 signatures nothing is grouped


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_entry_points_two_ranks_real_engine():
    """msm_sharded, schnorr_batch_verify_sharded and the device-resident verification flow (gather_valid_device with the
    engine's bitmap packing) in TWO rank processes that share GPU 0 (gloo collectives; fresh children, started before they
    touch a GPU): the multi-scalar sum over 2^16 known-discrete-log terms equals (sum k_i d_i) G on every rank, the BIP-340
    batch is accepted, rejected by both ranks when one signature in the last rank's shard is damaged, and the gathered
    bitmap / count of the verification shards match the seeded damage pattern of BOTH ranks."""
    port = _free_port()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(r), "2", str(port), "16"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
    assert sorted(o["rank"] for o in outs) == [0, 1]
    for o in outs:
        assert o["msm_ok"] and o["schnorr_all_valid"] and not o["schnorr_one_bad"], o
        assert o["ecdsa_bitmap_ok"] and o["ecdsa_count_ok"] and o["keyed"] > 0, o
    assert outs[0]["msm_sum"] == outs[1]["msm_sum"]


def test_sharded_entry_points_through_rccl_group_of_one():
    """The same worker with backend nccl (= RCCL): a one-GPU box cannot hold two RCCL ranks (one device per rank), so the
    group has ONE member and S2K_FORCE_COLLECTIVES=1 sends it through the collectives anyway: all_gather_into_tensor of the
    uint8 bitmap and of the 65-byte partial sum, all_reduce SUM of the int64 count and MIN of the int32 verdict and the
    barrier all run in RCCL on device buffers, and the results are the single-process ones.  (What two or eight RCCL ranks
    add is transport over xGMI, which this box does not have.)"""
    port = _free_port()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["S2K_FORCE_COLLECTIVES"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), "0", "1", str(port), "14", "nccl"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    o = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert o["backend"] == "nccl", o
    assert o["msm_ok"] and o["schnorr_all_valid"] and not o["schnorr_one_bad"], o
    assert o["ecdsa_bitmap_ok"] and o["ecdsa_count_ok"] and o["keyed"] > 0, o


def test_bench_two_ranks_full_per_rank_size():
    """`bench.py --gpus 2 --oversubscribe` at the REAL per-rank size of BASELINE config 5 (2^21 verifications per rank), two
    ranks sharing GPU 0: the corrupted-bitmap guard on the gathered bitmap must pass and the line must describe 2^22 in
    total."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "2^22 in total" in d["config"]["workload"], d["config"]


def test_bench_step_through_rccl_group_of_one():
    """bench.py with S2K_FORCE_COLLECTIVES=1: one rank, but the step's bitmap all-gather and count all-reduce, the barrier
    and the MAX over ranks of the time all go through RCCL (backend nccl), with the same guards on the gathered bitmap."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["S2K_FORCE_COLLECTIVES"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-pcie",
                        "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "RCCL group of one" in d["config"]["parallelism"], d["config"]


def test_compiler_miscompile_reproducer_and_shipped_shape(eng, tmp_path):
    """hipcc 7.2 miscompiled the first form of the point worklist kernel (per-lane garbage for 1 * G);
    the shipped kernel avoids the shape.  This builds tools/dbg/miscompile_point_fallback.hip with the box's compiler and
    runs it: the two variants whose shapes the product code relies on (kB: no generator part, kD: no worklist
    indirection) must be right - if a toolchain change breaks one of THOSE, the worklist kernels are at risk and the
    end-to-end tests below would be the only other witness - and the status of the four known-bad variants is recorded.
    Then the shipped worklist kernel itself: k * P for every lane forced onto it (P + (-P) sums inside the ladder)."""
    import shutil
    import secp256k1_voi_amd as S
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "mpf")
    src = os.path.join(ROOT, "tools", "dbg", "miscompile_point_fallback.hip")
    c = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", src, "-o", exe], capture_output=True, text=True, timeout=900)
    assert c.returncode == 0, c.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    status = {}
    for ln in r.stdout.splitlines():
        name, rest = ln.split(":")
        status[name.strip()] = int(rest.split()[0])
    assert set(status) == {"kA", "kB", "kC", "kD", "kE", "kF"}, r.stdout
    assert status["kB"] == 0 and status["kD"] == 0, status
    print("miscompile reproducer with this toolchain (items wrong of 8):", status)
    # the shipped kernel: u1 = 0, u2 = n - 1 ... on points whose ladder meets P - P; every result against big integers
    rnd = __import__("random").Random(9)
    pts, ks = [], []
    for _ in range(300):
        p = R.mul(rnd.randrange(1, R.N), R.G)
        pts.append(p)
        ks.append(rnd.choice([1, 2, R.N - 1, R.N - 2, R.LAMBDA, R.N - R.LAMBDA, rnd.randrange(R.N)]))
    recs = [b"\x04" + b32(p[0]) + b32(p[1]) for p in pts]
    out = eng.double_scalar_mult_basepoint_batch_ex(S.IMPL_FAST, None, [b32(k) for k in ks], recs)
    for k, p, o in zip(ks, pts, out):
        e = R.mul(k, p)
        assert bytes(o) == (bytes(65) if e is None else b"\x04" + b32(e[0]) + b32(e[1]))


def test_second_context_shares_generator_tables(eng, oracle):
    """The 3 GiB generator tables are shared by the contexts of a device (reference counted): a second context neither
    rebuilds nor holds another copy, both give the oracle's verdicts, and the first keeps working after the second is gone."""
    import time

    import torch
    import secp256k1_voi_amd as S
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 600, seed=77, corrupt_every=5)
    exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], nthreads=os.cpu_count() or 1)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    t0 = time.perf_counter()
    e2 = S.Engine(0)
    dt = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (512 << 20), "a second context took %.1f MiB" % ((free0 - free1) / 2**20)
    assert dt < 0.1, "second context took %.3f s to create" % dt
    assert np.array_equal(e2.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"]), exp)
    e2.close()
    assert np.array_equal(eng.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"]), exp)


def test_synthetic_signatures_against_libcrypto(eng):
    """The synthetic workload is made by the engine itself (synth.py: batched base multiplications and scalar arithmetic on
    the device).  An oracle that shares nothing with this repository - libcrypto's ECDSA_do_verify - accepts 2^12 of
    those signatures and rejects them once damaged, exactly as the engine does (second oracle)."""
    import openssl_ref
    if not openssl_ref.available():
        pytest.skip("no usable libcrypto")
    from secp256k1_voi_amd.synth import synth_batch
    n = 1 << 12
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, n, 256, seed=31337))
    rng = np.random.default_rng(8)
    for i in range(0, n, 5):
        a = (dig, r, s)[int(rng.integers(0, 3))]
        a[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    got = eng.ecdsa_verify_batch(pub, dig, r, s)
    ref = np.array([openssl_ref.ecdsa_verify(bytes(pub[i]), bytes(dig[i]), bytes(r[i]), bytes(s[i])) for i in range(n)], dtype=np.uint8)
    assert np.array_equal(got, ref)
    assert int(got.sum()) >= n - n // 5 - 1 and int(got.sum()) < n


@pytest.mark.parametrize("layout", [1, 2, 3, 4])      # S2K_KEYSET_CHUNKS, S2K_KEYSET_JOINT, S2K_KEYSET_JOINT5, S2K_KEYSET_JOINT6
def test_keyset_matches_batch_verifier(eng, oracle, layout):
    """s2k_keyset_*: tables of a fixed key list built once, signatures name their key by index.  Same verdicts as
    s2k_ecdsa_verify_batch on the expanded key array and as the oracle: valid and damaged signatures, keys that are no
    public keys (off the curve, coordinate >= p, all zero: every signature under them false), key indices outside the set
    (false), ragged use of the keys (some unused, some used hundreds of times), RejectMalleable, two calls on one set."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    rng = np.random.default_rng(77)
    nk, n = 300, 20000
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, n, nk, seed=555))
    keys, inv = np.unique(pub, axis=0, return_inverse=True)          # the set: the distinct keys, in sorted order
    kidx = inv.astype(np.uint32)
    keys = keys.copy()
    # three keys of the set that are no public keys
    bad_keys = [5, 77, len(keys) - 1]
    keys[5, 63] ^= 1                                                   # off the curve
    keys[77, :32] = 0xFF                                               # x >= p
    keys[len(keys) - 1, :] = 0
    for i in range(0, n, 6):                                           # damaged signatures
        a = (dig, r, s)[int(rng.integers(0, 3))]
        a[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    kidx[11] = len(keys)                                               # outside the set
    kidx[12] = 0xFFFFFFFF
    ks = eng.keyset_create(keys, layout)
    assert ks.layout() == layout
    joint_bytes = {1: 0, 2: 32 * 128 * 80, 3: (26 * 512 + 2) * 64, 4: (22 * 2048 + 2) * 64}[layout]
    assert len(ks) == len(keys) and len(keys) * (36864 + joint_bytes) <= ks.device_bytes() <= len(keys) * (36864 + joint_bytes) + (1 << 20)
    vk = ks.valid_keys()
    assert [k for k in range(len(keys)) if not vk[k]] == sorted(bad_keys)
    got = eng.ecdsa_verify_batch_keyset(ks, kidx, dig, r, s)
    # the same through the batch verifier on the expanded key array (indices outside the set: an all-zero key)
    full = np.zeros((n, 64), np.uint8)
    inside = kidx < len(keys)
    full[inside] = keys[kidx[inside]]
    for mode in (S.KEYS_AUTO, S.KEYS_OFF):
        eng.set_key_grouping(mode)
        assert np.array_equal(got, eng.ecdsa_verify_batch(full, dig, r, s))
    eng.set_key_grouping(S.KEYS_AUTO)
    m = 3000
    assert np.array_equal(got[:m], oracle.ecdsa_verify_batch(full[:m], dig[:m], r[:m], s[:m], nthreads=os.cpu_count() or 1))
    assert not got[11] and not got[12] and not got[np.isin(kidx, bad_keys)].any()
    assert 0.7 * n < int(got.sum()) < n
    # high-s forms of valid signatures: accepted, rejected with RejectMalleable; second call on the same set
    s2 = s.copy()
    for i in range(1, 200, 2):
        s2[i] = np.frombuffer(b32(R.N - int.from_bytes(bytes(s[i]), "big")), np.uint8)
    a1 = eng.ecdsa_verify_batch_keyset(ks, kidx, dig, r, s2)
    a2 = eng.ecdsa_verify_batch_keyset(ks, kidx, dig, r, s2, reject_malleable=True)
    assert np.array_equal(a1, eng.ecdsa_verify_batch(full, dig, r, s2))
    assert np.array_equal(a2, eng.ecdsa_verify_batch(full, dig, r, s2, reject_malleable=True))
    assert int(a1.sum()) > int(a2.sum())
    ks.close()


@pytest.mark.parametrize("layout", [1, 2, 3, 4])
def test_keyset_worklist_and_full_size(eng, layout):
    """Key set at BASELINE size: 2^20 signatures of 2^16 keys against the batch verifier (grouping on), and an adversarial
    batch whose every lane ends on the complete-formula worklist (u1 G + u2 Q = infinity), which must reach the worklist
    kernel's key-set form and come back all false."""
    from secp256k1_voi_amd.synth import synth_all_fallback_batch, synth_batch
    n, nk = 1 << 20, (1 << 16 if layout != 4 else 1 << 13)      # (6-bit joint tables: 3.6 MiB per key)
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, n, nk, seed=4))
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    ks = eng.keyset_create(keys, layout)
    r[::9, 3] ^= 4
    got = eng.ecdsa_verify_batch_keyset(ks, inv.astype(np.uint32), dig, r, s)
    assert np.array_equal(got, eng.ecdsa_verify_batch(pub, dig, r, s))
    assert int(got.sum()) == n - len(range(0, n, 9))
    ks.close()
    m = 1 << 14
    pub, dig, r, s = (np.array(a) for a in synth_all_fallback_batch(eng, m, 64, seed=6))
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    ks = eng.keyset_create(keys, layout)
    got = eng.ecdsa_verify_batch_keyset(ks, inv.astype(np.uint32), dig, r, s)
    assert not got.any() and eng.key_grouping_stats()["complete"] == 0     # (decided in the ladder's final addition since round 4)
    got = eng.ecdsa_verify_batch_keyset(ks, inv.astype(np.uint32), dig, r, s, force_worklist=True)
    assert not got.any() and eng.key_grouping_stats()["complete"] == m     # the key-set form of the worklist kernel
    ks.close()


@pytest.mark.parametrize("env", [{}, {"S2K_MSM_WIDE_PAIRS": "1"}, {"S2K_MSM_SPLIT_WINDOW": "3"}, {"S2K_MSM_LANES": "4096", "S2K_MSM_CHUNK_LOG2": "5"}])
def test_msm_randomised_stress(env):
    """tools/stress_msm.py: random sizes around every geometry switch, scalar patterns that stress the signed recoding and
    spread buckets over many ranges, repeated / negated / identity points, against big-integer arithmetic on points with
    known discrete logarithms - as shipped, with two-word sort pairs, with the two-part flow, and with long ranges."""
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_msm.py"), "30", str(len(env) + 11)], env=e,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "ok:" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_schnorr_batch_randomised_stress():
    """tools/stress_rlc.py: random batch sizes and key multiplicities through the two-stream whole-batch check (accepts the
    valid batch, rejects it with damaged signatures) and the bisection (per-signature verdicts equal the per-signature verifier's)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_rlc.py"), "25", "77"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "ok:" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_full_size_ragged_keys_differential_vs_oracle(eng, oracle):
    """2^20 signatures whose keys repeat 1, 3, 4 (the table threshold), 5, 64 and 2048 times, shuffled together, one in
    six damaged (bit flips in r / s / digest / key, another signer's key): the batch splits between the per-key tables
    and the general ladder inside one call, and every verdict must be the CPU oracle's - with the grouping on (default
    threshold), forced, and off."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    parts = [(1 << 18, 1 << 18), (3 << 16, 1 << 16), (1 << 18, 1 << 16), (5 << 14, 1 << 14), (1 << 17, 1 << 11), (1 << 16, 1 << 5)]
    used = sum(p[0] for p in parts)
    parts.append(((1 << 20) - used, max(((1 << 20) - used) >> 3, 1)))
    cols = [[], [], [], []]
    for j, (m, k) in enumerate(parts):
        for c, a in zip(cols, synth_batch(eng, m, k, seed=900 + j)):
            c.append(np.array(a))
    pub, dig, r, s = (np.concatenate(c) for c in cols)
    n = pub.shape[0]
    assert n == 1 << 20
    rng = np.random.default_rng(901)
    perm = rng.permutation(n)
    pub, dig, r, s = pub[perm], dig[perm], r[perm], s[perm]
    kind = rng.integers(0, 30, size=n)
    idx = lambda k: np.nonzero(kind == k)[0]
    for k, a in ((0, r), (1, s), (2, dig), (3, pub)):
        i = idx(k)
        a[i, rng.integers(0, a.shape[1], size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
    i = idx(4)
    pub[i] = pub[(i + 12345) % n]
    exp = oracle.ecdsa_verify_batch(pub, dig, r, s, nthreads=os.cpu_count() or 1)
    assert 0 < int(exp.sum()) < n and exp[kind >= 5].all()
    try:
        for mode in (S.KEYS_AUTO, S.KEYS_ALWAYS, S.KEYS_OFF):
            eng.set_key_grouping(mode)
            got = eng.ecdsa_verify_batch(pub, dig, r, s)
            assert np.array_equal(got, exp), (mode, np.nonzero(got != exp)[0][:10])
            st = eng.key_grouping_stats()
            if mode == S.KEYS_AUTO:
                assert st["keyed"] > 0 and st["general"] > 0, st      # both ladders took part
    finally:
        eng.set_key_grouping(S.KEYS_AUTO)
