"""Signatures that share public keys (keyed.hip, k_verify_fast<MODE_ECDSA_KEYED>): the verdicts must not
depend on how the batch is grouped.  Every case runs the same inputs with grouping off, automatic and
forced, compares all three with the CPU oracle (the reference's algorithm, oracle/), and checks through
s2k_ctx_key_grouping_stats that the path under test actually ran."""
import os

import numpy as np
import pytest

import pyref as R

pytestmark = pytest.mark.gpu
R_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
P_FIELD = 2**256 - 2**32 - 977


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    e = S.Engine(0)
    yield e
    e.set_key_grouping(S.KEYS_AUTO)


def _device_run(eng, pub, dig, r, s, flags=0):
    import torch
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pub, dig, r, s)]
    out = torch.empty(len(pub), dtype=torch.uint8, device=dev)
    eng.ecdsa_verify_batch_device(len(pub), *(x.data_ptr() for x in t), out.data_ptr(), flags=flags)
    torch.cuda.synchronize()
    return out.cpu().numpy(), eng.key_grouping_stats()


def _ragged_batch(eng, seed, sizes):
    """Signatures of len(sizes) keys, key j signing sizes[j] of them, shuffled."""
    from secp256k1_voi_amd.synth import synth_batch
    key_idx = np.random.default_rng(seed).permutation(np.repeat(np.arange(len(sizes)), sizes))
    return synth_batch(eng, len(key_idx), len(sizes), seed=seed, key_idx=key_idx)


def _damage(pub, dig, r, s, seed):
    from secp256k1_voi_amd.synth import N_ORDER
    n = len(pub)
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 24, size=n)
    idx = lambda k: np.nonzero(kind == k)[0]
    i = idx(0); r[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
    i = idx(1); s[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
    i = idx(2); dig[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
    i = idx(3); r[i] = 0
    i = idx(4); s[i] = 0
    i = idx(5); r[i] = np.frombuffer(N_ORDER.to_bytes(32, "big"), np.uint8)
    i = idx(6); pub[i] = pub[(i + 1) % n]                      # somebody else's key
    return kind


def _check_all_modes(eng, oracle, pub, dig, r, s, expect_keyed=None, **grouping):
    import secp256k1_voi_amd as S
    exp = oracle.ecdsa_verify_batch(pub, dig, r, s, nthreads=os.cpu_count() or 1)
    n = len(pub)
    eng.set_key_grouping(S.KEYS_OFF)
    got, st = _device_run(eng, pub, dig, r, s)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]
    assert st["keyed"] == 0 and st["tables"] == 0
    eng.set_key_grouping(S.KEYS_AUTO, **grouping)
    got, st_auto = _device_run(eng, pub, dig, r, s)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]
    assert st_auto["keyed"] + st_auto["general"] == n
    if expect_keyed is not None:
        assert st_auto["keyed"] == expect_keyed, st_auto
    eng.set_key_grouping(S.KEYS_ALWAYS, **{k: v for k, v in grouping.items() if k != "min_group"})
    got, st_all = _device_run(eng, pub, dig, r, s)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]
    assert st_all["keyed"] + st_all["general"] == n
    eng.set_key_grouping(S.KEYS_AUTO)
    return exp, st_auto, st_all


def test_ragged_groups_match_oracle(eng, oracle):
    """Keys with 1 .. 70 signatures each, shuffled, a quarter of the signatures damaged: identical verdicts
    with grouping off / automatic / forced; automatic grouping puts exactly the groups of >= 4 (the default
    threshold), or of >= 3 when asked, on the tables."""
    sizes = np.array([1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 15, 16, 17, 33, 64, 70] * 6 + [1] * 100)
    pub, dig, r, s = _ragged_batch(eng, 11, sizes)
    _damage(pub, dig, r, s, 12)
    # damage kind 6 copies other keys around, so recount the groups from the bytes
    _, counts = np.unique(pub, axis=0, return_counts=True)
    exp, st_auto, st_all = _check_all_modes(eng, oracle, pub, dig, r, s, expect_keyed=int(counts[counts >= 4].sum()))
    assert st_auto["tables"] == int((counts >= 4).sum())
    _, st3, _ = _check_all_modes(eng, oracle, pub, dig, r, s, expect_keyed=int(counts[counts >= 3].sum()), min_group=3)
    assert st3["tables"] == int((counts >= 3).sum())
    assert st_all["keyed"] == len(pub) and st_all["tables"] == len(counts)
    assert 0 < exp.sum() < len(pub)


def test_invalid_keys_in_groups(eng, oracle):
    """Groups whose key is not a public key (off the curve, coordinate >= p, all zero): every member is
    rejected, exactly as NewPublicKey would refuse the key (secec.go:188-216), on the table path too."""
    sizes = np.array([8] * 40 + [1] * 30)
    pub, dig, r, s = _ragged_batch(eng, 21, sizes)
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    bad_off = inv == 0
    pub[bad_off, 63] ^= 1                                         # y off the curve
    big = np.frombuffer((P_FIELD + 5).to_bytes(32, "big"), np.uint8)
    pub[inv == 1, :32] = big                                      # x >= p
    pub[inv == 2, 32:] = np.frombuffer(P_FIELD.to_bytes(32, "big"), np.uint8)   # y = p
    pub[inv == 3] = 0
    # a valid point that is not the signer's: x = G.x with -G.y
    gx = np.frombuffer(R.GX.to_bytes(32, "big"), np.uint8)
    gyn = np.frombuffer((P_FIELD - R.GY).to_bytes(32, "big"), np.uint8)
    pub[inv == 4, :32] = gx
    pub[inv == 4, 32:] = gyn
    exp, st_auto, _ = _check_all_modes(eng, oracle, pub, dig, r, s)
    for k in range(5):
        assert exp[inv == k].sum() == 0
    assert exp[inv >= 5].all()
    assert st_auto["keyed"] >= 8 * 35


def test_hash_table_pressure(eng, oracle):
    """The grouping must stay exact whatever the hash table does: 64 slots for 600 keys (probe chains hit their
    limit, those signatures take the general kernel), 1024 slots (long chains), and caps on the number of tables."""
    sizes = np.array([8] * 500 + [1] * 100)
    pub, dig, r, s = _ragged_batch(eng, 31, sizes)
    _damage(pub, dig, r, s, 32)
    _, st, _ = _check_all_modes(eng, oracle, pub, dig, r, s, hash_bits=6)
    assert 0 < st["keyed"] <= 64 * 12 and st["general"] > 0
    _, st, _ = _check_all_modes(eng, oracle, pub, dig, r, s, hash_bits=10)
    assert st["keyed"] > 3000
    # a cap of 5 tables for 4100 signatures raises the threshold to 820 signatures per key: nothing qualifies
    _, st, st_all = _check_all_modes(eng, oracle, pub, dig, r, s, max_tables=5)
    assert st["tables"] == 0 and st["keyed"] == 0 and st["general"] == len(pub)
    assert st_all["keyed"] == 0
    # a cap of 300: threshold 14, still nothing; 600: threshold 7, the groups of 8 and more get tables
    _, st, _ = _check_all_modes(eng, oracle, pub, dig, r, s, max_tables=300)
    assert st["keyed"] == 0
    _, st, _ = _check_all_modes(eng, oracle, pub, dig, r, s, max_tables=600)
    assert 0 < st["tables"] <= 600 and st["keyed"] >= 7 * st["tables"]


def test_exceptional_ladders_with_shared_keys(eng, oracle):
    """Signatures built so that the incomplete formulas break down in the FINAL addition, many per key: u1 G = -u2 Q
    (synth_all_fallback_batch: R = infinity, the reference says false) and u1 G = u2 Q (synth_equal_points_batch: R = 2 u1 G).
    The ladders must notice (Z = 0) and decide: opposite points in place - nothing on the worklist -, equal points by the
    worklist kernel's short form (tagged entries: twice the generator part), the general ladder in place.  The same inputs
    with S2K_ECDSA_FORCE_WORKLIST: every lane through the complete kernel, same verdicts."""
    from secp256k1_voi_amd.synth import synth_all_fallback_batch, synth_batch, synth_equal_points_batch
    n = 4096
    pub, e, r, s = synth_all_fallback_batch(eng, n, 64, seed=41)
    exp, st_auto, _ = _check_all_modes(eng, oracle, pub, e, r, s, expect_keyed=n)
    assert st_auto["complete"] == 0
    assert not exp.any()                                          # R = infinity: the reference says false
    assert not eng.ecdsa_verify_batch(pub, e, r, s, force_worklist=True).any() and eng.key_grouping_stats()["complete"] == n
    # equal points; a few of them made VALID: r = x(2 u1 G) mod n, s = r / u2, e = u1 s with u1 = u2 d needs r first -
    # take r from the point itself: R = 2 u2 d G
    pub, e, r, s = (np.array(a) for a in synth_equal_points_batch(eng, n, 64, seed=42))
    from secp256k1_voi_amd import OP_INV, OP_MUL
    m = 256
    u2inv, _ = eng.fn_op_batch(OP_INV, s[:m])                      # u2 = r / s
    u2, _ = eng.fn_op_batch(OP_MUL, r[:m], u2inv)
    u1 = np.zeros((m, 32), np.uint8)
    for i in range(m):                                            # u1 = e / s
        u1[i] = np.frombuffer(oracle.fn_mul(bytes(e[i]), oracle.fn_inv(bytes(s[i]))), np.uint8)
    two_u1, _ = eng.fn_op_batch(2, u1, u1)                         # OP_ADD
    Rp = eng.scalar_base_mult_batch(two_u1)
    for i in range(m):
        rn = int.from_bytes(bytes(Rp[i, 1:33]), "big") % R_N
        r[i] = np.frombuffer(rn.to_bytes(32, "big"), np.uint8)
        # keep u1 = u2 d and u2: s = r / u2, e = u1 s
        s[i] = np.frombuffer(oracle.fn_mul(bytes(r[i]), oracle.fn_inv(bytes(u2[i]))), np.uint8)
        e[i] = np.frombuffer(oracle.fn_mul(bytes(u1[i]), bytes(s[i])), np.uint8)
    exp, st_auto, _ = _check_all_modes(eng, oracle, pub, e, r, s, expect_keyed=n)
    assert exp[:m].all() and not exp[m:].any()
    assert st_auto["complete"] == n                               # tagged worklist entries: R = 2 u1 G
    assert np.array_equal(eng.ecdsa_verify_batch(pub, e, r, s, force_worklist=True), exp)
    # a valid batch through the forced worklist
    vb = synth_batch(eng, 2048, 32, seed=43)
    assert eng.ecdsa_verify_batch(*vb, force_worklist=True).all() and eng.key_grouping_stats()["complete"] == 2048


def test_chosen_scalars_on_tables(eng, oracle):
    """VALID signatures with chosen u2 = r/s (and random u1): for a key with known d, R = (u1 + u2 d) G,
    r = x(R) mod n, s = r / u2, e = u1 s.  The u2 are the values around which a windowed ladder over
    2^(16c) Q has its corner cases: tiny values, powers of two at the chunk borders, lambda and its
    neighbours (one GLV half collapses), n - small.  Whatever happens inside the ladder (an addition
    that degenerates sends the lane to the complete kernel), every signature must verify."""
    from secp256k1_voi_amd.synth import N_ORDER
    n_keys, per_key = 8, 64
    rng = np.random.default_rng(81)
    lam = R.LAMBDA
    base = [1, 2, 3, 15, 16, 17, 255, 2**16 - 1, 2**16, 2**16 + 1, 2**32, 2**48 + 2**16, 2**64, 2**112, 2**116, 2**116 + 1,
            2**127, 2**128 - 1, 2**128, 2**128 + 1, 2**129, lam, lam + 1, lam - 1, 2 * lam % N_ORDER, (lam * 2**16) % N_ORDER,
            (lam + 2**16) % N_ORDER, N_ORDER - lam, N_ORDER - 1, N_ORDER - 2, N_ORDER - 2**16, (N_ORDER - 1) // 2, (N_ORDER + 1) // 2,
            (1 + lam) * 2**116 % N_ORDER, sum(2**(16 * c) for c in range(8)), sum(15 * 2**(4 * i + 1) for i in range(32)) + 1]
    u2 = [base[i % len(base)] if i % 2 == 0 else (base[(i // 2) % len(base)] * int(rng.integers(1, 1 << 20)) + int(rng.integers(0, 3))) % N_ORDER or 1
          for i in range(n_keys * per_key)]
    d = [int.from_bytes(rng.bytes(32), "big") % (N_ORDER - 1) + 1 for _ in range(n_keys)]
    u1 = [int.from_bytes(rng.bytes(32), "big") % N_ORDER for _ in u2]
    key = [i % n_keys for i in range(len(u2))]
    kR = [(a + b * d[k]) % N_ORDER or 1 for a, b, k in zip(u1, u2, key)]
    as_rows = lambda v: np.frombuffer(b"".join(int(x).to_bytes(32, "big") for x in v), np.uint8).reshape(-1, 32).copy()
    Rp = eng.scalar_base_mult_batch(as_rows(kR))
    Q = eng.scalar_base_mult_batch(as_rows(d))[:, 1:]
    rr = [int.from_bytes(bytes(Rp[i, 1:33]), "big") % N_ORDER for i in range(len(u2))]
    ss = [x * pow(b, -1, N_ORDER) % N_ORDER for x, b in zip(rr, u2)]
    ee = [a * x % N_ORDER for a, x in zip(u1, ss)]
    pub = np.ascontiguousarray(Q[key])
    exp, st_auto, _ = _check_all_modes(eng, oracle, pub, as_rows(ee), as_rows(rr), as_rows(ss), expect_keyed=len(u2))
    fixed = [i for i in range(len(u2)) if (u1[i] + u2[i] * d[key[i]]) % N_ORDER]
    assert exp[fixed].all()


def test_small_and_odd_batches(eng, oracle):
    """Batch sizes around the wave / workgroup / grouping thresholds, one key for everything."""
    from secp256k1_voi_amd.synth import synth_batch
    import secp256k1_voi_amd as S
    pub, dig, r, s = synth_batch(eng, 1500, 1, seed=51)
    r[::7, 5] ^= 1
    exp = oracle.ecdsa_verify_batch(pub, dig, r, s, nthreads=os.cpu_count() or 1)
    eng.set_key_grouping(S.KEYS_AUTO)
    for n in (1, 63, 255, 256, 257, 511, 1025, 1500):
        got, st = _device_run(eng, pub[:n], dig[:n], r[:n], s[:n])
        assert np.array_equal(got, exp[:n]), n
        assert st["keyed"] == (n if n >= 256 else 0), (n, st)
        assert st["tables"] == (1 if n >= 256 else 0)


def test_reject_malleable_on_tables(eng, oracle):
    from secp256k1_voi_amd.synth import N_ORDER, synth_batch
    import secp256k1_voi_amd as S
    n = 2048
    pub, dig, r, s = synth_batch(eng, n, 16, seed=61)
    flip = np.arange(n) % 3 == 0                                  # s -> n - s
    for i in np.nonzero(flip)[0]:
        s[i] = np.frombuffer((N_ORDER - int.from_bytes(bytes(s[i]), "big")).to_bytes(32, "big"), np.uint8)
    eng.set_key_grouping(S.KEYS_AUTO)
    for rm in (False, True):
        exp = oracle.ecdsa_verify_batch(pub, dig, r, s, reject_malleable=rm, nthreads=os.cpu_count() or 1)
        got, st = _device_run(eng, pub, dig, r, s, flags=S.REJECT_MALLEABLE if rm else 0)
        assert np.array_equal(got, exp)
        assert st["keyed"] == n
    assert exp.sum() == n - flip.sum()


def test_full_size_tables_vs_general(eng):
    """2^20 signatures of 2^16 keys (the bench's workload) with seeded damage: the table path and the general
    path give the same 2^20 verdicts (the general path is compared with the oracle at this size in
    test_gpu_round2.test_full_size_differential_vs_oracle, which since the grouping is on by default also
    exercises the tables)."""
    from secp256k1_voi_amd.synth import synth_batch
    import secp256k1_voi_amd as S
    n = 1 << 20
    pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=71)
    kind = _damage(pub, dig, r, s, 72)
    eng.set_key_grouping(S.KEYS_OFF)
    a, st0 = _device_run(eng, pub, dig, r, s)
    eng.set_key_grouping(S.KEYS_AUTO)
    b, st1 = _device_run(eng, pub, dig, r, s)
    assert np.array_equal(a, b)
    assert st0["keyed"] == 0 and st1["keyed"] > n * 0.9 and st1["tables"] >= (1 << 16) * 0.99
    assert a[kind >= 7].all() and a[kind <= 5].sum() == 0


# ---- BIP-340 whole-batch check with one term pair per DISTINCT key (msm.hip, aggregated form) ----
def test_schnorr_batch_aggregates_keys(eng, oracle):
    """s2k_schnorr_batch_verify_rlc / the bisecting entry point on batches whose keys repeat: group sizes from
    1 to 3000 (more than one virtual group of 1024), an invalid key shared by several signatures, hash tables
    too small for the keys (signatures that find no slot form their own groups), damaged signatures.
    Whole-batch verdict == all per-signature verdicts; per-signature verdicts == the oracle's."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    n = 1 << 18                                                   # above the bisection's leaf size: combinations run
    seed = bytes(range(32))
    # keys: i mod 5000 for the first half, then a block of 3000 signatures under key 7, then distinct keys
    pk, msgs, sig = synth_schnorr_batch(eng, n, n, seed=91)       # n distinct keys, valid signatures
    half = n // 2
    pk2, msgs2, sig2 = synth_schnorr_batch(eng, half + 3000, 5000, seed=92)
    pk[:half], msgs[:half], sig[:half] = pk2[:half], msgs2[:half], sig2[:half]
    big = np.nonzero(np.arange(half + 3000) % 5000 == 7)[0]
    pkb, mb, sb = synth_schnorr_batch(eng, 3000, 1, seed=93)
    pk[half:half + 3000], msgs[half:half + 3000], sig[half:half + 3000] = pkb, mb, sb
    perm = np.random.default_rng(94).permutation(n)
    pk, msgs, sig = pk[perm].copy(), msgs[perm].copy(), sig[perm].copy()

    def check(pk, msgs, sig, expect_bad):
        ok = eng.schnorr_batch_verify_rlc(pk, msgs, sig, seed)
        assert ok == (len(expect_bad) == 0)
        v, st = eng.schnorr_verify_batch_auto(pk, msgs, sig, seed, return_stats=True)
        exp = np.ones(n, dtype=np.uint8)
        exp[list(expect_bad)] = 0
        assert np.array_equal(v, exp), (np.nonzero(v != exp)[0][:10], st)
        for i in list(expect_bad)[:8]:
            assert oracle.schnorr_verify(bytes(pk[i]), bytes(msgs[i]), bytes(sig[i])) != 1
        return st

    for bits in (0, 12):                                           # default table; 4096 slots for ~136 000 keys
        eng.set_key_grouping(S.KEYS_AUTO, hash_bits=bits)
        check(pk, msgs, sig, [])
        # one bad signature inside the 3000-signature group, one in a singleton group
        bad = sig.copy()
        i_big = int(np.nonzero((pk == pkb[0]).all(axis=1))[0][1500])
        i_one = int(np.nonzero(perm >= half + 3000)[0][17])
        bad[i_big, 40] ^= 1
        bad[i_one, 63] ^= 0x80
        check(pk, msgs, bad, [i_big, i_one])
        # a key that is no x-coordinate, shared by every signature of one 5000-cycle key: all of them invalid
        not_x = next(x for x in range(2, 100) if R.lift_x(x, 0) is None)
        victim = pk[int(np.nonzero(perm < half)[0][3])].copy()
        members = np.nonzero((pk == victim).all(axis=1))[0]
        assert len(members) >= 20
        pk_bad = pk.copy()
        pk_bad[members] = np.frombuffer(R.b32(not_x), np.uint8)
        check(pk_bad, msgs, sig, list(members))
    eng.set_key_grouping(S.KEYS_AUTO)


# ---- BIP-340 per-signature verification with per-key tables (MODE_SCHNORR_KEYED / _LEFT) ----
def test_schnorr_per_signature_on_tables(eng, oracle):
    """s2k_schnorr_verify_batch with grouping off / automatic / forced: ragged groups of x-only keys, damaged
    r / s / message, keys that are no x-coordinate or >= p (shared by whole groups), r = a non-residue;
    all three modes give the oracle's verdicts (SchnorrPublicKey.Verify, schnorr.go:221-253)."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    n_keys, n = 300, 6000
    pk, msgs, sig = synth_schnorr_batch(eng, n, n_keys, seed=101)          # key i mod 300: 20 signatures each
    pk2, msgs2, sig2 = synth_schnorr_batch(eng, 500, 500, seed=102)        # 500 singletons
    pk, msgs, sig = np.concatenate([pk, pk2]), np.concatenate([msgs, msgs2]), np.concatenate([sig, sig2])
    n = len(pk)
    perm = np.random.default_rng(103).permutation(n)
    pk, msgs, sig = pk[perm].copy(), msgs[perm].copy(), sig[perm].copy()
    rng = np.random.default_rng(104)
    kind = rng.integers(0, 20, size=n)
    i = np.nonzero(kind == 0)[0]; sig[i, rng.integers(0, 32, size=i.size)] ^= 1          # r
    i = np.nonzero(kind == 1)[0]; sig[i, 32 + rng.integers(0, 32, size=i.size)] ^= 0x10  # s
    i = np.nonzero(kind == 2)[0]; msgs[i, 0] ^= 0x80
    i = np.nonzero(kind == 3)[0]; sig[i, 32:] = 0xFF                                     # s >= n
    not_x = next(x for x in range(2, 100) if R.lift_x(x, 0) is None)
    keys = np.unique(pk, axis=0)
    for j, bad in ((0, R.b32(not_x)), (1, R.b32(R.P + 1)), (2, bytes(32))):
        pk[(pk == keys[j]).all(axis=1)] = np.frombuffer(bad, np.uint8)
    exp = np.array([1 if oracle.schnorr_verify(bytes(pk[j]), bytes(msgs[j]), bytes(sig[j])) == 1 else 0 for j in range(n)], dtype=np.uint8)
    assert 0 < exp.sum() < n
    for mode in (S.KEYS_OFF, S.KEYS_AUTO, S.KEYS_ALWAYS):
        eng.set_key_grouping(mode)
        got = eng.schnorr_verify_batch(pk, msgs, sig)
        assert np.array_equal(got, exp), (mode, np.nonzero(got != exp)[0][:10])
        st = eng.key_grouping_stats()
        if mode == S.KEYS_OFF:
            assert st["keyed"] == 0
        elif mode == S.KEYS_AUTO:
            assert st["keyed"] >= 5000 and st["general"] >= 400 and st["keyed"] + st["general"] == n
        else:
            assert st["keyed"] == n
    eng.set_key_grouping(S.KEYS_AUTO)


def test_schnorr_exceptional_ladders_on_tables(eng, oracle):
    """BIP-340 signatures with s*G - e*P = infinity (R' has no x: invalid), many per key: the table ladder ends
    with Z = 0 and the worklist kernel decides, as on the general path."""
    import hashlib
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import N_ORDER
    rng = np.random.default_rng(111)
    n_keys, per = 16, 64
    d = [int.from_bytes(rng.bytes(32), "big") % (N_ORDER - 1) + 1 for _ in range(n_keys)]
    P = eng.scalar_base_mult_batch(np.frombuffer(b"".join(R.b32(x) for x in d), np.uint8).reshape(-1, 32))
    d = [N_ORDER - x if P[j, 64] & 1 else x for j, x in enumerate(d)]          # even-y keys
    pks, msgs, sigs = [], [], []
    th = hashlib.sha256(b"BIP0340/challenge").digest()
    for j in range(n_keys * per):
        k = j % n_keys
        pkb = bytes(P[k, 1:33])
        r = rng.bytes(32)                         # any r: e is fixed by (r, pk, m), then s = e d makes s G - e P vanish
        m = rng.bytes(32)
        e = int.from_bytes(hashlib.sha256(th + th + r + pkb + m).digest(), "big") % N_ORDER
        pks.append(pkb); msgs.append(m); sigs.append(r + R.b32(e * d[k] % N_ORDER))
    pk = np.frombuffer(b"".join(pks), np.uint8).reshape(-1, 32)
    mm = np.frombuffer(b"".join(msgs), np.uint8).reshape(-1, 32)
    sg = np.frombuffer(b"".join(sigs), np.uint8).reshape(-1, 64)
    eng.set_key_grouping(S.KEYS_OFF)
    assert not eng.schnorr_verify_batch(pk, mm, sg).any()      # 1024 signatures: the wave-per-signature ladder, complete formulas
    eng.set_small_batch_max(0)                                 # ... and the lane kernels with their worklist
    eng.set_mid_batch_max(0)
    try:
        for mode in (S.KEYS_OFF, S.KEYS_AUTO):
            eng.set_key_grouping(mode)
            got = eng.schnorr_verify_batch(pk, mm, sg)
            assert not got.any()
            st = eng.key_grouping_stats()
            assert st["complete"] >= len(pk) - 8       # (a random r that is no x-coordinate is rejected before the ladder matters)
    finally:
        eng.set_small_batch_max(3072)
        eng.set_mid_batch_max(32768)
    for j in range(0, len(pk), 97):
        assert oracle.schnorr_verify(bytes(pk[j]), bytes(mm[j]), bytes(sg[j])) != 1
    eng.set_key_grouping(S.KEYS_AUTO)


def test_randomised_grouping_stress():
    """tools/stress_keyed.py: random batch sizes, key reuse patterns, damage and grouping settings; ECDSA, BIP-340
    per-signature and whole-batch verdicts against the oracle (25 iterations here; the tool runs any number)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_keyed.py"), "25", "7"], capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "stress ok" in res.stdout


def test_two_part_flow(oracle):
    """S2K_KEYED_PARTS=2 (off by default): tables of the second half of the keys are built beside the
    first half's ladder.  Enough keys for the split to happen (>= 4096 tables), damaged signatures, ECDSA and
    BIP-340 per-signature verification: the oracle's verdicts, and a context of its own for the setting."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch, synth_schnorr_batch
    os.environ["S2K_KEYED_PARTS"] = "2"
    try:
        e2 = S.Engine(0)
    finally:
        del os.environ["S2K_KEYED_PARTS"]
    e2.set_mid_batch_max(0)            # (the BIP-340 batch below has 29400 signatures: this test is about the tables' flow)
    n_keys, n = 5000, 5000 * 9
    pub, dig, r, s = synth_batch(e2, n, n_keys, seed=121)
    _damage(pub, dig, r, s, 122)
    exp = oracle.ecdsa_verify_batch(pub, dig, r, s, nthreads=os.cpu_count() or 1)
    got = e2.ecdsa_verify_batch(pub, dig, r, s)
    st = e2.key_grouping_stats()
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]
    assert st["tables"] >= 4096 and st["keyed"] > n * 0.8
    m = 4200 * 7
    pk, msgs, sig = synth_schnorr_batch(e2, m, 4200, seed=123)
    sig[::11, 40] ^= 1
    e_s = np.array([1 if oracle.schnorr_verify(bytes(pk[j]), bytes(msgs[j]), bytes(sig[j])) == 1 else 0 for j in range(m)], dtype=np.uint8)
    g_s = e2.schnorr_verify_batch(pk, msgs, sig)
    assert np.array_equal(g_s, e_s)
    assert e2.key_grouping_stats()["tables"] >= 4096
    e2.close()
