"""Round-4 GPU tests: the boundary.  submit / wait (two batches in flight per context), the single-process group of
devices, EncodingCompactRecoverable in the encoded entry point.  Everything through the C-ABI, verdicts against the
synchronous entry points and the CPU oracle.  Needs a real MI355X.
"""
import os
import random

import numpy as np
import pytest

import pyref as R
from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b32 = R.b32
H = bytes.fromhex


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    return S.Engine(0)


def damaged_batch(eng, n, n_keys, seed):
    """synthetic valid signatures with every seventh item damaged somewhere (key, digest, r or s)"""
    from secp256k1_voi_amd.synth import synth_batch
    arrs = [np.array(a) for a in synth_batch(eng, n, n_keys, seed=seed)]
    rng = np.random.default_rng(seed)
    for i in range(0, n, 7):
        a = arrs[int(rng.integers(0, 4))]
        a[i, int(rng.integers(0, a.shape[1]))] ^= 1 << int(rng.integers(0, 8))
    return arrs


# ---- submit / wait ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pinned", [False, True])
def test_submit_wait_matches_synchronous_call(eng, oracle, pinned):
    """Five batches of different sizes (one empty, one below the grouping threshold, ragged ones) submitted back to back
    with up to three in flight: every ticket's verdicts equal the synchronous call's and the oracle's; the fourth submit
    retires the first ticket by itself and a later wait on it returns at once; s2k_poll never blocks and ends as done."""
    import secp256k1_voi_amd as S
    sizes = [70001, 200, 0, 300003, 4096]
    batches = [damaged_batch(eng, n, max(n // 11, 3), 400 + n) if n else [np.zeros((0, w), np.uint8) for w in (64, 32, 32, 32)]
               for n in sizes]
    refs = [eng.ecdsa_verify_batch(*b) if b[0].shape[0] else np.zeros(0, np.uint8) for b in batches]
    for b, ref in zip(batches, refs):
        m = min(b[0].shape[0], 2048)
        if m:
            assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(*(a[:m] for a in b), nthreads=os.cpu_count() or 1))
            assert 0 < int(ref.sum()) < ref.size
    if pinned:
        src = []
        for b in batches:
            pb = [S.pinned_array(a.shape) for a in b]
            for d, a in zip(pb, b):
                d[...] = a
            src.append(pb)
        outs = [S.pinned_array((b[0].shape[0],)) for b in batches]
    else:
        src, outs = batches, [None] * len(batches)
    for mode in (S.KEYS_AUTO, S.KEYS_OFF):
        eng.set_key_grouping(mode)
        try:
            tickets = []
            for b, o in zip(src, outs):
                if o is not None:
                    o[...] = 7
                tickets.append(eng.ecdsa_verify_batch_submit(*b, out=o))
            # the first tickets have been retired by the submits behind them; waiting in reverse order is allowed
            assert tickets[0].done()
            import time
            t_end = time.time() + 60
            while not tickets[-1].done():
                assert time.time() < t_end
            for t, ref in reversed(list(zip(tickets, refs))):
                assert np.array_equal(t.wait(), ref)
        finally:
            eng.set_key_grouping(S.KEYS_AUTO)
    assert [t.ticket for t in tickets] == sorted(t.ticket for t in tickets)
    # a ticket that was never issued is refused; a context without anything in flight waits for nothing
    with pytest.raises(S.EngineError):
        eng._wait(tickets[-1].ticket + 100)
    eng.wait_all()


def test_submit_wait_interleaved_with_synchronous_calls(eng, oracle):
    """A synchronous call on the context while two tickets are in flight uses the context's own workspaces: none of the
    three results is disturbed.  RejectMalleable travels with the submit."""
    n = 50000
    a = damaged_batch(eng, n, 900, 61)
    b = damaged_batch(eng, n, 5000, 62)
    c = damaged_batch(eng, 3000, 10, 63)
    # make some s high so that the low-s rule has something to reject
    for arrs in (a, b):
        s_int = [int.from_bytes(bytes(x), "big") for x in arrs[3][:64]]
        for i, v in enumerate(s_int):
            if 0 < v < R.N and i % 2:
                arrs[3][i] = np.frombuffer(b32(R.N - v), np.uint8)
    ref_a = eng.ecdsa_verify_batch(*a, reject_malleable=True)
    ref_b = eng.ecdsa_verify_batch(*b)
    ref_c = eng.ecdsa_verify_batch(*c)
    exp_a = oracle.ecdsa_verify_batch(*(x[:1024] for x in a), reject_malleable=True, nthreads=os.cpu_count() or 1)
    assert np.array_equal(ref_a[:1024], exp_a)
    ta = eng.ecdsa_verify_batch_submit(*a, reject_malleable=True)
    tb = eng.ecdsa_verify_batch_submit(*b)
    got_c = eng.ecdsa_verify_batch(*c)
    assert np.array_equal(got_c, ref_c)
    assert np.array_equal(tb.wait(), ref_b)
    assert np.array_equal(ta.wait(), ref_a)
    assert not np.array_equal(ref_a, eng.ecdsa_verify_batch(*a))      # (the rule did reject something)


def test_submit_wait_full_size_pipeline(eng, oracle):
    """2^20 signatures per batch, four batches through the two slots from pinned memory, each batch damaged at its own
    seeded positions: every ticket's verdicts are the expected pattern (a slot that delivered another batch's verdicts,
    or none, cannot pass), and the head of one batch is checked against the oracle."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    n = 1 << 20
    base = [np.array(x) for x in synth_batch(eng, n, 1 << 16, seed=0x5EC9)]
    bufs, outs, masks = [], [], []
    for k in range(4):
        pb = [S.pinned_array(x.shape) for x in base]
        for d, x in zip(pb, base):
            d[...] = x
        i = np.arange(n, dtype=np.uint64)
        bad = ((i * np.uint64(2654435761) + np.uint64(k * 7919 + 5)) % np.uint64(53)) == 0
        pb[3][bad, 31] ^= 1
        bufs.append(pb)
        masks.append(~bad)
        o = S.pinned_array((n,))
        o[...] = 9
        outs.append(o)
    tickets = [eng.ecdsa_verify_batch_submit(*bufs[0], out=outs[0]), eng.ecdsa_verify_batch_submit(*bufs[1], out=outs[1])]
    for k in range(2, 4):
        assert np.array_equal(tickets[k - 2].wait(), masks[k - 2].astype(np.uint8))
        tickets.append(eng.ecdsa_verify_batch_submit(*bufs[k], out=outs[k]))
    for k in range(2, 4):
        assert np.array_equal(tickets[k].wait(), masks[k].astype(np.uint8))
    m = 4096
    assert np.array_equal(outs[3][:m], oracle.ecdsa_verify_batch(*(x[:m] for x in bufs[3]), nthreads=os.cpu_count() or 1))


def test_encoded_submit_wait(eng, oracle):
    """s2k_ecdsa_verify_encoded_batch_submit: DER and compact forms of a damaged batch, two in flight, against the
    synchronous encoded call and the raw verifier."""
    import secp256k1_voi_amd as S
    n = 30000
    arrs = damaged_batch(eng, n, 700, 77)
    ref = eng.ecdsa_verify_batch(*arrs)

    def der_int(b):
        b = bytes(b).lstrip(b"\0") or b"\0"
        if b[0] & 0x80:
            b = b"\0" + b
        return b"\x02" + bytes([len(b)]) + b
    pubs = [b"\x04" + bytes(q) for q in arrs[0]]
    digs = [bytes(d) for d in arrs[1]]
    der = []
    for r, s in zip(arrs[2], arrs[3]):
        body = der_int(r) + der_int(s)
        der.append(b"\x30" + bytes([len(body)]) + body)
    compact = [bytes(r) + bytes(s) for r, s in zip(arrs[2], arrs[3])]
    sync_der = eng.ecdsa_verify_encoded_batch(pubs, digs, der)
    # (an r or s of zero or >= n does not parse: both paths say invalid)
    assert np.array_equal(sync_der, ref)
    t1 = eng.ecdsa_verify_encoded_batch_submit(pubs, digs, der)
    t2 = eng.ecdsa_verify_encoded_batch_submit(pubs, digs, compact, encoding=S.ENCODING_COMPACT)
    t3 = eng.ecdsa_verify_encoded_batch_submit(pubs, digs, der, digest_len=48)      # wrong digest length: all false
    assert np.array_equal(t1.wait(), ref)
    assert np.array_equal(t2.wait(), ref)
    assert not t3.wait().any()


# ---- several members in one process ----------------------------------------------------------------------------------
def test_group_two_members_one_device(eng, oracle):
    """A group with two members (both on device 0 on this pool: two contexts, two host threads) verifies contiguous
    shards of one batch; verdicts equal the single-context call's and the oracle's, for ragged sizes, sizes below the
    member count, an empty batch, grouping on and off, and two group batches in flight."""
    import secp256k1_voi_amd as S
    assert S.device_count() >= 1
    g = S.Group([0, 0])
    try:
        assert len(g) == 2
        for n in (100001, 3, 0, 257):
            if n == 0:
                assert g.ecdsa_verify_batch(np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8),
                                            np.zeros((0, 32), np.uint8)).size == 0
                continue
            arrs = damaged_batch(eng, n, max(n // 13, 2), 900 + n)
            ref = eng.ecdsa_verify_batch(*arrs)
            m = min(n, 2048)
            assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(*(a[:m] for a in arrs), nthreads=os.cpu_count() or 1))
            for mode in (S.KEYS_AUTO, S.KEYS_OFF):
                g.set_key_grouping(mode)
                assert np.array_equal(g.ecdsa_verify_batch(*arrs), ref)
            g.set_key_grouping(S.KEYS_AUTO)
            st = g.member_stats()
            assert sum(x["n"] for x in st) == n and st[0]["first"] == 0 and [x["device"] for x in st] == [0, 0]
            assert n < 2 or st[1]["first"] == st[0]["n"]
        # pipelined: three group batches submitted before the first wait (the third submit blocks until the first is done)
        batches = [damaged_batch(eng, 60000 + 1000 * k, 500, 950 + k) for k in range(3)]
        refs = [eng.ecdsa_verify_batch(*b) for b in batches]
        tickets = [g.ecdsa_verify_batch_submit(*b) for b in batches]
        for t, ref in zip(tickets, refs):
            assert np.array_equal(t.wait(), ref)
        # low-s rule through the group
        a = batches[0]
        assert np.array_equal(g.ecdsa_verify_batch(*a, reject_malleable=True), eng.ecdsa_verify_batch(*a, reject_malleable=True))
        with pytest.raises(S.EngineError):
            g._wait(10 ** 6)
        # encoded items across the group: shared blobs, item ranges per member; DER, compact and recoverable forms
        n = 20001
        arrs = damaged_batch(eng, n, 300, 991)
        ref = eng.ecdsa_verify_batch(*arrs)
        pubs = [b"\x04" + bytes(q) if i % 3 else oracle.point_compressed(b"\x04" + bytes(q)) if oracle.point_on_curve_xy(bytes(q[:32]), bytes(q[32:])) else b"\x04" + bytes(q)
                for i, q in enumerate(arrs[0])]
        digs = [bytes(d) for d in arrs[1]]
        compact = [bytes(r) + bytes(s_) for r, s_ in zip(arrs[2], arrs[3])]
        assert np.array_equal(g.ecdsa_verify_encoded_batch_submit(pubs, digs, compact, encoding=S.ENCODING_COMPACT).wait(), ref)
        assert np.array_equal(eng.ecdsa_verify_encoded_batch(pubs, digs, compact, encoding=S.ENCODING_COMPACT), ref)
        rec = [c + bytes([i & 3]) for i, c in enumerate(compact)]
        assert np.array_equal(g.ecdsa_verify_encoded_batch_submit(pubs, digs, rec, encoding=S.ENCODING_COMPACT_RECOVERABLE).wait(),
                              eng.ecdsa_verify_encoded_batch(pubs, digs, rec, encoding=S.ENCODING_COMPACT_RECOVERABLE))
    finally:
        g.close()
    with pytest.raises(S.EngineError):
        S.Group([S.device_count()])            # no such device


def test_group_whole_batch_forms_two_members_one_device(eng, oracle):
    """BASELINE configs 3 and 4 across a group (two members on device 0): the multiscalar multiplication of sharded terms equals
    the single-context call's and the known answer (sum k_i d_i) G, for sizes around the shard rounding, with identity and repeated
    points; the BIP-340 whole-batch check accepts valid batches (fixed-length and ragged messages) and rejects them with one bad
    signature in the first member's shard, in the last member's, or at the shard border."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    rnd = random.Random(12)
    g = S.Group([0, 0])
    try:
        for n in (0, 1, 255, 256, 257, 700, 5000):
            ds = [rnd.randrange(1, R.N) for _ in range(n)]
            ks_ = [rnd.randrange(0, R.N) for _ in range(n)]
            pts = eng.scalar_base_mult_batch(np.frombuffer(b"".join(b32(d) for d in ds), np.uint8).reshape(-1, 32)) if n else np.zeros((0, 65), np.uint8)
            pts = np.array(pts)
            if n > 10:
                pts[3] = 0                                   # the identity among the points
                pts[7] = pts[8]                              # a repeated point
                ds[3], ds[7] = 0, ds[8]
            scal = np.frombuffer(b"".join(b32(k) for k in ks_), np.uint8).reshape(-1, 32) if n else np.zeros((0, 32), np.uint8)
            got = g.multi_scalar_mult(scal, pts)
            assert got == eng.multi_scalar_mult(scal, pts), n
            tot = sum(k * d for k, d in zip(ks_, ds)) % R.N
            assert got == (oracle.scalar_base_mult_vartime(b32(tot)) if tot else bytes(65)), n
        for n, ml in ((600, 32), (513, None), (2, 32), (0, 32)):
            if n == 0:
                assert g.schnorr_batch_verify_rlc(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8))
                continue
            if ml is None:                                   # ragged messages
                dd = [rnd.randrange(1, R.N) for _ in range(n)]
                msgs = [rnd.randbytes(rnd.choice([0, 1, 31, 33, 64, 100])) for _ in range(n)]
                sigs = [R.schnorr_sign(d, m, rnd.randbytes(32)) for d, m in zip(dd, msgs)]
                pk = np.frombuffer(b"".join(b32(R.mul(d, R.G)[0]) for d in dd), np.uint8).reshape(-1, 32)
                sig = np.frombuffer(b"".join(sigs), np.uint8).reshape(-1, 64).copy()
            else:
                pk, msgs, sig = (np.array(a) for a in synth_schnorr_batch(eng, n, max(n // 7, 1), seed=n))
            assert g.schnorr_batch_verify_rlc(pk, msgs, sig) and eng.schnorr_batch_verify_rlc(pk, msgs, sig)
            per = (((n + 1) // 2) + 255) // 256 * 256        # the members' shards: [0, per) and [per, n)
            for bad in sorted({0, min(per, n) - 1, min(per, n - 1), n - 1}):
                sig[bad, 50] ^= 1
                assert not g.schnorr_batch_verify_rlc(pk, msgs, sig), (n, bad)
                sig[bad, 50] ^= 1
    finally:
        g.close()


@pytest.mark.parametrize("layout", [1, 2, 3])
def test_group_keyset_two_members_one_device(eng, oracle, layout):
    """s2k_group_keyset_*: the key set on both members, batches of (key index, digest, r, s) sharded across them; verdicts
    equal the single-context key-set call's, the batch verifier's on the expanded keys and the oracle's - ragged sizes,
    sizes below the member count, an empty batch, three batches in flight, the low-s rule, a set of another group."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    nk, total = 700, 150000
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, total, nk, seed=23))
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    keys = keys.copy()
    keys[9, 1] ^= 0x10                                                 # no public key
    kidx = inv.reshape(-1).astype(np.uint32)
    rng = np.random.default_rng(5)
    for i in range(0, total, 8):
        a = (dig, r, s)[int(rng.integers(0, 3))]
        a[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    kidx[5] = len(keys)
    full = np.zeros((total, 64), np.uint8)
    inside = kidx < len(keys)
    full[inside] = keys[kidx[inside]]
    ks1 = eng.keyset_create(keys, layout)
    ref = eng.ecdsa_verify_batch_keyset(ks1, kidx, dig, r, s)
    assert np.array_equal(ref, eng.ecdsa_verify_batch(full, dig, r, s))
    m = 2048
    assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(full[:m], dig[:m], r[:m], s[:m], nthreads=os.cpu_count() or 1))
    assert 0 < int(ref.sum()) < total and not ref[5]
    g = S.Group([0, 0])
    try:
        gks = g.keyset_create(keys, layout)
        assert len(gks) == len(keys) and gks.layout() == layout and gks.device_bytes() == ks1.device_bytes()
        for lo, n in ((0, total), (7, 100001), (11, 1), (0, 0), (300, 513)):
            got = g.ecdsa_verify_batch_keyset(gks, kidx[lo:lo + n], dig[lo:lo + n], r[lo:lo + n], s[lo:lo + n])
            assert np.array_equal(got, ref[lo:lo + n]), (lo, n)
            if n > 1:
                st = g.member_stats()
                assert sum(x["n"] for x in st) == n and st[1]["first"] == st[0]["n"]
        cuts = [(0, 50000), (50000, 50001), (100001, 49999)]
        tickets = [g.ecdsa_verify_batch_keyset_submit(gks, kidx[a:a + n], dig[a:a + n], r[a:a + n], s[a:a + n]) for a, n in cuts]
        plain = g.ecdsa_verify_batch_submit(full[:30000], dig[:30000], r[:30000], s[:30000])       # and a plain one among them
        for t, (a, n) in zip(tickets, cuts):
            assert np.array_equal(t.wait(), ref[a:a + n])
        assert np.array_equal(plain.wait(), ref[:30000])
        assert np.array_equal(g.ecdsa_verify_batch_keyset(gks, kidx[:20000], dig[:20000], r[:20000], s[:20000], reject_malleable=True),
                              eng.ecdsa_verify_batch_keyset(ks1, kidx[:20000], dig[:20000], r[:20000], s[:20000], reject_malleable=True))
        g2 = S.Group([0])
        try:
            with pytest.raises(S.EngineError):
                g2.ecdsa_verify_batch_keyset(gks, kidx[:100], dig[:100], r[:100], s[:100])
        finally:
            g2.close()
        gks.close()
    finally:
        g.close()
        ks1.close()


# ---- EncodingCompactRecoverable -------------------------------------------------------------------------------------
@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_compact_recoverable(eng, oracle, fn):
    """secec/wycheproof_test.go:417-438 builds the recoverable forms of every parseable case with all four recovery
    ids and asks whether one of them gives the signer's key.  Here every (case, id) pair is one item of
    s2k_ecdsa_verify_encoded_batch(S2K_ENCODING_COMPACT_RECOVERABLE) = PublicKey.Verify with EncodingCompactRecoverable
    (ecdsa.go:204-205,220-226): valid iff RecoverPublicKey succeeds and the key equals the supplied one (oracle's recover
    + equality); per case, some id verifies iff the case is valid."""
    import secp256k1_voi_amd as S
    d = load_golden(fn)
    pubs, digs, sigs, exp, meta = [], [], [], [], []
    for c in d["cases"]:
        rs = oracle.parse_asn1_signature(H(c["sig"]))
        if rs is None:
            continue
        for rid in range(5):                      # 4 is no recovery id
            pubs.append(H(c["pub"]))
            digs.append(H(c["digest"]))
            sigs.append(rs[0] + rs[1] + bytes([rid]))
            q = oracle.ecdsa_recover(H(c["digest"])[:32], rs[0], rs[1], rid)
            exp.append(int(q is not None and q == H(c["pub"])))
            meta.append((c, rid))
    got = eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE)
    assert got.tolist() == exp
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE, force_complete=True).tolist() == exp
    by_case = {}
    for (c, rid), v in zip(meta, got):
        by_case[c["tcId"]] = by_case.get(c["tcId"], 0) | int(v)
    for c in d["cases"]:
        if c["tcId"] in by_case:
            assert by_case[c["tcId"]] == int(c["valid"]), c["tcId"]
    # compressed keys name the same key
    comp = [oracle.point_compressed(p) for p in pubs]
    assert eng.ecdsa_verify_encoded_batch(comp, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE).tolist() == exp
    # opts.Hash = SHA-256 (ecdsa.go:184-188)
    got32 = eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE, digest_len=32).tolist()
    assert got32 == (exp if fn.endswith("sha256.json") else [0] * len(exp))
    # submit form
    assert eng.ecdsa_verify_encoded_batch_submit(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE).wait().tolist() == exp


def test_compact_recoverable_options_and_edges(eng, oracle):
    """Random signatures with their true recovery id (found with the oracle), then: a wrong id, the signature of another
    key, wrong lengths (64, 66 bytes), V = 4 and 255, r / s out of range, high s with and without RejectMalleable,
    malformed supplied keys.  Expected values from the oracle's recover + byte equality."""
    import secp256k1_voi_amd as S
    rnd = random.Random(404)
    pubs, digs, sigs, exp, exp_low = [], [], [], [], []

    def add(pub, dg, sig):
        pubs.append(pub); digs.append(dg); sigs.append(sig)
        ok, ok_low = 0, 0
        if len(sig) == 65 and len(dg) >= 32:
            r, s, v = sig[:32], sig[32:64], sig[64]
            ri, si = int.from_bytes(r, "big"), int.from_bytes(s, "big")
            if 0 < ri < R.N and 0 < si < R.N:
                q = oracle.ecdsa_recover(dg[:32], r, s, v) if v < 4 else None
                ok = int(q is not None and len(pub) == 65 and q == pub)
                if q is not None and len(pub) == 33:
                    ok = int(oracle.point_compressed(q) == pub)
                ok_low = ok if si <= R.N // 2 else 0
        exp.append(ok); exp_low.append(ok_low)

    keys = []
    for _ in range(8):
        d = rnd.randrange(1, R.N)
        keys.append((d, oracle.scalar_base_mult_vartime(b32(d))))
    for i in range(300):
        d, q = keys[i % len(keys)]
        dg = rnd.randbytes(32)
        k = rnd.randrange(1, R.N)
        r, s = R.ecdsa_sign(d, dg, k)
        if i % 3 == 0:
            s = R.N - s                                           # high s: still recovers (with the other parity)
        rid = next(v for v in range(4) if oracle.ecdsa_recover(dg, b32(r), b32(s), v) == q)
        sig = b32(r) + b32(s) + bytes([rid])
        kind = i % 10
        if kind == 0:
            add(q, dg, sig)
        elif kind == 1:
            add(q, dg, sig[:64] + bytes([rid ^ 1]))                # the other y: another key
        elif kind == 2:
            add(keys[(i + 1) % len(keys)][1], dg, sig)             # somebody else's key
        elif kind == 3:
            add(q, dg, rnd.choice([sig[:64], sig + b"\0", b""]))   # wrong length
        elif kind == 4:
            add(q, dg, sig[:64] + bytes([rnd.choice([4, 5, 255])]))
        elif kind == 5:
            add(q, dg, rnd.choice([b32(0), b32(R.N), b32(R.N + 5)]) + sig[32:])   # r out of range
        elif kind == 6:
            add(q, dg, sig[:32] + rnd.choice([b32(0), b32(R.N)]) + sig[64:])      # s out of range
        elif kind == 7:
            add(oracle.point_compressed(q), dg, sig)               # compressed form of the same key
        elif kind == 8:
            bad = bytearray(q); bad[64] ^= 1
            add(rnd.choice([bytes(bad), b"\x05" + q[1:], q[:64], b"\x00"]), dg, sig)   # not a key NewPublicKey accepts
        else:
            add(q, rnd.randbytes(32), sig)                         # another digest: recovers some other key
    got = eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE)
    assert got.tolist() == exp
    assert 0 < sum(exp) < len(exp)
    got_low = eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, encoding=S.ENCODING_COMPACT_RECOVERABLE, reject_malleable=True)
    assert got_low.tolist() == exp_low and exp_low != exp
    # the other two encodings refuse 65-byte signatures; BIP-0066 goes with ASN.1 only
    assert not eng.ecdsa_verify_encoded_batch(pubs[:10], digs[:10], [s_ for s_ in sigs[:10]], encoding=S.ENCODING_COMPACT)[
        [len(s_) == 65 for s_ in sigs[:10]]].any()
    with pytest.raises(S.EngineError):
        eng.ecdsa_verify_encoded_batch(pubs[:4], digs[:4], sigs[:4], encoding=S.ENCODING_COMPACT_RECOVERABLE, bip0066=True)
    with pytest.raises(S.EngineError):
        eng.ecdsa_verify_encoded_batch(pubs[:4], digs[:4], sigs[:4], encoding=3)


# ---- the parse kernel's two ways to its bytes ---------------------------------------------------------------------------
def test_encoded_parse_staging_paths(eng, oracle):
    """k_parse_encoded copies a workgroup's stretch of every blob into LDS and parses from there; a stretch that does not fit
    (digests, keys or signatures far longer than valid ones) is read from global memory instead.  3 000 items - DER signatures
    with leading-zero and short integers, compressed and uncompressed keys at every alignment, damaged items - in five
    shapes: as they are; with 64-byte digests (staged); with 200-byte digests (not staged: the leftmost 32 bytes count);
    with one 5 000-byte signature and one 5 000-byte key in the middle of a workgroup (that workgroup unstaged, the items
    false); compact and recoverable forms.  Every verdict against per-item host parsing + the raw verifier, a sample
    against the oracle."""
    import secp256k1_voi_amd as S
    n = 3000
    arrs = damaged_batch(eng, n, 200, 4141)
    pub, dig, r, s = arrs
    for i in range(0, n, 50):                            # short scalars: leading zero bytes are stripped in DER
        r[i, :3] = 0
        s[i, 0] = 0
    raw = eng.ecdsa_verify_batch(pub, dig, r, s)
    m = 1024
    assert np.array_equal(raw[:m], oracle.ecdsa_verify_batch(pub[:m], dig[:m], r[:m], s[:m], nthreads=os.cpu_count() or 1))

    def der(rb, sb):
        body = b""
        for x in (int.from_bytes(bytes(rb), "big"), int.from_bytes(bytes(sb), "big")):
            bb = x.to_bytes((x.bit_length() + 8) // 8 or 1, "big")
            body += b"\x02" + bytes([len(bb)]) + bb
        return b"\x30" + bytes([len(body)]) + body
    sigs = [der(r[i], s[i]) for i in range(n)]
    pubs = []
    for i in range(n):
        q = bytes(pub[i])
        full = b"\x04" + q
        pubs.append(oracle.point_compressed(full) if (i % 3 == 0 and oracle.point_on_curve_xy(q[:32], q[32:])) else full)
    digs = [bytes(d) for d in dig]
    exp = raw.copy()
    for i in range(n):                                   # what the host parser says: a zero or out-of-range scalar is errInvalidScalar
        if oracle.parse_asn1_signature(sigs[i]) is None:
            exp[i] = 0
    assert 0 < int(exp.sum()) < n
    assert np.array_equal(eng.ecdsa_verify_encoded_batch(pubs, digs, sigs), exp)
    # longer digests: the leftmost 32 bytes are the scalar (hashToScalar, ecdsa.go:477-486)
    for extra in (32, 168):
        assert np.array_equal(eng.ecdsa_verify_encoded_batch(pubs, [d + bytes(extra) for d in digs], sigs), exp), extra
        assert not eng.ecdsa_verify_encoded_batch(pubs[:600], [d + bytes(extra) for d in digs[:600]], sigs[:600], digest_len=32).any()
    # one oversized signature and one oversized key inside a workgroup: those two false, their neighbours untouched
    sig2, pub2, exp2 = list(sigs), list(pubs), exp.copy()
    sig2[700] = b"\x30\x82\x13\x84" + bytes(4996)
    pub2[1800] = b"\x04" + bytes(4999)
    exp2[700] = exp2[1800] = 0
    assert np.array_equal(eng.ecdsa_verify_encoded_batch(pub2, digs, sig2), exp2)
    # compact and recoverable forms of the same items
    compact = [bytes(r[i]) + bytes(s[i]) for i in range(n)]
    exp_c = raw                                          # (ParseCompactSignature's range checks are the raw verifier's)
    assert np.array_equal(eng.ecdsa_verify_encoded_batch(pubs, digs, compact, encoding=S.ENCODING_COMPACT), exp_c)
    rec = [c + bytes([i & 3]) for i, c in enumerate(compact)]
    got_r = eng.ecdsa_verify_encoded_batch(pubs, digs, rec, encoding=S.ENCODING_COMPACT_RECOVERABLE)
    assert np.array_equal(got_r, eng.ecdsa_verify_encoded_batch(pubs, digs, rec, encoding=S.ENCODING_COMPACT_RECOVERABLE, force_complete=True))
    assert not got_r[exp_c == 0].any() and 0 < int(got_r.sum()) < int(exp_c.sum())      # a valid item verifies under ONE recovery id
    # through the ticket form too
    assert np.array_equal(eng.ecdsa_verify_encoded_batch_submit(pub2, digs, sig2).wait(), exp2)


# ---- the table buffer degrades instead of failing -------------------------------------------------------
def test_table_buffer_allocation_failure_degrades(oracle):
    """s2k_set_table_memory_budgets(0, limit) makes s2k_internal_key_reserve treat larger table buffers as unobtainable: the cap is
    halved until the buffer fits (fewer, longer groups get tables), below 1024 tables the batch is verified without tables
    - the verification call never fails, the verdicts never change."""
    import secp256k1_voi_amd as S
    eng = S.Engine(0)                                  # a context of its own: the cap it finds stays with it
    n = 1 << 17
    arrs = damaged_batch(eng, n, n // 32, 555)         # 32 signatures per key
    ref = eng.ecdsa_verify_batch(*arrs)
    st0 = eng.key_grouping_stats()
    assert st0["tables"] == n // 32 and st0["keyed"] + st0["general"] == n
    m = 2048
    assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(*(a[:m] for a in arrs), nthreads=os.cpu_count() or 1))
    try:
        # default plan: n / 4 = 32768 tables of 9 KiB = 302 MB.  100 MB: the cap goes 32768 -> 16384 -> 8192 (75 MB),
        # threshold 16 signatures per key: all 4096 keys still get their tables
        eng2 = S.Engine(0)
        S.load_library().s2k_set_table_memory_budgets(0, 100 << 20)
        assert np.array_equal(eng2.ecdsa_verify_batch(*arrs), ref)
        st = eng2.key_grouping_stats()
        assert st["tables"] == n // 32 and st["keyed"] + st["general"] == n
        assert eng2.device_bytes(n) < eng.device_bytes(n)                       # the memory query follows the cap
        # 20 MB: 2048 tables, threshold 64 per key: no key qualifies, everything on the general ladder
        eng3 = S.Engine(0)
        S.load_library().s2k_set_table_memory_budgets(0, 20 << 20)
        assert np.array_equal(eng3.ecdsa_verify_batch(*arrs), ref)
        st = eng3.key_grouping_stats()
        assert st["tables"] == 0 and st["general"] == n
        # 1 MB: not even 1024 tables: verified without the grouping at all
        eng4 = S.Engine(0)
        S.load_library().s2k_set_table_memory_budgets(0, 1 << 20)
        assert np.array_equal(eng4.ecdsa_verify_batch(*arrs), ref)
        st = eng4.key_grouping_stats()
        assert st["tables"] == 0 and st["keyed"] == 0
        # BIP-340 takes the same way out
        from secp256k1_voi_amd.synth import synth_schnorr_batch
        pk, msgs, sig = synth_schnorr_batch(eng4, 4096, 64, seed=9)
        assert eng4.schnorr_verify_batch(pk, msgs, sig).all()
    finally:
        S.load_library().s2k_set_table_memory_budgets(0, 0)
    # the limit gone and the setting renewed: tables again
    eng4.set_key_grouping(S.KEYS_AUTO)
    assert np.array_equal(eng4.ecdsa_verify_batch(*arrs), ref) and eng4.key_grouping_stats()["tables"] == n // 32


# ---- the reference's golden vectors through the key-set ladders ---------------------------------------------------------------
@pytest.mark.parametrize("layout", [1, 2, 3, 4])
@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_through_key_sets(eng, oracle, fn, layout):
    """The Wycheproof ECDSA cases (secec/wycheproof_test.go:317-334: edge public keys, Shamir and modular-inverse edge cases, point
    duplication, x(R) >= n, ...) with their public keys as a KEY SET and every case naming its key by index: the verdicts of the
    chunk ladder and of the 4-, 5- and 6-bit joint ladders must be the expected ones, case by case.  Then the RFC 6979 signatures
    and the reused-nonce pair of the reference's tests the same way."""
    d = load_golden(fn)
    items, exp = [], []
    for c in d["cases"]:
        rs = oracle.parse_asn1_signature(H(c["sig"]))
        if rs is None:
            continue
        items.append((H(c["pub"])[1:], H(c["digest"])[:32], rs[0], rs[1]))
        exp.append(int(c["valid"]))
    keys, inv = np.unique(np.frombuffer(b"".join(i[0] for i in items), np.uint8).reshape(-1, 64), axis=0, return_inverse=True)
    ks = eng.keyset_create(keys, layout)
    assert ks.layout() == layout
    got = eng.ecdsa_verify_batch_keyset(ks, inv.reshape(-1).astype(np.uint32), [i[1] for i in items], [i[2] for i in items], [i[3] for i in items])
    assert got.tolist() == exp
    assert sum(exp) == {"wycheproof_ecdsa_sha256.json": 164, "wycheproof_ecdsa_sha512.json": 233}[fn]
    ks.close()
    if fn.endswith("sha256.json"):
        k = load_golden("kats.json")["reused_k_pairs"]
        cases = load_golden("rfc6979.json")["cases"]
        pubs = [oracle.scalar_base_mult_vartime(H(c["private"]))[1:] for c in cases] + [oracle.scalar_base_mult_vartime(H(k["private"]))[1:]] * 2
        digs = [H(c["digest"]) for c in cases] + [H(s_["digest"]) for s_ in k["sigs"]]
        rs_ = [oracle.parse_asn1_signature(H(c["sig"])) for c in cases]
        rr = [x[0] for x in rs_] + [H(s_["r"]) for s_ in k["sigs"]]
        ss = [x[1] for x in rs_] + [H(s_["s"]) for s_ in k["sigs"]]
        keys2, inv2 = np.unique(np.frombuffer(b"".join(pubs), np.uint8).reshape(-1, 64), axis=0, return_inverse=True)
        ks2 = eng.keyset_create(keys2, layout)
        kidx = inv2.reshape(-1).astype(np.uint32)
        assert eng.ecdsa_verify_batch_keyset(ks2, kidx, digs, rr, ss).all()
        m = len(cases)                                   # (RFC 6979 signatures are low-s; the reused-nonce pair need not be)
        assert eng.ecdsa_verify_batch_keyset(ks2, kidx[:m], digs[:m], rr[:m], ss[:m], reject_malleable=True).all()
        assert not eng.ecdsa_verify_batch_keyset(ks2, kidx[:m], digs[1:m] + digs[:1], rr[:m], ss[:m]).any()
        ks2.close()


# ---- BIP-340 over key sets ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [1, 2, 3, 4])
def test_schnorr_over_key_sets(eng, oracle, layout):
    """s2k_schnorr_verify_batch_keyset: signatures name their key by index in a set of X || Y keys; BIP-340 verifies under the x
    coordinate, i.e. under lift_x(X) - the set holds HALF of its keys with odd Y on purpose.  Verdicts must equal
    s2k_schnorr_verify_batch on the expanded x-only keys and the oracle's: random valid and damaged signatures with messages of
    many lengths, keys that are no curve points, indices outside the set; then the official BIP-340 vectors with their keys as
    the set; then 2^17 signatures of 2^11 keys."""
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    rnd = random.Random(77)
    nk, n = 40, 1500
    ds = [rnd.randrange(1, R.N) for _ in range(nk)]
    pts = [R.mul(dd, R.G) for dd in ds]
    keys = np.zeros((nk + 2, 64), np.uint8)
    for k, P in enumerate(pts):
        y = P[1] if k % 2 == 0 else R.P - P[1]           # every other key: the point with the OTHER y (same x-only key)
        keys[k] = np.frombuffer(b32(P[0]) + b32(y), np.uint8)
    keys[nk] = np.frombuffer(b32(pts[0][0]) + b32((pts[0][1] + 1) % R.P), np.uint8)     # not on the curve
    keys[nk + 1] = 0xFF                                                                   # coordinates >= p
    pk, msg, sig, kidx = [], [], [], []
    for i in range(n):
        k = i % nk
        m = rnd.randbytes(rnd.choice([0, 1, 31, 32, 33, 55, 56, 63, 64, 65, 119, 120, 200]))
        s_ = R.schnorr_sign(ds[k], m, rnd.randbytes(32))
        kind = i % 7
        if kind == 1:
            s_ = s_[:32] + b32((int.from_bytes(s_[32:], "big") + 1) % R.N)
        elif kind == 2:
            m = m + b"x"
        elif kind == 3 and i % 21 == 3:
            s_ = b32(R.P + 5) + s_[32:]                   # r >= p
        elif kind == 4 and i % 28 == 4:
            k = nk + (i % 2)                              # a key of the set that is no public key
        elif kind == 5 and i % 35 == 5:
            k = nk + 2 + (i % 3)                          # an index outside the set
        kidx.append(k); msg.append(m); sig.append(s_)
        pk.append(bytes(keys[k][:32]) if k < nk + 2 else bytes(32))
    ks = eng.keyset_create(keys, layout)
    got = eng.schnorr_verify_batch_keyset(ks, np.array(kidx, np.uint32), msg, sig)
    ref = eng.schnorr_verify_batch(pk, msg, sig)
    out_of_set = np.array([k >= nk for k in kidx])
    assert not got[out_of_set].any()
    assert np.array_equal(got[~out_of_set], ref[~out_of_set])
    exp = np.array([int(oracle.schnorr_verify(p_, m_, s_) == 1) for p_, m_, s_ in zip(pk[:400], msg[:400], sig[:400])], np.uint8)
    assert np.array_equal(got[:400][~out_of_set[:400]], exp[~out_of_set[:400]])
    assert 0.5 * n < int(got.sum()) < n
    ks.close()
    # the official vectors: their x-only keys lifted (odd y for every other one); keys that do not lift stand in as off-curve points
    d = load_golden("bip340.json")
    vk = sorted({c["public_key"] for c in d["cases"]})
    vkeys = np.zeros((len(vk), 64), np.uint8)
    for j, hx in enumerate(vk):
        x = int(hx, 16)
        pt = R.lift_x(x, j % 2)                          # (None: x >= p or no point has this x)
        vkeys[j] = np.frombuffer(H(hx) + (b32(pt[1]) if pt else bytes(32)), np.uint8)
    ks = eng.keyset_create(vkeys, layout)
    vidx = np.array([vk.index(c["public_key"]) for c in d["cases"]], np.uint32)
    assert eng.schnorr_verify_batch_keyset(ks, vidx, [H(c["message"]) for c in d["cases"]], [H(c["signature"]) for c in d["cases"]]).tolist() == \
        [int(c["valid"]) for c in d["cases"]]
    ks.close()
    # size: 2^17 signatures of 2^11 keys, every ninth damaged
    big, nkb = 1 << 17, 1 << 11
    pkb, msgb, sigb = (np.array(a) for a in synth_schnorr_batch(eng, big, nkb, seed=31))
    sigb[::9, 40] ^= 2
    xs, inv = np.unique(pkb, axis=0, return_inverse=True)
    pts65, okd = eng.point_decode_batch(np.concatenate([np.full((len(xs), 1), 2, np.uint8), xs], axis=1), 33)
    assert okd.all()
    kb = np.ascontiguousarray(pts65[:, 1:])
    kb[1::2, 32:] = np.frombuffer(b"".join(b32(R.P - int.from_bytes(bytes(y_), "big")) for y_ in kb[1::2, 32:]), np.uint8).reshape(-1, 32)
    ks = eng.keyset_create(kb, layout)
    gotb = eng.schnorr_verify_batch_keyset(ks, inv.reshape(-1).astype(np.uint32), msgb, sigb)
    assert np.array_equal(gotb, eng.schnorr_verify_batch(pkb, msgb, sigb)) and int(gotb.sum()) == big - len(range(0, big, 9))
    # the ticket form: three batches in flight (fixed-length messages; ragged ones as a list), mixed with an ECDSA ticket
    kx = inv.reshape(-1).astype(np.uint32)
    cuts = [(0, 50000), (50000, 40001), (90001, big - 90001)]
    tk = [eng.schnorr_verify_batch_keyset_submit(ks, kx[a:a + c], np.ascontiguousarray(msgb[a:a + c]), np.ascontiguousarray(sigb[a:a + c])) for a, c in cuts]
    tl = eng.schnorr_verify_batch_keyset_submit(ks, kx[:300], [bytes(m_) for m_ in msgb[:300]], sigb[:300])
    for t_, (a, c) in zip(tk, cuts):
        assert np.array_equal(t_.wait(), gotb[a:a + c])
    assert np.array_equal(tl.wait(), gotb[:300])
    eng.wait_all()
    ks.close()


# ---- small and odd key sets ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [0, 1, 2, 3, 4])
def test_keyset_small_and_duplicate_keys(eng, oracle, layout):
    """A set of ONE key; a set that lists the same key several times (every copy is a key of its own: same verdicts under any of its
    indices); a set whose only key is no curve point; batches of one signature and of a few thousand under one key (every lane of
    every wave on the same table).  All layouts, AUTO included."""
    from secp256k1_voi_amd.synth import synth_batch
    n = 5000
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, n, 1, seed=61))
    s[::5, 9] ^= 0x20
    ref = eng.ecdsa_verify_batch(pub, dig, r, s)
    assert np.array_equal(ref[:512], oracle.ecdsa_verify_batch(pub[:512], dig[:512], r[:512], s[:512], nthreads=os.cpu_count() or 1))
    one = eng.keyset_create(pub[:1], layout)
    zeros = np.zeros(n, np.uint32)
    assert np.array_equal(eng.ecdsa_verify_batch_keyset(one, zeros, dig, r, s), ref)
    assert np.array_equal(eng.ecdsa_verify_batch_keyset(one, zeros[:1], dig[:1], r[:1], s[:1]), ref[:1])
    one.close()
    other = np.array(synth_batch(eng, 1, 1, seed=62)[0])
    dup = eng.keyset_create(np.concatenate([pub[:1], other, pub[:1], pub[:1]]), layout)       # copies of the key at 0, 2, 3
    kidx = np.array([(0, 2, 3, 1)[i % 4] for i in range(n)], np.uint32)
    got = eng.ecdsa_verify_batch_keyset(dup, kidx, dig, r, s)
    assert np.array_equal(got[kidx != 1], ref[kidx != 1]) and not got[kidx == 1].any()
    dup.close()
    bad = pub[:1].copy()
    bad[0, 63] ^= 1
    nokey = eng.keyset_create(bad, layout)
    assert not nokey.valid_keys().any() and not eng.ecdsa_verify_batch_keyset(nokey, zeros, dig, r, s).any()
    nokey.close()


# ---- the layout a key set gets when memory is short -------------------------------------------------------------------------
def test_keyset_layout_choice_under_memory_limits(eng):
    """s2k_set_table_memory_budgets(free, 0) makes s2k_keyset_create_ex count only that much room for joint tables:
    S2K_KEYSET_AUTO takes 5-bit joint tables when they fit in half of it, else 4-bit ones, else chunk tables, and the set verifies
    the same in each; an explicit joint layout that does not fit is an error (and leaves nothing behind: the next set is built)."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    nk, n = 400, 6000
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, n, nk, seed=88))
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    kidx = inv.reshape(-1).astype(np.uint32)
    s[::7, 4] ^= 1
    ref = eng.ecdsa_verify_batch(pub, dig, r, s)
    j5 = nk * (26 * 512 + 2) * 64 + nk * 26 * 16 * 80        # joint tables + build scratch at 5 bits
    j4 = nk * 32 * 128 * 80
    try:
        for free, want in ((None, S.KEYSET_JOINT5), (2 * j5 + 4096, S.KEYSET_JOINT5), (2 * j5 - 4096, S.KEYSET_JOINT), (2 * j4 - 4096, S.KEYSET_CHUNKS), (1, S.KEYSET_CHUNKS)):
            S.load_library().s2k_set_table_memory_budgets(free or 0, 0)
            ks = eng.keyset_create(keys)
            assert ks.layout() == want, (free, ks.layout())
            assert np.array_equal(eng.ecdsa_verify_batch_keyset(ks, kidx, dig, r, s), ref)
            ks.close()
        S.load_library().s2k_set_table_memory_budgets(j4 - 1, 0)
        for layout in (S.KEYSET_JOINT, S.KEYSET_JOINT5, S.KEYSET_JOINT6):
            with pytest.raises(S.EngineError):
                eng.keyset_create(keys, layout)
        ks = eng.keyset_create(keys, S.KEYSET_CHUNKS)
        assert np.array_equal(eng.ecdsa_verify_batch_keyset(ks, kidx, dig, r, s), ref)
        ks.close()
    finally:
        S.load_library().s2k_set_table_memory_budgets(0, 0)


# ---- key sets through submit / wait ------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [1, 2, 3, 4])      # S2K_KEYSET_CHUNKS, S2K_KEYSET_JOINT, S2K_KEYSET_JOINT5, S2K_KEYSET_JOINT6
def test_keyset_submit_wait(eng, oracle, layout):
    """s2k_ecdsa_verify_batch_keyset_submit: six batches (ragged sizes, one empty, pageable and page-locked buffers, indices
    outside the set, keys that are no public keys) with four in flight, mixed with tickets of the plain submit: every ticket's
    verdicts equal the synchronous key-set call's, the batch verifier's on the expanded keys and the oracle's; a key set of
    another context is refused."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    nk = 500
    sizes = [30011, 0, 70000, 257, 131072, 5000]
    rng = np.random.default_rng(91)
    total = sum(sizes)
    pub, dig, r, s = (np.array(a) for a in synth_batch(eng, total, nk, seed=17))     # one synthetic batch, cut into the six
    keys, inv = np.unique(pub, axis=0, return_inverse=True)
    keys = keys.copy()
    kidx_all = inv.reshape(-1).astype(np.uint32)
    keys[3, 40] ^= 4                                                   # no public key
    ks = eng.keyset_create(keys, layout)
    for i in range(0, total, 9):
        a = (dig, r, s)[int(rng.integers(0, 3))]
        a[i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
    batches, lo = [], 0
    for n in sizes:
        kidx = kidx_all[lo:lo + n].copy()
        if n > 20:
            kidx[7] = len(keys)
            kidx[13] = 0xFFFFFFFF
        full = np.zeros((n, 64), np.uint8)
        inside = kidx < len(keys)
        full[inside] = keys[kidx[inside]]
        batches.append((kidx, dig[lo:lo + n].copy(), r[lo:lo + n].copy(), s[lo:lo + n].copy(), full))
        lo += n
    refs = [eng.ecdsa_verify_batch_keyset(ks, b[0], b[1], b[2], b[3]) if len(b[0]) else np.zeros(0, np.uint8) for b in batches]
    for b, ref in zip(batches, refs):
        if len(b[0]):
            assert np.array_equal(ref, eng.ecdsa_verify_batch(b[4], b[1], b[2], b[3]))
            m = min(len(b[0]), 1500)
            assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(b[4][:m], b[1][:m], b[2][:m], b[3][:m], nthreads=os.cpu_count() or 1))
            assert 0 < int(ref.sum()) < len(ref)
    src = []
    for bi, b in enumerate(batches):                                   # every other batch from page-locked memory
        if bi % 2 == 0 and len(b[0]):
            pk = [S.pinned_array(a.shape, a.dtype) for a in b[:4]]
            for d, a in zip(pk, b[:4]):
                d[...] = a
            src.append(pk)
        else:
            src.append(list(b[:4]))
    tickets = []
    for bi, q in enumerate(src):
        tickets.append(eng.ecdsa_verify_batch_keyset_submit(ks, *q))
        if bi == 2:                                                    # a ticket of the plain submit between them
            plain = eng.ecdsa_verify_batch_submit(batches[0][4], *batches[0][1:4])
    assert np.array_equal(plain.wait(), refs[0])
    for tk, ref in zip(tickets, refs):
        assert np.array_equal(tk.wait(), ref)
    # low-s rule through the flag
    b = batches[0]
    t1 = eng.ecdsa_verify_batch_keyset_submit(ks, *b[:4], reject_malleable=True)
    assert np.array_equal(t1.wait(), eng.ecdsa_verify_batch_keyset(ks, *b[:4], reject_malleable=True))
    other = S.Engine(0)
    with pytest.raises(S.EngineError):
        other.ecdsa_verify_batch_keyset_submit(ks, *b[:4])
    other.close()
    eng.wait_all()
    ks.close()


# ---- S2K_KEYS_ADAPTIVE: the grouping that stops looking when there is nothing to find ------------------------------------
def test_adaptive_key_grouping(oracle):
    """The default setting of a new context.  Two observed batches (>= 2^16 signatures) without a repeated key, and the next
    15 are verified without looking (no grouping ran: the statistics of the call are empty), whatever their keys; the 16th
    looks again, finds the groups of a batch that has them, and the one after it builds tables as if nothing had been
    learned.  Verdicts never depend on any of it; S2K_KEYS_AUTO looks in every call; submit / wait children share the
    state of their parent; reset forgets."""
    import secp256k1_voi_amd as S
    eng = S.Engine(0)
    n = 1 << 16
    lone = damaged_batch(eng, n, n, 71)                 # every signature under its own key
    shared = damaged_batch(eng, n, n // 16, 72)         # 16 per key
    eng.set_key_grouping(S.KEYS_OFF)
    ref_lone, ref_shared = eng.ecdsa_verify_batch(*lone), eng.ecdsa_verify_batch(*shared)
    m = 2048
    for arrs, ref in ((lone, ref_lone), (shared, ref_shared)):
        assert np.array_equal(ref[:m], oracle.ecdsa_verify_batch(*(a[:m] for a in arrs), nthreads=os.cpu_count() or 1))
        assert 0 < int(ref.sum()) < n
    eng.set_key_grouping(S.KEYS_ADAPTIVE)
    a0 = eng.key_grouping_adaptive()
    assert a0["skipped"] == 0 and a0["miss_streak"] == 0
    for k in range(2):                                  # looked at, nothing found
        assert np.array_equal(eng.ecdsa_verify_batch(*lone), ref_lone)
        st = eng.key_grouping_stats()
        assert st["keyed"] == 0 and st["general"] == n, (k, st)
    idx = np.arange(n).reshape(16, n // 16)[:, :3072].reshape(-1)     # (key of signature i: i mod n / 16) 3072 keys, 16 each
    small = [np.ascontiguousarray(a[idx]) for a in shared]            # 49152 signatures, below 2^16: neither learned from nor skipped
    assert np.array_equal(eng.ecdsa_verify_batch(*small), ref_shared[idx]) and eng.key_grouping_stats()["tables"] == 3072
    for k in range(15):                                 # not looked at - even where there would be something to find
        arrs, ref = (lone, ref_lone) if k < 8 else (shared, ref_shared)
        assert np.array_equal(eng.ecdsa_verify_batch(*arrs), ref)
        st = eng.key_grouping_stats()
        assert st["keyed"] == 0 and st["tables"] == 0 and st["general"] == 0, (k, st)
        ad = eng.key_grouping_adaptive()
        assert ad["skipped"] == k + 1 and ad["skip_left"] == 14 - k and ad["observed"] == 2, (k, ad)
    assert np.array_equal(eng.ecdsa_verify_batch(*shared), ref_shared)      # the sixteenth looks again
    st = eng.key_grouping_stats()
    assert st["tables"] == n // 16 and st["keyed"] + st["general"] == n and st["keyed"] > n * 9 // 10, st   # (damaged keys stand alone)
    assert eng.key_grouping_adaptive()["probes"] == 1
    assert np.array_equal(eng.ecdsa_verify_batch(*shared), ref_shared)      # ... and what it found ends the skipping
    assert eng.key_grouping_stats()["tables"] == n // 16
    ad = eng.key_grouping_adaptive()
    assert ad["miss_streak"] == 0 and ad["skip_left"] == 0 and ad["skipped"] == 15 and ad["observed"] == 3, ad
    # learn again, then S2K_KEYS_AUTO: looks in every call whatever was learned; the learned state is kept for later
    for k in range(3):
        assert np.array_equal(eng.ecdsa_verify_batch(*lone), ref_lone)
    assert eng.key_grouping_adaptive()["skipped"] == 16
    eng.set_key_grouping(S.KEYS_AUTO)
    assert np.array_equal(eng.ecdsa_verify_batch(*shared), ref_shared) and eng.key_grouping_stats()["tables"] == n // 16
    eng.set_key_grouping(S.KEYS_ADAPTIVE)
    assert np.array_equal(eng.ecdsa_verify_batch(*shared), ref_shared) and eng.key_grouping_stats()["tables"] == 0
    assert eng.key_grouping_adaptive(reset=True)["skipped"] == 17
    assert np.array_equal(eng.ecdsa_verify_batch(*shared), ref_shared) and eng.key_grouping_stats()["tables"] == n // 16
    # submit / wait: the children's calls are the parent's observations
    eng2 = S.Engine(0)
    for k in range(6):
        assert np.array_equal(eng2.ecdsa_verify_batch_submit(*lone).wait(), ref_lone)
    ad = eng2.key_grouping_adaptive()
    assert ad["observed"] == 2 and ad["skipped"] == 4, ad
    eng2.close()
    eng.close()


# ---- inputs that would make the LAST ladder addition exceptional ----------------------------------------------------
def test_last_ladder_addition_collisions(eng, oracle):
    """u2 = r/s = -26 lambda (general ladder) and -26 * 16^28 lambda (ladder over per-key tables): with the plain odd split the
    last table addition of the ladder adds a point to itself, Z is 0 from there on and the lane is the worklist's - anyone
    can build such a batch (any r, s = r / u2) and it cost 3.4 x a step.  sc_split_glv_odd now takes the other lattice vector
    for these (tests/test_glv_odd_model.py: the rule, and the simulation that finds no other such value among 32 800
    structured scalars): the device split must BE the model's, nothing may reach the worklist, and the verdicts - a fifth of
    the items are made valid - must be the oracle's on every path."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import U2_LAST_ADDITION_GENERAL, U2_LAST_ADDITION_KEYED, synth_ladder_collision_batch
    import test_glv_odd_model as M
    # the device's split against the model's, on the grid of small natural halves (incl. the two special values)
    ks = [(a * w + b * w * M.LAM) % M.N for w in (1, 16 ** 28) for a in range(-30, 31, 2) for b in range(-30, 31)]
    ks = [k for k in ks if k] + [U2_LAST_ADDITION_GENERAL, U2_LAST_ADDITION_KEYED]
    k1, k2, sg = eng.fn_split_glv_odd_batch([b32(v) for v in ks])
    for v, a, b, sgn in zip(ks, k1, k2, sg):
        m1, m2 = M.make_odd(*M.split_reference(v))
        a, b = int.from_bytes(bytes(a), "big"), int.from_bytes(bytes(b), "big")
        assert (-a if sgn & 1 else a, -b if sgn & 2 else b) == (m1, m2), hex(v)
    for v in (U2_LAST_ADDITION_GENERAL, U2_LAST_ADDITION_KEYED):
        assert M.make_odd(*M.split_reference(v)) != M.make_odd(*M.split_reference(v), avoid_last_addition=False)
    n = 8192
    for u2v in (U2_LAST_ADDITION_KEYED, U2_LAST_ADDITION_GENERAL):
        pub, e, r, s = synth_ladder_collision_batch(eng, n, 64, seed=77, u2_value=u2v, valid_every=5)
        exp = oracle.ecdsa_verify_batch(pub, e, r, s, nthreads=os.cpu_count() or 1)
        assert exp[::5].all() and int(exp.sum()) == len(range(0, n, 5))
        for mode in (S.KEYS_AUTO, S.KEYS_OFF):
            eng.set_key_grouping(mode)
            try:
                got = eng.ecdsa_verify_batch(pub, e, r, s)
                st = eng.key_grouping_stats()
            finally:
                eng.set_key_grouping(S.KEYS_AUTO)
            assert np.array_equal(got, exp), (hex(u2v), mode)
            assert st["complete"] == 0, st                  # decided by the ladders themselves
        assert np.array_equal(eng.ecdsa_verify_batch(pub, e, r, s, force_complete=True), exp)
        assert np.array_equal(eng.ecdsa_verify_batch(pub, e, r, s, force_worklist=True), exp)


def test_streaming_boundary_randomised_stress():
    """tools/stress_pipeline.py: random batches, sizes, grouping modes, flags, pinned / pageable buffers, depths and waiting
    orders through submit / wait on one context and through a two-member group, synchronous calls in between; every ticket
    against the synchronous call, samples against the oracle."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_pipeline.py"), "25", "404"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "ok:" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


def test_group_bench_tool_one_device():
    """tools/group_bench.py (the single-process route to all GPUs of a node) on the one device this box has: runs, checks every
    batch's verdict pattern itself, prints one JSON line."""
    import json
    import subprocess
    import sys
    for extra in ([], ["--keyset"]):                  # the second run: key set on the member, batches naming keys by index
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "group_bench.py"), "--devices", "0", "--batch-log2", "16", "--batches", "6",
                            "--keys-log2", "10"] + extra, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-1500:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["member_stats_last_shard"][0]["n"] == 1 << 16
        assert (d["keyset"] is not None and d["keyset"]["keys"] == 1 << 10) if extra else d["keyset"] is None
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "group_bench.py"), "--devices", "0,0", "--batch-log2", "14", "--whole-batch"],
                       capture_output=True, text=True, timeout=900)          # configs 3 and 4, two members on the one device
    assert p.returncode == 0, p.stderr[-1500:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["msm"]["terms"] == 2 << 14 and d["schnorr_rlc"]["sigs_per_s"] > 0


# ---- key sets with joint tables -------------------------------------------------------------------------------------------
def test_keyset_joint_tables_edge_scalars(eng, oracle):
    """S2K_KEYSET_JOINT: per digit position the sums E_a + s phi(E_b) of the chunk's entries, one table addition for both half
    scalars.  VALID signatures with chosen u2 = r/s around the recoding's corners (every digit pair (d1, d2) must pick the right
    entry and sign: tiny values, all-ones and alternating nibbles, lambda and its neighbours, n - small, values whose halves
    have opposite signs) under a few keys, plus damaged ones: verdicts of the joint layout, the chunk layout and the batch
    verifier must agree with the oracle - the same for the 5- and 6-bit layouts (S2K_KEYSET_JOINT5 / JOINT6: 26 / 22 positions),
    with single digits at every one of their positions; S2K_KEYSET_AUTO picks the 5-bit joint layout for a small set."""
    import secp256k1_voi_amd as S
    rnd = random.Random(909)
    ds = [rnd.randrange(1, R.N) for _ in range(5)]
    Q = [oracle.scalar_base_mult_vartime(b32(d)) for d in ds]
    u2s = [1, 2, 3, 15, 16, 17, 255, 256, R.N - 1, R.N - 2, R.N - 16, R.LAMBDA, R.LAMBDA + 1, R.LAMBDA - 1, R.N - R.LAMBDA, 2 * R.LAMBDA % R.N,
           (1 << 128) - 1, 1 << 128, (1 << 128) + 1, (1 << 127), int("8" * 64, 16) % R.N, int("7" * 64, 16) % R.N, int("f0" * 32, 16) % R.N,
           int("0f" * 32, 16) % R.N, int("a5" * 32, 16) % R.N, (-26 * R.LAMBDA) % R.N, (-26 * 16 ** 28 * R.LAMBDA) % R.N]
    u2s += [rnd.randrange(1, R.N) for _ in range(300)] + [(rnd.randrange(1, 1 << 20) << (4 * rnd.randrange(0, 60))) % R.N or 1 for _ in range(200)]
    # the wider layouts' corners: single 5- and 6-bit digits at every position, all-ones and alternating windows, next to lambda too
    u2s += [(d << (5 * i)) % R.N or 1 for i in range(26) for d in (1, 15, 16, 17, 31)] + [(d << (6 * i)) % R.N or 1 for i in range(22) for d in (1, 31, 32, 33, 63)]
    u2s += [int("1" * 128, 2), int("10" * 64, 2), int("01" * 64, 2), int("11111" * 25 + "0" * 3, 2) % R.N, int("100000" * 21, 2),
            (R.LAMBDA * 31) % R.N, (R.LAMBDA * 63 + 31) % R.N, (R.LAMBDA << 5) % R.N, (R.LAMBDA << 6) % R.N]
    pub, dig, rr, ss, kidx = [], [], [], [], []
    for i, u2 in enumerate(u2s):
        ki = i % len(ds)
        k = rnd.randrange(1, R.N)                               # R = k G, k = u1 + u2 d
        Rp = oracle.scalar_base_mult_vartime(b32(k))
        r = int.from_bytes(Rp[1:33], "big") % R.N
        if r == 0:
            continue
        u1 = (k - u2 * ds[ki]) % R.N
        s = r * pow(u2, -1, R.N) % R.N
        e = u1 * s % R.N
        if i % 9 == 8:
            e = (e + 1) % R.N                                   # damaged
        pub.append(Q[ki][1:]); dig.append(b32(e)); rr.append(b32(r)); ss.append(b32(s)); kidx.append(ki)
    exp = oracle.ecdsa_verify_batch(b"".join(pub), b"".join(dig), b"".join(rr), b"".join(ss), nthreads=os.cpu_count() or 1)
    assert 0.8 * len(exp) < int(exp.sum()) < len(exp)
    keys = np.frombuffer(b"".join(q[1:] for q in Q), np.uint8).reshape(-1, 64)
    for layout in (S.KEYSET_JOINT, S.KEYSET_JOINT5, S.KEYSET_JOINT6, S.KEYSET_CHUNKS, S.KEYSET_AUTO):
        ks = eng.keyset_create(keys, layout)
        assert ks.layout() == (S.KEYSET_JOINT5 if layout == S.KEYSET_AUTO else layout)      # (AUTO: the widest layout a small set has room for)
        got = eng.ecdsa_verify_batch_keyset(ks, np.array(kidx, np.uint32), dig, rr, ss)
        assert np.array_equal(got, exp), (layout, np.nonzero(got != exp)[0][:10])
        ks.close()
    assert np.array_equal(eng.ecdsa_verify_batch(pub, dig, rr, ss), exp)
