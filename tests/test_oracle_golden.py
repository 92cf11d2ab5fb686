"""Pins the CPU oracle (oracle/) against every golden vector the reference's own tests
hold for the verify / scalar-mult path, and against an independent
big-int implementation (tests/pyref.py).  CPU only.
"""
import random

import pytest

import pyref as R
from conftest import load_golden

b32 = R.b32
H = bytes.fromhex


# ---- 1. Wycheproof ECDSA (secec/wycheproof_test.go:317-334, :400-403) ----
@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_ecdsa(oracle, fn):
    d = load_golden(fn)
    n_valid = 0
    for c in d["cases"]:
        pub, digest, sig = H(c["pub"]), H(c["digest"]), H(c["sig"])
        # one-shot: PublicKey.Verify(hBytes, sig, nil)
        got = oracle.ecdsa_verify_asn1(pub, digest, sig)
        assert got == int(c["valid"]), (fn, c["tcId"], c["flags"])
        # split: ParseASN1Signature + verify
        rs = oracle.parse_asn1_signature(sig)
        if rs is None:
            assert not c["valid"]
            continue
        ok = oracle.ecdsa_verify_raw(pub[1:], digest, rs[0], rs[1])
        assert ok == c["valid"], (fn, c["tcId"], c["flags"])
        # independent check of the arithmetic part
        Q = (int.from_bytes(pub[1:33], "big"), int.from_bytes(pub[33:], "big"))
        assert R.ecdsa_verify(Q, digest, int.from_bytes(rs[0], "big"), int.from_bytes(rs[1], "big")) == c["valid"]
        n_valid += c["valid"]
    assert n_valid == {"wycheproof_ecdsa_sha256.json": 164, "wycheproof_ecdsa_sha512.json": 233}[fn]


# ---- 2. Wycheproof ECDH: variable-base scalar mult KATs (secec.go:53-56) ----
def test_wycheproof_ecdh_scalar_mult(oracle):
    d = load_golden("wycheproof_ecdh.json")
    assert len(d["cases"]) > 400
    for c in d["cases"]:
        p = oracle.point_from_bytes(H(c["point"]))
        assert p is not None
        out = oracle.scalar_mult_vartime(H(c["private"]), p)
        assert out[1:33] == H(c["shared"]), c["tcId"]


# ---- 3. generator table blob (internal/gentable/point_mul_table.go:20-51) ----
def test_generator_table(oracle):
    d = load_golden("gentable.json")
    assert oracle.generator_table_sha256().hex() == d["sha256"]
    for s in d["samples"]:
        assert oracle.generator_table_entry(s["i"], s["j"]).hex() == s["xy"]
        # (j+1) * 2^(8i) * G, independently
        exp = R.mul((s["j"] + 1) << (8 * s["i"]), R.G)
        assert s["xy"] == (b32(exp[0]) + b32(exp[1])).hex()


# ---- 4. KATs quoted from the reference's tests ----
def test_kats(oracle):
    k = load_golden("kats.json")
    g = oracle.point_generator()
    assert g.hex() == k["generator"]["uncompressed"]
    assert oracle.point_from_bytes(H(k["generator"]["compressed"])) == g
    assert oracle.point_compressed(g).hex() == k["generator"]["compressed"]
    assert oracle.point_from_bytes(b"\x00") == bytes(65)
    assert oracle.point_compressed(bytes(65)) == b"\x00"
    assert oracle.point_from_bytes(b"\x05" + g[1:]) is None
    assert oracle.point_from_bytes(g[:64]) is None
    kat = k["libsecp256k1_ecmult_const"]
    assert oracle.scalar_mult_vartime(H(kat["xn"]), H(kat["a"])).hex() == kat["b"]
    assert oracle.scalar_mult_trivial(H(kat["xn"]), H(kat["a"])).hex() == kat["b"]
    for raw, red in k["field_geq_p"].items():
        out, did = oracle.fp_reduce(H(raw))
        assert did == 1 and int.from_bytes(out, "big") == red
        assert not oracle.fp_is_canonical(H(raw))
    for raw, red in k["scalar_geq_n"].items():
        out, did = oracle.fn_reduce(H(raw))
        assert did == 1 and int.from_bytes(out, "big") == int(red, 16)
        assert not oracle.fn_is_canonical(H(raw))
    for s in k["half_n"]["leq"]:
        assert not oracle.fn_is_gt_half_n(H(s))
    for s in k["half_n"]["gt"]:
        assert oracle.fn_is_gt_half_n(H(s))
    rk = k["reused_k_pairs"]
    Q = R.mul(int(rk["private"], 16), R.G)
    for s in rk["sigs"]:
        assert oracle.ecdsa_verify_raw(b32(Q[0]) + b32(Q[1]), H(s["digest"]), H(s["r"]), H(s["s"]))


# ---- 5. GLV split properties (point_mul_glv_test.go:48-95) ----
def test_glv_split(oracle):
    k = load_golden("kats.json")["glv"]
    lam = int(k["lambda"], 16)
    assert (lam + int(k["neg_lambda"], 16)) % R.N == 0 and lam == R.LAMBDA and int(k["beta"], 16) == R.BETA
    rnd = random.Random(5)
    scalars = [0, 1, rnd.randrange(1, R.N)] + [int(s, 16) for s in k["boundary_scalars"]]
    P = R.mul(rnd.randrange(1, R.N), R.G)
    Pp = (P[0] * R.BETA % R.P, P[1])
    assert R.mul(lam, P) == Pp
    for v in scalars:
        k1b, k2b = oracle.fn_split_glv(b32(v))
        k1, k2 = int.from_bytes(k1b, "big"), int.from_bytes(k2b, "big")
        assert (k1 + k2 * lam) % R.N == v
        a1 = R.N - k1 if oracle.fn_is_gt_half_n(k1b) else k1
        a2 = R.N - k2 if oracle.fn_is_gt_half_n(k2b) else k2
        assert a1 < 2**128 and a2 < 2**128
        assert R.add(R.mul(k1, P), R.mul(k2, Pp)) == R.mul(v, P)
        assert oracle.scalar_mult_vartime(b32(v), R.enc65(P)) == R.enc65(R.mul(v, P))


# ---- 6. BIP-340 CSV (secec/bitcoin/schnorr_test.go:149-245) ----
def test_bip340(oracle):
    d = load_golden("bip340.json")
    assert len(d["cases"]) == 19
    for c in d["cases"]:
        pk, msg, sig = H(c["public_key"]), H(c["message"]), H(c["signature"])
        got = oracle.schnorr_verify(pk, msg, sig)
        if c["index"] in (5, 14):          # invalid public keys (schnorr_test.go:174-177)
            assert got == -1
        else:
            assert got == int(c["valid"]), c["index"]
        assert R.schnorr_verify(pk, msg, sig) == c["valid"]
        if c["secret_key"]:
            d_ = int(c["secret_key"], 16)
            assert b32(R.mul(d_, R.G)[0]) == pk
            assert R.schnorr_sign(d_, msg, H(c["aux_rand"])) == sig


# ---- 7. RFC 6979 signatures are valid signatures (ecdsa_k_test.go:244-278) ----
def test_rfc6979_vectors_verify(oracle):
    d = load_golden("rfc6979.json")
    assert len(d["cases"]) >= 10
    for c in d["cases"]:
        Q = oracle.scalar_base_mult_vartime(H(c["private"]))
        assert oracle.ecdsa_verify_asn1(Q, H(c["digest"]), H(c["sig"])) == 1
        assert oracle.ecdsa_verify_asn1(Q, H(c["digest"]), H(c["sig"]), reject_malleable=True) == 1   # low-s normalised
        comp = oracle.point_compressed(Q)
        assert oracle.ecdsa_verify_asn1(comp, H(c["digest"]), H(c["sig"])) == 1
        bad = bytearray(H(c["digest"])); bad[0] ^= 1
        assert oracle.ecdsa_verify_asn1(Q, bytes(bad), H(c["sig"])) == 0


# ---- 8. randomised cross-checks against the independent implementation ----
def test_random_field_scalar(oracle):
    rnd = random.Random(11)
    edge = [0, 1, 2, R.P - 1, R.P - 2, 2**255, 2**32 + 977, (R.P + 1) // 2]
    vals = edge + [rnd.randrange(R.P) for _ in range(300)]
    for a in vals:
        b = rnd.choice(vals)
        assert oracle.fp_mul(b32(a), b32(b)) == b32(a * b % R.P)
        assert oracle.fp_sqr(b32(a)) == b32(a * a % R.P)
        assert oracle.fp_add(b32(a), b32(b)) == b32((a + b) % R.P)
        assert oracle.fp_sub(b32(a), b32(b)) == b32((a - b) % R.P)
        assert oracle.fp_neg(b32(a)) == b32(-a % R.P)
        assert oracle.fp_inv(b32(a)) == b32(pow(a, R.P - 2, R.P))        # Invert(0) = 0
        r, ok = oracle.fp_sqrt(b32(a))
        exp = R.sqrt_p(a)
        assert ok == (exp is not None)
        if ok:
            assert int.from_bytes(r, "big") in (exp, R.P - exp)
        else:
            assert r == bytes(32)
        an, bn = a % R.N, b % R.N
        assert oracle.fn_mul(b32(an), b32(bn)) == b32(an * bn % R.N)
        assert oracle.fn_add(b32(an), b32(bn)) == b32((an + bn) % R.N)
        assert oracle.fn_sub(b32(an), b32(bn)) == b32((an - bn) % R.N)
        assert oracle.fn_neg(b32(an)) == b32(-an % R.N)
        assert oracle.fn_inv(b32(an)) == b32(pow(an, R.N - 2, R.N))
        assert oracle.fn_is_gt_half_n(b32(an)) == (an > R.N // 2)


def test_random_point_algebra(oracle):
    # point_test.go:136-213 (a+0, a+a=2a, commutativity, a-a=0) with random Z (:359-373)
    rnd = random.Random(12)
    for _ in range(30):
        a = R.mul(rnd.randrange(1, R.N), R.G)
        b = R.mul(rnd.randrange(1, R.N), R.G)
        za, zb = b32(rnd.randrange(1, R.P)), b32(rnd.randrange(1, R.P))
        A, B, O = R.enc65(a), R.enc65(b), bytes(65)
        assert oracle.point_add_randz(A, za, B, zb) == R.enc65(R.add(a, b))
        assert oracle.point_add_randz(B, zb, A, za) == R.enc65(R.add(a, b))
        assert oracle.point_add_randz(A, za, A, zb) == R.enc65(R.add(a, a)) == oracle.point_double(A)
        assert oracle.point_add(A, O) == A and oracle.point_add(O, A) == A and oracle.point_add(O, O) == O
        assert oracle.point_add_randz(A, za, oracle.point_neg(A), zb) == O
        assert oracle.point_equal_randz(A, za, A, zb) and not oracle.point_equal_randz(A, za, B, zb)
        assert oracle.point_double(O) == O


def test_random_scalar_mults(oracle):
    # point_test.go:262-347: every variant against the trivially-correct double-and-add
    rnd = random.Random(13)
    for it in range(25):
        k = rnd.randrange(R.N) if it else 0
        u1 = rnd.randrange(R.N)
        q = R.mul(rnd.randrange(1, R.N), R.G)
        Q = R.enc65(q)
        z = b32(rnd.randrange(1, R.P))
        exp = R.enc65(R.mul(k, q))
        assert oracle.scalar_mult_vartime_randz(b32(k), Q, z) == exp
        assert oracle.scalar_mult_trivial(b32(k), Q) == exp
        assert oracle.scalar_base_mult_vartime(b32(k)) == R.enc65(R.mul(k, R.G))
        assert oracle.double_scalar_mult_basepoint_vartime(b32(u1), b32(k), Q) == R.enc65(R.add(R.mul(u1, R.G), R.mul(k, q)))
    # u1*G + u2*Q = identity
    d = rnd.randrange(1, R.N)
    q = R.mul(d, R.G)
    u2 = rnd.randrange(1, R.N)
    u1 = (-u2 * d) % R.N
    assert oracle.double_scalar_mult_basepoint_vartime(b32(u1), b32(u2), R.enc65(q)) == bytes(65)


@pytest.mark.parametrize("n", [0, 1, 2, 32, 64])
def test_multi_scalar_mult(oracle, n):
    # point_mul_multi_test.go:14-72
    rnd = random.Random(100 + n)
    ks = [rnd.randrange(R.N) for _ in range(n)]
    ps = [R.mul(rnd.randrange(1, R.N), R.G) for _ in range(n)]
    exp = None
    for k, p in zip(ks, ps):
        exp = R.add(exp, R.mul(k, p))
    assert oracle.multi_scalar_mult_vartime([b32(k) for k in ks], [R.enc65(p) for p in ps]) == R.enc65(exp)


def test_random_ecdsa_and_edge_rules(oracle):
    rnd = random.Random(14)
    for _ in range(20):
        d = rnd.randrange(1, R.N)
        q = R.mul(d, R.G)
        Q = b32(q[0]) + b32(q[1])
        dig = rnd.randbytes(rnd.choice([32, 48, 64]))
        r, s = R.ecdsa_sign(d, dig, rnd.randrange(1, R.N))
        assert oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(s))
        assert oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(R.N - s))                 # malleable twin accepted ...
        hi = max(s, R.N - s)
        assert not oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(hi), True)           # ... unless RejectMalleable
        assert oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(R.N - hi), True)
        assert not oracle.ecdsa_verify_raw(Q, dig[:31], b32(r), b32(s))             # short digest (ecdsa.go:478)
        assert not oracle.ecdsa_verify_raw(Q, dig, bytes(32), b32(s))               # r = 0
        assert not oracle.ecdsa_verify_raw(Q, dig, b32(r), bytes(32))               # s = 0
        assert not oracle.ecdsa_verify_raw(Q, dig, b32(R.N), b32(s))                # r = n (non-canonical)
        assert not oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(R.N + s if R.N + s < 2**256 else R.N))
        bad_q = b32(q[0]) + b32((q[1] + 1) % R.P)
        assert not oracle.ecdsa_verify_raw(bad_q, dig, b32(r), b32(s))              # not on curve
        flip = bytearray(dig); flip[5] ^= 0x40
        assert not oracle.ecdsa_verify_raw(Q, bytes(flip), b32(r), b32(s))
    # digest >= n is reduced (ecdsa.go:483-484): e = 2^256-1 mod n
    d = 7
    q = R.mul(d, R.G)
    dig = b"\xff" * 32
    r, s = R.ecdsa_sign(d, dig, 12345)
    assert oracle.ecdsa_verify_raw(b32(q[0]) + b32(q[1]), dig, b32(r), b32(s))


def test_batch_matches_single(oracle):
    import numpy as np
    rnd = random.Random(15)
    n = 64
    q, dg, rr, ss, exp = b"", b"", b"", b"", []
    for i in range(n):
        d = rnd.randrange(1, R.N)
        pt = R.mul(d, R.G)
        dig = rnd.randbytes(32)
        r, s = R.ecdsa_sign(d, dig, rnd.randrange(1, R.N))
        if i % 3 == 0:
            s = (s + 1) % R.N
        q += b32(pt[0]) + b32(pt[1]); dg += dig; rr += b32(r); ss += b32(s)
        exp.append(R.ecdsa_verify(pt, dig, r, s))
    for th in (1, 3):
        out = oracle.ecdsa_verify_batch(q, dg, rr, ss, nthreads=th)
        assert out.tolist() == [int(e) for e in exp]


def test_op_counts(oracle):
    # ≈2916 Fp + ≈304 Fn Montgomery multiplications per verification
    rnd = random.Random(16)
    d = rnd.randrange(1, R.N)
    q = R.mul(d, R.G)
    dig = rnd.randbytes(32)
    r, s = R.ecdsa_sign(d, dig, rnd.randrange(1, R.N))
    oracle.counters_reset()
    assert oracle.ecdsa_verify_raw(b32(q[0]) + b32(q[1]), dig, b32(r), b32(s))
    fp, fn = oracle.counters_get()
    assert 2700 < fp < 3100 and 290 < fn < 320, (fp, fn)


# ---- 9. public-key recovery: exhaustive recovery finds the key iff the signature verifies
#         (secec/wycheproof_test.go:417-438) ----
@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_recovery(oracle, fn):
    d = load_golden(fn)
    n_found = 0
    for c in d["cases"]:
        rs = oracle.parse_asn1_signature(H(c["sig"]))
        if rs is None:
            continue
        pub, digest = H(c["pub"]), H(c["digest"])
        ok = oracle.ecdsa_verify_raw(pub[1:], digest, rs[0], rs[1])
        found = any(oracle.ecdsa_recover(digest, rs[0], rs[1], rid) == pub for rid in range(4))
        assert found == ok, (fn, c["tcId"])
        n_found += found
        assert oracle.ecdsa_recover(digest, rs[0], rs[1], 4) is None
    assert n_found > 150


def test_recovery_random(oracle):
    rnd = random.Random(17)
    for _ in range(30):
        d = rnd.randrange(1, R.N)
        q = R.mul(d, R.G)
        dig = rnd.randbytes(32)
        k = rnd.randrange(1, R.N)
        Rp = R.mul(k, R.G)
        r, s = R.ecdsa_sign(d, dig, k)
        rid = (Rp[1] & 1) | (2 if Rp[0] >= R.N else 0)
        assert oracle.ecdsa_recover(dig, b32(r), b32(s), rid) == R.enc65(q)
        other = oracle.ecdsa_recover(dig, b32(r), b32(s), rid ^ 1)
        assert other is not None and other != R.enc65(q)
        assert oracle.ecdsa_recover(dig, bytes(32), b32(s), rid) is None
        assert oracle.ecdsa_recover(dig, b32(r), bytes(32), rid) is None
        assert oracle.ecdsa_recover(dig, b32(R.N), b32(s), rid) is None
        if r >= R.P - R.N:
            assert oracle.ecdsa_recover(dig, b32(r), b32(s), rid | 2) is None     # r + n >= p


def test_oracle_against_libcrypto(oracle):
    """Second, independent oracle: libcrypto's ECDSA_do_verify and the C restatement agree on valid and
    damaged signatures (keys off the curve excluded: libcrypto refuses to build such a key, the reference returns false)."""
    import openssl_ref
    if not openssl_ref.available():
        pytest.skip("no usable libcrypto")
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 400, seed=4242, corrupt_every=3)
    got = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
    for i in range(400):
        ref = openssl_ref.ecdsa_verify(bytes(w["pub"][i]), bytes(w["digest"][i]), bytes(w["r"][i]), bytes(w["s"][i]))
        assert bool(got[i]) == ref, (i, w["kinds"][i])
    assert 0 < int(got.sum()) < 400
