"""The constant-time CPU twins (s2k_ct_*) against the oracle and the reference's
vectors: Wycheproof ECDH (secec/wycheproof_test.go:303-305), the libsecp256k1 KAT
(point_test.go:242-261), GLV boundary scalars (point_mul_glv_test.go:25-45), RFC 6979 signatures
(ecdsa_k_test.go:244-278).  Runs without a GPU; the library has to be built (hipcc).
"""
import os
import random

import pytest

import pyref as R
from conftest import load_golden

b32 = R.b32
H = bytes.fromhex


@pytest.fixture(scope="module")
def S():
    import secp256k1_voi_amd as S
    if not os.path.exists(S.LIB_PATH):
        pytest.skip("library not built")
    S.load_library()
    return S


def test_ct_scalar_mult_kats(S, oracle):
    kat = load_golden("kats.json")["libsecp256k1_ecmult_const"]
    assert S.ct_scalar_mult(H(kat["xn"]), H(kat["a"])).hex() == kat["b"]
    d = load_golden("wycheproof_ecdh.json")
    n = 0
    for c in d["cases"]:
        pt = oracle.point_from_bytes(H(c["point"]))
        out = S.ct_scalar_mult(H(c["private"]), pt)
        assert out[1:33] == H(c["shared"]), c["tcId"]
        if len(c["point"]) == 130 and int(c["private"], 16) < R.N and int(c["private"], 16) > 0:
            assert S.ct_ecdh(H(c["private"]), H(c["point"])) == H(c["shared"]), c["tcId"]
            n += 1
    assert n > 100


def test_ct_scalar_mult_random_and_edges(S, oracle):
    rnd = random.Random(201)
    g = load_golden("kats.json")["glv"]
    ks = [0, 1, 2, 3, R.N - 1, R.N - 2, R.N, R.N + 1, 2**256 - 1, R.LAMBDA, R.N - R.LAMBDA, 2**128, 2**128 - 1, 2**127, 15, 16, 17] + \
         [int(s, 16) for s in g["boundary_scalars"]] + [rnd.randrange(2**256) for _ in range(150)]
    for v in ks:
        p = R.enc65(R.mul(rnd.randrange(1, R.N), R.G))
        assert S.ct_scalar_mult(b32(v), p) == oracle.scalar_mult_trivial(b32(v % R.N) if v >= R.N else b32(v), p), hex(v)
    # identity in, identity out; malformed records are refused
    assert S.ct_scalar_mult(b32(5), bytes(65)) == bytes(65)
    good = R.enc65(R.G)
    off = bytearray(good); off[64] ^= 1
    assert S.ct_scalar_mult(b32(5), bytes(off)) is None
    assert S.ct_scalar_mult(b32(5), b"\x04" + b32(R.P) + b32(7)) is None
    assert S.ct_scalar_mult(b32(5), b"\x00" + bytes(63) + b"\x01") is None


def test_ct_scalar_base_mult(S, oracle):
    rnd = random.Random(202)
    ks = [0, 1, 2, R.N - 1, R.N, R.N + 5, 2**256 - 1, 0xF, 0x10, 0xFF, 2**255, 2**252] + [1 << (4 * i) for i in range(64)] + \
         [15 << (4 * i) for i in range(64)] + [rnd.randrange(2**256) for _ in range(200)]
    for v in ks:
        assert S.ct_scalar_base_mult(b32(v)) == oracle.scalar_base_mult_vartime(b32(v)), hex(v)


def test_ct_ecdh_rejects_bad_inputs(S):
    q = R.enc65(R.mul(12345, R.G))
    assert S.ct_ecdh(b32(0), q) is None
    assert S.ct_ecdh(b32(R.N), q) is None
    assert S.ct_ecdh(b32(7), bytes(65)) is None          # identity is not a public key
    x = S.ct_ecdh(b32(7), q)
    assert x == b32(R.mul(7 * 12345, R.G)[0])


def test_ct_sign_raw(S, oracle):
    """signatures made by the CT primitive verify with the oracle (and an independent big-int check of r, s,
    low-s and the recovery id); the RFC 6979 vectors are reproduced when fed their nonces."""
    rnd = random.Random(203)
    for _ in range(60):
        d, k = rnd.randrange(1, R.N), rnd.randrange(1, R.N)
        digest = rnd.randbytes(32)
        r, s, rid = S.ct_ecdsa_sign_raw(b32(d), digest, b32(k))
        Rp = R.mul(k, R.G)
        e = int.from_bytes(digest, "big") % R.N
        s_ref = pow(k, -1, R.N) * (e + (Rp[0] % R.N) * d) % R.N
        flipped = s_ref > R.N // 2
        if flipped:
            s_ref = R.N - s_ref
        assert (int.from_bytes(r, "big"), int.from_bytes(s, "big")) == (Rp[0] % R.N, s_ref)
        assert rid == ((Rp[1] & 1) ^ int(flipped)) | (2 if Rp[0] >= R.N else 0)
        Q = R.mul(d, R.G)
        assert oracle.ecdsa_verify_raw(b32(Q[0]) + b32(Q[1]), digest, r, s, reject_malleable=True)
        assert oracle.ecdsa_recover(digest, r, s, rid) == R.enc65(Q)
    assert S.ct_ecdsa_sign_raw(b32(0), bytes(32), b32(1)) is None
    assert S.ct_ecdsa_sign_raw(b32(1), bytes(32), b32(0)) is None
    assert S.ct_ecdsa_sign_raw(b32(R.N), bytes(32), b32(1)) is None
    assert S.ct_ecdsa_sign_raw(b32(1), bytes(32), b32(R.N)) is None


def test_ct_multi_scalar_mult_vs_oracle(S, oracle):
    """Point.MultiScalarMult (point_mul_multi.go:25-67), constant-time form: equal to the oracle's Straus
    (its restatement of MultiScalarMultVartime :73-117 — same group element) for 0, 1, 2, 32, 64 terms, with the
    edge scalars and the identity / repeated / opposite points the reference's tests mix in
    (point_mul_multi_test.go:14-72)."""
    rnd = random.Random(515)
    assert S.ct_multi_scalar_mult([], []) == bytes(65)
    for n in (1, 2, 3, 32, 64):
        pts = [R.enc65(R.mul(rnd.randrange(1, R.N), R.G)) for _ in range(n)]
        ks = [b32(rnd.randrange(R.N)) for _ in range(n)]
        if n >= 3:
            ks[0], ks[1] = b32(0), b32(R.N - 1)
            pts[2] = bytes(65)                              # the identity as a term
        if n >= 32:
            pts[5] = pts[4]                                 # the same point twice
            pts[7] = oracle.point_neg(pts[6])               # and a pair of opposites with equal scalars: they cancel
            ks[7] = ks[6]
            ks[8] = b32((1 << 256) - 1)                     # SetBytes semantics: reduced mod n
            ks[9] = b32(0x0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f)
        exp = oracle.multi_scalar_mult_vartime([oracle.fn_reduce(k)[0] for k in ks], pts)
        assert S.ct_multi_scalar_mult(ks, pts) == exp, n
    # one term: exactly Point.ScalarMult (:31-33)
    p, k = R.enc65(R.mul(77, R.G)), b32(rnd.randrange(R.N))
    assert S.ct_multi_scalar_mult([k], [p]) == S.ct_scalar_mult(k, p)
    # known discrete logs: sum k_i (d_i G) == (sum k_i d_i) G
    ds = [rnd.randrange(1, R.N) for _ in range(17)]
    kk = [rnd.randrange(R.N) for _ in range(17)]
    pts = [S.ct_scalar_base_mult(b32(d)) for d in ds]
    assert S.ct_multi_scalar_mult([b32(k) for k in kk], pts) == S.ct_scalar_base_mult(b32(sum(a * b for a, b in zip(ds, kk)) % R.N))
    # malformed record -> None; length mismatch -> the reference's panic
    assert S.ct_multi_scalar_mult([k, k], [p, b"\x05" + p[1:]]) is None
    with pytest.raises(ValueError):
        S.ct_multi_scalar_mult([k], [p, p])


def test_ct_multi_scalar_mult_operation_count(S):
    """The sequence of field operations of the constant-time MultiScalarMult depends on the number of terms only:
    same multiplication count for zero, sparse, dense and random scalars, and the count is what Straus with full
    table scans costs — 15 l table entries (7 doublings + 7 additions + the copy), 252 shared doublings, 64 l
    additions (point_mul_multi.go:35-66); the variable-time form would skip the zero digits."""
    lib = S.load_library()
    rnd = random.Random(516)
    S.ct_scalar_base_mult(b32(1))
    for n in (2, 5, 32):
        pts = [S.ct_scalar_base_mult(b32(rnd.randrange(1, R.N))) for _ in range(n)]
        lib.s2k_ct_debug_fe_mul_count()
        counts = set()
        for fill in (lambda: 0, lambda: 1, lambda: R.N - 1, lambda: (1 << 256) - 1, lambda: 1 << 255, lambda: 0x10001,
                     lambda: rnd.randrange(R.N), lambda: rnd.randrange(R.N), lambda: rnd.randrange(1 << 64)):
            assert S.ct_multi_scalar_mult([b32(fill()) for _ in range(n)], pts) is not None
            counts.add(lib.s2k_ct_debug_fe_mul_count())
        assert len(counts) == 1, (n, counts)
        # per operation in ct_cpu.cpp: addition 14 products (12 + two by the constant 21), doubling 9 (8 + one by 21);
        # output: one inversion (255 squarings + 15 products) + 2; input validation: 3 per record
        add, dbl = 14, 9
        expect = n * (7 * dbl + 7 * add) + 252 * dbl + 64 * n * add + (255 + 15 + 2) + 3 * n
        assert counts == {expect}, (n, counts, expect)


def test_ct_operation_count_is_scalar_independent(S):
    """The variable-time paths skip zero digits and pick formulas by value; the constant-time twins must
    execute the same sequence whatever the scalar: same number of field multiplications for 0, 1, sparse,
    dense, boundary and random scalars (table scans and selections are masks, not branches)."""
    lib = S.load_library()
    rnd = random.Random(204)
    q = R.enc65(R.mul(rnd.randrange(1, R.N), R.G))
    S.ct_scalar_base_mult(b32(1))             # builds the generator tables once
    lib.s2k_ct_debug_fe_mul_count()
    scalars = [0, 1, 2, R.N - 1, 2**128, 2**255, (1 << 256) - 1, R.LAMBDA, 0x1111111111111111, 1 << 252] + \
        [rnd.randrange(R.N) for _ in range(20)]
    counts = set()
    for v in scalars:
        S.ct_scalar_mult(b32(v), q)
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts
    counts = set()
    for v in scalars:
        S.ct_scalar_base_mult(b32(v))
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts
    counts = set()
    for v in scalars[1:4] + scalars[10:]:
        assert S.ct_ecdsa_sign_raw(b32(v % R.N or 1), bytes(range(32)), b32((v * 7 + 1) % R.N or 1)) is not None
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts


def test_ct_machine_code_has_no_data_dependent_branches(S):
    """The constant-time claim at the level that matters: the COMPILED functions.  Optimisers turn
    `mask = (w == j)` scans back into compare-and-branch chains and secret-indexed loads (clang did, before
    the masks were routed through value_barrier()); this disassembles the object the library was linked
    from and bounds the conditional branches of every secret-handling function by its loop counters and
    argument checks."""
    import re
    import shutil
    import subprocess
    obj = os.path.join(os.path.dirname(S.LIB_PATH), "build", "ct_cpu.o")
    objdump = shutil.which("objdump")
    if not (os.path.exists(obj) and objdump):
        pytest.skip("object file or objdump not available")
    asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", "-C", obj], capture_output=True, text=True).stdout
    counts, fn = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            fn = re.sub(r"\(anonymous namespace\)::", "", m.group(1)).split("(")[0]
            counts.setdefault(fn, 0)
        elif fn and re.search(r"\tj[a-z]+\s", line) and "\tjmp" not in line:
            counts[fn] += 1
    assert "ct_scalar_mult" in counts and "ct_scalar_base_mult" in counts, sorted(counts)
    # allowed: loop back-edges (fixed trip counts), the call_once flag, NULL / range checks of the arguments, the
    # bits of the PUBLIC exponent n - 2 in the scalar inversion
    limits = {"ct_scalar_mult": 2, "ct_scalar_base_mult": 4, "lookup_projective": 1, "make_table": 1, "pt_add": 0, "pt_add_mixed": 0,
              "pt_double": 0, "pt_to_record": 0, "sc_split_glv": 0, "fe_inv": 2, "fe_sqr_n": 1, "sc_inv": 3, "sc_reduce_wide": 0,
              "s2k_ct_scalar_mult": 3, "s2k_ct_scalar_base_mult": 2, "s2k_ct_ecdh": 6, "s2k_ct_ecdsa_sign_raw": 6,
              # loops over the (public) number of terms and over the 32 scalar bytes; allocation / argument checks
              "ct_multi_scalar_mult": 8, "s2k_ct_multi_scalar_mult": 24}
    for name, lim in limits.items():
        assert counts.get(name, 0) <= lim, (name, counts.get(name))
    # nothing that looks like a switch over a 4-bit window value
    for name in ("ct_scalar_mult", "ct_scalar_base_mult", "lookup_projective"):
        assert counts.get(name, 0) < 8
    assert "ct_multi_scalar_mult" in counts or "s2k_ct_multi_scalar_mult" in counts, sorted(counts)
