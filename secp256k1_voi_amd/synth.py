"""Synthetic workloads built with the engine's own batched primitives: valid
low-s ECDSA signatures for bench.py and the full-size GPU tests.  Needs a GPU.
"""
import numpy as np

from . import OP_ADD, OP_INV, OP_MUL, OP_NEG

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
HALF_N = np.frombuffer((N_ORDER >> 1).to_bytes(32, "big"), dtype=np.uint8)


def be_gt(a, b):
    """row-wise a > b for big-endian byte rows (a: (n,32), b: (32,))."""
    diff = a != b
    first = diff.argmax(axis=1)
    anyd = diff.any(axis=1)
    rows = np.arange(a.shape[0])
    return anyd & (a[rows, first] > b[first])


def synth_batch(eng, n, n_keys, seed, key_idx=None):
    """Valid low-s ECDSA signatures built with the engine's own batched primitives
    (scalar_base_mult, Fn inverse/mul/add); returns uint8 arrays pub (n,64), digest, r, s.
    Signature i is made with key key_idx[i] (default i mod n_keys)."""
    rng = np.random.default_rng(seed)

    def rand_scalars(m):
        a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        a[:, 0] &= 0x7F             # < 2^255 < n, non-zero with overwhelming probability
        a[:, 31] |= 1
        return a

    d = rand_scalars(n_keys)
    Q = eng.scalar_base_mult_batch(d)[:, 1:]
    key_idx = np.arange(n) % n_keys if key_idx is None else np.asarray(key_idx)
    k = rand_scalars(n)
    digest = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    Rp = eng.scalar_base_mult_batch(k)
    zero = np.zeros((n, 32), np.uint8)
    r, _ = eng.fn_op_batch(OP_ADD, Rp[:, 1:33], zero)          # x(R) mod n
    e, _ = eng.fn_op_batch(OP_ADD, digest, zero)                 # digest mod n
    rd, _ = eng.fn_op_batch(OP_MUL, r, d[key_idx])
    t, _ = eng.fn_op_batch(OP_ADD, e, rd)
    kinv, _ = eng.fn_op_batch(OP_INV, k)
    s, _ = eng.fn_op_batch(OP_MUL, kinv, t)
    sneg, _ = eng.fn_op_batch(OP_NEG, s)
    hi = be_gt(s, HALF_N)
    s[hi] = sneg[hi]                                               # low-s (ecdsa.go:385-387)
    return np.ascontiguousarray(Q[key_idx]), digest, r, s




def _rand_scalars(rng, m):
    a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    a[:, 0] &= 0x7F
    a[:, 31] |= 1
    return a


def synth_all_fallback_batch(eng, n, n_keys, seed):
    """Adversarial ECDSA inputs: u1*G + u2*Q = infinity for every item, so the
    Jacobian ladder ends at Z = 0 in every lane.  Anyone holding a key pair d can make these: pick u2 and r,
    set u1 = -u2*d, s = r/u2, e = u1*s (or, for given digests: r = -e/d, any s).  All verdicts are 0 (R = infinity,
    ecdsa.go:450).  Until round 4 the whole batch then went through the complete-formula worklist kernel; the ladder
    kernel now recognises the case in its final addition (the name of this function is history)."""
    rng = np.random.default_rng(seed)
    d = _rand_scalars(rng, n_keys)
    Q = eng.scalar_base_mult_batch(d)[:, 1:]
    key_idx = np.arange(n) % n_keys
    u2 = _rand_scalars(rng, n)
    r = _rand_scalars(rng, n)
    u2d, _ = eng.fn_op_batch(OP_MUL, u2, d[key_idx])
    u1, _ = eng.fn_op_batch(OP_NEG, u2d)
    u2inv, _ = eng.fn_op_batch(OP_INV, u2)
    s, _ = eng.fn_op_batch(OP_MUL, r, u2inv)
    e, _ = eng.fn_op_batch(OP_MUL, u1, s)
    return np.ascontiguousarray(Q[key_idx]), e, r, s


def synth_equal_points_batch(eng, n, n_keys, seed):
    """The other exceptional case an input can force in the FINAL addition: u1*G == u2*Q (P + P), i.e. r = e/d - pick u2
    and r, set u1 = u2*d, s = r/u2, e = u1*s.  R = 2*u1*G; with a random r the verdicts are 0 (x(R) != r)."""
    rng = np.random.default_rng(seed)
    d = _rand_scalars(rng, n_keys)
    Q = eng.scalar_base_mult_batch(d)[:, 1:]
    key_idx = np.arange(n) % n_keys
    u2 = _rand_scalars(rng, n)
    r = _rand_scalars(rng, n)
    u1, _ = eng.fn_op_batch(OP_MUL, u2, d[key_idx])
    u2inv, _ = eng.fn_op_batch(OP_INV, u2)
    s, _ = eng.fn_op_batch(OP_MUL, r, u2inv)
    e, _ = eng.fn_op_batch(OP_MUL, u1, s)
    return np.ascontiguousarray(Q[key_idx]), e, r, s


# u2 = r/s for which - with the PLAIN odd split - the LAST table addition of a ladder meets its own partial sum (P + P inside the
# ladder, Z = 0 from there on; sc_split_glv_odd takes the other lattice vector for them since round 4):
# u2 = 2 d c with d the last signed digit of the lambda-half and c = lambda (general ladder: digit 0 is added last) or
# 16^28 lambda (ladder over per-key tables: round 0, chunk 7 is added last); d = -13 is the self-consistent one.  Found by
# integer simulation of the ladders' partial sums.  No key is needed to use them: any r, s = r / u2.
U2_LAST_ADDITION_GENERAL = 0x87e0663476a3092f3a2127be2e21ceceb7763854dc6939d318220a5890470db5
U2_LAST_ADDITION_KEYED = 0xcecf64212ab6eb5a5997b96cac25ca2bb234b7d1e5bd755372f392d91409e89f


def synth_ladder_collision_batch(eng, n, n_keys, seed, u2_value=U2_LAST_ADDITION_KEYED, valid_every=0):
    """Signatures with the SAME u2 = r/s in every item, chosen so that the last table addition of the ladder would be
    exceptional under the plain odd split (the fast kernels then cannot decide these lanes and the worklist does).  r random, s = r / u2, random digests:
    invalid (verdict 0) - except every `valid_every`-th item, which is made valid for its key (R = u1 G + u2 Q computed by the
    engine, r = x(R) mod n; needs the key's d: u1 + u2 d = k)."""
    rng = np.random.default_rng(seed)
    d = _rand_scalars(rng, n_keys)
    Q = eng.scalar_base_mult_batch(d)[:, 1:]
    key_idx = np.arange(n) % n_keys
    u2 = np.tile(np.frombuffer(int(u2_value).to_bytes(32, "big"), np.uint8), (n, 1))
    u2inv, _ = eng.fn_op_batch(OP_INV, u2)
    r = _rand_scalars(rng, n)
    e = _rand_scalars(rng, n)
    if valid_every:
        idx = np.arange(0, n, valid_every)
        k = _rand_scalars(rng, len(idx))                          # R = k G with k = u1 + u2 d  ->  u1 = k - u2 d
        Rp = eng.scalar_base_mult_batch(k)
        zero = np.zeros((len(idx), 32), np.uint8)
        rv, _ = eng.fn_op_batch(OP_ADD, Rp[:, 1:33], zero)        # x(R) mod n
        u2d, _ = eng.fn_op_batch(OP_MUL, u2[idx], d[key_idx[idx]])
        nu2d, _ = eng.fn_op_batch(OP_NEG, u2d)
        u1, _ = eng.fn_op_batch(OP_ADD, k, nu2d)
        sv, _ = eng.fn_op_batch(OP_MUL, rv, u2inv[idx])           # s = r / u2
        ev, _ = eng.fn_op_batch(OP_MUL, u1, sv)                   # e = u1 s
        r[idx], e[idx] = rv, ev
    s, _ = eng.fn_op_batch(OP_MUL, r, u2inv)
    return np.ascontiguousarray(Q[key_idx]), e, r, s


def synth_msm_terms(eng, n, seed):
    """n points with known discrete logarithms (P_i = d_i*G, 65-byte records) and scalars k_i
    (uniform plus a sprinkle of edge values: 0, 1, n-1, zero windows), and the expected sum's
    discrete logarithm sum k_i d_i mod n."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    d[:, 0] &= 0x7F
    d[:, 31] |= 1
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    k[:, 0] &= 0x7F
    edge = [0, 1, N_ORDER - 1, 1 << 128, (1 << 128) - 1, 0xFFFF << 64, N_ORDER >> 1]
    for j, v in enumerate(edge):
        for i in range(j * 7 + 3, n, max(n // 13, 1) + j):
            k[i] = np.frombuffer(int(v).to_bytes(32, "big"), dtype=np.uint8)
    pts = eng.scalar_base_mult_batch(d)
    kb, db = k.tobytes(), d.tobytes()
    tot = 0
    for i in range(n):
        tot += int.from_bytes(kb[32 * i:32 * i + 32], "big") * int.from_bytes(db[32 * i:32 * i + 32], "big")
    return k, pts, tot % N_ORDER


def synth_schnorr_batch(eng, n, n_keys, seed, msg_len=32):
    """n distinct valid BIP-340 signatures over random messages (schnorr.go:158-218 signing
    equations: even-y key and nonce points, e = tagged hash, s = k + e d), built with the
    engine's batched primitives and hashlib.  Returns pk (n,32), msgs (n,msg_len), sig (n,64)."""
    import hashlib
    rng = np.random.default_rng(seed)
    d = _rand_scalars(rng, n_keys)
    P = eng.scalar_base_mult_batch(d)
    dneg, _ = eng.fn_op_batch(OP_NEG, d)
    odd = (P[:, 64] & 1) == 1
    d[odd] = dneg[odd]
    key_idx = np.arange(n) % n_keys
    k = _rand_scalars(rng, n)
    Rp = eng.scalar_base_mult_batch(k)
    kneg, _ = eng.fn_op_batch(OP_NEG, k)
    odd = (Rp[:, 64] & 1) == 1
    k[odd] = kneg[odd]
    msgs = rng.integers(0, 256, size=(n, msg_len), dtype=np.uint8)
    pk = np.ascontiguousarray(P[key_idx, 1:33])
    rx = np.ascontiguousarray(Rp[:, 1:33])
    th = hashlib.sha256(b"BIP0340/challenge").digest()
    h0 = hashlib.sha256(th + th)
    pre = np.concatenate([rx, pk, msgs], axis=1).tobytes()
    L = 64 + msg_len
    e = bytearray(32 * n)
    for i in range(n):
        h = h0.copy()
        h.update(pre[L * i:L * i + L])
        e[32 * i:32 * i + 32] = h.digest()
    e = np.frombuffer(bytes(e), dtype=np.uint8).reshape(n, 32)
    ed, _ = eng.fn_op_batch(OP_MUL, e, d[key_idx])            # inputs are reduced mod n first
    s, _ = eng.fn_op_batch(OP_ADD, k, ed)
    return pk, msgs, np.ascontiguousarray(np.concatenate([rx, s], axis=1))
