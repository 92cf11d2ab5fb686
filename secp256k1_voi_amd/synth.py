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


def synth_batch(eng, n, n_keys, seed):
    """Valid low-s ECDSA signatures built with the engine's own batched primitives
    (scalar_base_mult, Fn inverse/mul/add); returns uint8 arrays pub (n,64), digest, r, s."""
    rng = np.random.default_rng(seed)

    def rand_scalars(m):
        a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        a[:, 0] &= 0x7F             # < 2^255 < n, non-zero with overwhelming probability
        a[:, 31] |= 1
        return a

    d = rand_scalars(n_keys)
    Q = eng.scalar_base_mult_batch(d)[:, 1:]
    key_idx = np.arange(n) % n_keys
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


