"""ctypes loader for the CPU oracle (oracle/libsecp256k1_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (secp256k1_voi_amd) never imports this module.

Every function takes/returns canonical big-endian ``bytes`` exactly as the C header
documents (oracle/secp256k1_oracle.h); points are 65-byte buffers (0x04‖X‖Y or 65 zero
bytes for the identity).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsecp256k1_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Returns the .so path."""
    src = os.path.join(_HERE, "secp256k1_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_multi_scalar_mult_vartime.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p]
        _lib.orc_ecdsa_verify_raw.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_int]
        _lib.orc_ecdsa_verify_batch.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_void_p, C.c_int]
        _lib.orc_ecdsa_verify_batch.restype = None
        _lib.orc_ecdsa_recover.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_uint]
        _lib.orc_parse_asn1_signature.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        _lib.orc_ecdsa_verify_asn1.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        _lib.orc_schnorr_verify.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        _lib.orc_point_from_bytes.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        _lib.orc_sha256.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        _lib.orc_generator_table_entry.argtypes = [C.c_char_p, C.c_uint, C.c_uint]
    return _lib


IDENTITY = bytes(65)


def _buf(n):
    return C.create_string_buffer(n)


def _un(name, a, n=32):
    o = _buf(n)
    getattr(lib(), name)(o, a)
    return o.raw


def _bin(name, a, b, n=32):
    o = _buf(n)
    getattr(lib(), name)(o, a, b)
    return o.raw


# ---- Fp ----
def fp_mul(a, b): return _bin("orc_fp_mul", a, b)
def fp_sqr(a): return _un("orc_fp_sqr", a)
def fp_add(a, b): return _bin("orc_fp_add", a, b)
def fp_sub(a, b): return _bin("orc_fp_sub", a, b)
def fp_neg(a): return _un("orc_fp_neg", a)
def fp_inv(a): return _un("orc_fp_inv", a)
def fp_is_canonical(a): return bool(lib().orc_fp_is_canonical(a))


def fp_sqrt(a):
    o = _buf(32)
    ok = lib().orc_fp_sqrt(o, a)
    return o.raw, bool(ok)


def fp_reduce(a):
    o = _buf(32)
    d = C.c_int(0)
    lib().orc_fp_reduce(o, a, C.byref(d))
    return o.raw, d.value


# ---- Fn ----
def fn_mul(a, b): return _bin("orc_fn_mul", a, b)
def fn_add(a, b): return _bin("orc_fn_add", a, b)
def fn_sub(a, b): return _bin("orc_fn_sub", a, b)
def fn_neg(a): return _un("orc_fn_neg", a)
def fn_inv(a): return _un("orc_fn_inv", a)
def fn_is_canonical(a): return bool(lib().orc_fn_is_canonical(a))
def fn_is_gt_half_n(a): return bool(lib().orc_fn_is_gt_half_n(a))


def fn_reduce(a):
    o = _buf(32)
    d = C.c_int(0)
    lib().orc_fn_reduce(o, a, C.byref(d))
    return o.raw, d.value


def fn_split_glv(k):
    k1, k2 = _buf(32), _buf(32)
    lib().orc_fn_split_glv(k1, k2, k)
    return k1.raw, k2.raw


# ---- group ----
def point_generator():
    o = _buf(65)
    lib().orc_point_generator(o)
    return o.raw


def point_from_bytes(src):
    o = _buf(65)
    rc = lib().orc_point_from_bytes(o, src, len(src))
    return o.raw if rc == 0 else None


def point_compressed(p):
    o = _buf(33)
    n = C.c_size_t(0)
    lib().orc_point_compressed(o, C.byref(n), p)
    return o.raw[: n.value]


def point_on_curve_xy(x, y): return bool(lib().orc_point_on_curve_xy(x, y))
def point_add(a, b): return _bin("orc_point_add", a, b, 65)
def point_double(a): return _un("orc_point_double", a, 65)
def point_neg(a): return _un("orc_point_neg", a, 65)


def point_add_randz(a, za, b, zb):
    o = _buf(65)
    lib().orc_point_add_randz(o, a, za, b, zb)
    return o.raw


def point_equal_randz(a, za, b, zb): return bool(lib().orc_point_equal_randz(a, za, b, zb))
def scalar_mult_vartime(k, p): return _bin("orc_scalar_mult_vartime", k, p, 65)
def scalar_mult_trivial(k, p): return _bin("orc_scalar_mult_trivial", k, p, 65)
def scalar_base_mult_vartime(k): return _un("orc_scalar_base_mult_vartime", k, 65)


def scalar_mult_vartime_randz(k, p, z):
    o = _buf(65)
    lib().orc_scalar_mult_vartime_randz(o, k, p, z)
    return o.raw


def double_scalar_mult_basepoint_vartime(u1, u2, p):
    o = _buf(65)
    lib().orc_double_scalar_mult_basepoint_vartime(o, u1, u2, p)
    return o.raw


def multi_scalar_mult_vartime(scalars, points):
    assert len(scalars) == len(points)
    o = _buf(65)
    lib().orc_multi_scalar_mult_vartime(o, len(scalars), b"".join(scalars), b"".join(points))
    return o.raw


def generator_table_entry(i, j):
    o = _buf(64)
    lib().orc_generator_table_entry(o, i, j)
    return o.raw


def generator_table_sha256():
    o = _buf(32)
    lib().orc_generator_table_sha256(o)
    return o.raw


# ---- ECDSA / Schnorr ----
def ecdsa_verify_raw(q64, digest, r, s, reject_malleable=False):
    return bool(lib().orc_ecdsa_verify_raw(q64, digest, len(digest), r, s, int(reject_malleable)))


def ecdsa_verify_batch(q, digest, r, s, reject_malleable=False, nthreads=1):
    """q: n*64 bytes, digest/r/s: n*32 bytes (bytes or numpy uint8 arrays). Returns bytes of 0/1."""
    import numpy as np
    q, digest, r, s = (np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x)
                       for x in (q, digest, r, s))
    n = r.size // 32
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_ecdsa_verify_batch(n, q.ctypes.data, digest.ctypes.data, r.ctypes.data, s.ctypes.data,
                                 int(reject_malleable), out.ctypes.data, int(nthreads))
    return out


def ecdsa_recover(digest, r, s, recovery_id):
    """RecoverPublicKey: the 65-byte uncompressed key, or None."""
    o = _buf(65)
    ok = lib().orc_ecdsa_recover(o, digest, len(digest), r, s, int(recovery_id))
    return o.raw if ok else None


def parse_asn1_signature(der):
    r, s = _buf(32), _buf(32)
    rc = lib().orc_parse_asn1_signature(r, s, der, len(der))
    return (r.raw, s.raw) if rc == 0 else None


def ecdsa_verify_asn1(pub, digest, sig, reject_malleable=False):
    return lib().orc_ecdsa_verify_asn1(pub, len(pub), digest, len(digest), sig, len(sig), int(reject_malleable))


def schnorr_verify(pk32, msg, sig):
    return lib().orc_schnorr_verify(pk32, msg, len(msg), sig, len(sig))


def sha256(data):
    o = _buf(32)
    lib().orc_sha256(o, data, len(data))
    return o.raw


def counters_reset(): lib().orc_counters_reset()


def counters_get():
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib().orc_counters_get(C.byref(a), C.byref(b))
    return a.value, b.value
