"""An independent second oracle for spot checks: libcrypto's ECDSA_do_verify on secp256k1 through ctypes.
Nothing of this repository's arithmetic is involved.  `available()` is False when no usable libcrypto is found."""
import ctypes as C
import ctypes.util

_lib = None
NID_secp256k1 = 714


def _load():
    global _lib
    if _lib is not None:
        return _lib
    for name in (ctypes.util.find_library("crypto"), "libcrypto.so.3", "libcrypto.so"):
        if not name:
            continue
        try:
            lib = C.CDLL(name)
            vp = C.c_void_p
            lib.EC_KEY_new_by_curve_name.restype = vp
            lib.EC_KEY_new_by_curve_name.argtypes = [C.c_int]
            lib.EC_KEY_free.argtypes = [vp]
            lib.EC_KEY_set_public_key_affine_coordinates.argtypes = [vp, vp, vp]
            lib.BN_bin2bn.restype = vp
            lib.BN_bin2bn.argtypes = [C.c_char_p, C.c_int, vp]
            lib.BN_free.argtypes = [vp]
            lib.ECDSA_SIG_new.restype = vp
            lib.ECDSA_SIG_free.argtypes = [vp]
            lib.ECDSA_SIG_set0.argtypes = [vp, vp, vp]
            lib.ECDSA_do_verify.argtypes = [C.c_char_p, C.c_int, vp, vp]
            k = lib.EC_KEY_new_by_curve_name(NID_secp256k1)
            if not k:
                continue
            lib.EC_KEY_free(k)
            _lib = lib
            return lib
        except (OSError, AttributeError):
            continue
    _lib = False
    return _lib


def available():
    return bool(_load())


def ecdsa_verify(pub64: bytes, digest32: bytes, r32: bytes, s32: bytes) -> bool:
    """ECDSA_do_verify(digest, (r, s), key) with key = the affine point pub64 = X || Y; False for keys libcrypto refuses."""
    lib = _load()
    key = lib.EC_KEY_new_by_curve_name(NID_secp256k1)
    x, y = lib.BN_bin2bn(pub64[:32], 32, None), lib.BN_bin2bn(pub64[32:], 32, None)
    try:
        if lib.EC_KEY_set_public_key_affine_coordinates(key, x, y) != 1:
            return False
        sig = lib.ECDSA_SIG_new()
        lib.ECDSA_SIG_set0(sig, lib.BN_bin2bn(r32, 32, None), lib.BN_bin2bn(s32, 32, None))   # sig owns r and s now
        ok = lib.ECDSA_do_verify(digest32, 32, sig, key) == 1
        lib.ECDSA_SIG_free(sig)
        return ok
    finally:
        lib.BN_free(x)
        lib.BN_free(y)
        lib.EC_KEY_free(key)
