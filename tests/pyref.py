"""Independent big-int secp256k1 arithmetic, written from the curve equation and SEC 1 /
BIP-340 text (NOT from the reference's code).  It breaks the circularity between the C
oracle and the HIP engine: both are checked against this on random inputs.

Pure Python, affine coordinates; slow, used only on small cases.
"""
import hashlib

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 7) % P == 0


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def mul(k, pt):
    k %= N
    acc = None
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def b32(x):
    return int(x).to_bytes(32, "big")


def enc65(pt):
    """oracle / engine boundary encoding: 0x04‖X‖Y or 65 zero bytes."""
    if pt is None:
        return bytes(65)
    return b"\x04" + b32(pt[0]) + b32(pt[1])


def dec65(b):
    if b[0] == 0:
        return None
    return (int.from_bytes(b[1:33], "big"), int.from_bytes(b[33:65], "big"))


def sqrt_p(a):
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


def lift_x(x, odd):
    if x >= P:
        return None
    y = sqrt_p((x * x * x + 7) % P)
    if y is None:
        return None
    if (y & 1) != odd:
        y = P - y
    return (x, y)


def ecdsa_verify(Q, digest, r, s):
    """SEC 1 v2 §4.1.4 with e = leftmost 32 bytes of digest."""
    if not (1 <= r < N and 1 <= s < N):
        return False
    if len(digest) < 32:
        return False
    e = int.from_bytes(digest[:32], "big") % N
    w = pow(s, -1, N)
    R = add(mul(e * w % N, G), mul(r * w % N, Q))
    if R is None:
        return False
    return R[0] % N == r


def ecdsa_sign(d, digest, k):
    e = int.from_bytes(digest[:32], "big") % N
    R = mul(k, G)
    r = R[0] % N
    s = pow(k, -1, N) * (e + r * d) % N
    return r, s


def tagged_hash(tag, *vals):
    th = hashlib.sha256(tag.encode()).digest()
    h = hashlib.sha256(th + th)
    for v in vals:
        h.update(v)
    return h.digest()


def schnorr_verify(pk32, msg, sig):
    """BIP-340 Verify."""
    if len(sig) != 64:
        return False
    Pt = lift_x(int.from_bytes(pk32, "big"), 0)
    if Pt is None:
        return False
    r = int.from_bytes(sig[:32], "big")
    s = int.from_bytes(sig[32:], "big")
    if r >= P or s >= N:
        return False
    e = int.from_bytes(tagged_hash("BIP0340/challenge", sig[:32], pk32, msg), "big") % N
    R = add(mul(s, G), mul(N - e, Pt))
    if R is None or R[1] & 1 or R[0] != r:
        return False
    return True


def schnorr_sign(d, msg, aux):
    """BIP-340 Sign (for generating synthetic batches)."""
    Pt = mul(d, G)
    if Pt[1] & 1:
        d = N - d
    t = (d ^ int.from_bytes(tagged_hash("BIP0340/aux", aux), "big")).to_bytes(32, "big")
    k0 = int.from_bytes(tagged_hash("BIP0340/nonce", t, b32(Pt[0]), msg), "big") % N
    R = mul(k0, G)
    k = N - k0 if R[1] & 1 else k0
    e = int.from_bytes(tagged_hash("BIP0340/challenge", b32(R[0]), b32(Pt[0]), msg), "big") % N
    return b32(R[0]) + b32((k + e * d) % N)
