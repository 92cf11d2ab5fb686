"""Host-side logic of the product (secp256k1_voi_amd/csrc/ingest.hip, der.h) — no GPU needed: strict
DER / compact signature parsing and the BIP-0066 shape check, against the oracle's
restatement, the Wycheproof encoding classes and the BIP-0066 vectors.
"""
import random

import secp256k1_voi_amd as S
from conftest import load_golden

H = bytes.fromhex
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def test_parse_asn1_on_wycheproof(oracle):
    # secec/wycheproof_test.go:349-352: these flag classes are always rejected by the parser
    early = {"BerEncodedSignature", "InvalidTypesInSignature", "InvalidEncoding", "MissingZero"}
    total = rejected = 0
    for fn in ("wycheproof_ecdsa_sha256.json", "wycheproof_ecdsa_sha512.json"):
        for c in load_golden(fn)["cases"]:
            sig = H(c["sig"])
            got = S.parse_asn1_signature(sig)
            assert got == oracle.parse_asn1_signature(sig), (fn, c["tcId"])
            total += 1
            if got is None:
                rejected += 1
                assert not c["valid"]
            if early & set(c["flags"]):
                assert got is None, (c["tcId"], c["flags"])
            if c["valid"]:
                assert got is not None
    assert total == 996 and rejected > 100


def der(r, s):
    def integer(x):
        b = x.to_bytes((x.bit_length() + 8) // 8 or 1, "big")
        return b"\x02" + bytes([len(b)]) + b
    body = integer(r) + integer(s)
    return b"\x30" + bytes([len(body)]) + body


def test_parse_asn1_constructed_and_mutated(oracle):
    rnd = random.Random(3)
    for _ in range(300):
        r, s = rnd.randrange(1, N), rnd.randrange(1, N)
        enc = der(r, s)
        assert S.parse_asn1_signature(enc) == (r.to_bytes(32, "big"), s.to_bytes(32, "big"))
        for _ in range(20):
            m = bytearray(enc)
            k = rnd.randrange(4)
            if k == 0:
                m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
            elif k == 1:
                del m[rnd.randrange(len(m))]
            elif k == 2:
                m.insert(rnd.randrange(len(m) + 1), rnd.randrange(256))
            else:
                m += bytes([rnd.randrange(256)])
            m = bytes(m)
            assert S.parse_asn1_signature(m) == oracle.parse_asn1_signature(m), m.hex()
    for r, s in [(0, 1), (1, 0), (N, 1), (1, N), (N - 1, N - 1), (1, 1), (2**255, 5)]:
        exp = (r.to_bytes(32, "big"), s.to_bytes(32, "big")) if 0 < r < N and 0 < s < N else None
        assert S.parse_asn1_signature(der(r, s)) == exp
    # long-form length where short form is required, indefinite length, trailing bytes
    good = der(5, 7)
    assert S.parse_asn1_signature(b"\x30\x81" + good[1:]) is None
    assert S.parse_asn1_signature(b"\x30\x80" + good[2:] + b"\x00\x00") is None
    assert S.parse_asn1_signature(good + b"\x00") is None
    assert S.parse_asn1_signature(b"") is None


def test_parse_compact():
    rnd = random.Random(4)
    for _ in range(100):
        r, s = rnd.randrange(1, N), rnd.randrange(1, N)
        b = r.to_bytes(32, "big") + s.to_bytes(32, "big")
        assert S.parse_compact_signature(b) == (b[:32], b[32:])
    z, one, n_ = bytes(32), (1).to_bytes(32, "big"), N.to_bytes(32, "big")
    for b in (z + one, one + z, n_ + one, one + n_, one + one + b"\x00", (one + one)[:63], b""):
        assert S.parse_compact_signature(b) is None


def test_bip0066_vectors(oracle):
    d = load_golden("bip0066.json")
    assert len(d["valid"]) == 9 and len(d["invalid_decode"]) > 10
    for i, v in enumerate(d["valid"]):
        b = H(v["der"]) + b"\x45"                       # with the sighash byte (asn1_shitcoin_test.go:60)
        assert S.is_valid_signature_encoding_bip0066(b)
        rs = S.parse_asn1_signature(b[:-1])
        r, s = int(v["r"], 16), int(v["s"], 16)
        if i == 8:                                       # r = s = 0 (asn1_shitcoin_test.go:78-81)
            assert r == 0 and s == 0 and rs is None
        else:
            assert rs == (r.to_bytes(32, "big"), s.to_bytes(32, "big"))
    for v in d["invalid_decode"]:
        assert not S.is_valid_signature_encoding_bip0066(H(v["der"]) + b"\x45"), v["exception"]
    assert not S.is_valid_signature_encoding_bip0066(b"")
    assert not S.is_valid_signature_encoding_bip0066(bytes(74))


def test_bip340_challenge_midstate_constant():
    """csrc/sha256.h starts every BIP-340 challenge hash from a constant state instead of compressing the block
    SHA256(tag) || SHA256(tag) per signature: the constant in the header must be that state (FIPS 180-4 compression
    re-done here), and hashing on from it must give hashlib's digest of the whole tagged message."""
    import hashlib
    import os
    import re
    import struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "secp256k1_voi_amd", "csrc", "sha256.h")).read()

    def table(name):
        m = re.search(name + r"\[\d+\]\s*=\s*\{([^}]*)\}", src)
        return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",") if x.strip()]
    K, IV, mid = table("SHA256_K"), table("SHA256_IV"), table("BIP340_CHALLENGE_MIDSTATE")
    assert len(K) == 64 and len(IV) == 8 and len(mid) == 8
    rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & 0xFFFFFFFF

    def compress(st, block):
        w = list(struct.unpack(">16I", block))
        for i in range(16, 64):
            s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
            s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
            w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
        a, b, c, d, e, f, g, h = st
        for i in range(64):
            t1 = (h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i]) & 0xFFFFFFFF
            t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
            h, g, f, e, d, c, b, a = g, f, e, (d + t1) & 0xFFFFFFFF, c, b, a, (t1 + t2) & 0xFFFFFFFF
        return [(x + y) & 0xFFFFFFFF for x, y in zip(st, [a, b, c, d, e, f, g, h])]
    tag = hashlib.sha256(b"BIP0340/challenge").digest()
    assert compress(IV, tag + tag) == mid
    body = bytes(range(96)) + b"message"
    msg = body + b"\x80"
    msg += b"\0" * ((56 - len(msg)) % 64) + struct.pack(">Q", (64 + len(body)) * 8)
    st = mid
    for i in range(0, len(msg), 64):
        st = compress(st, msg[i:i + 64])
    assert b"".join(struct.pack(">I", x) for x in st) == hashlib.sha256(tag + tag + body).digest()
