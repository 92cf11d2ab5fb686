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
