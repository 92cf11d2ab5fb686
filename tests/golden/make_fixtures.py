#!/usr/bin/env python3
"""Convert the reference's own test DATA (public third-party vectors and in-test KATs)
into the small JSON fixtures committed next to this script.

Run in the authoring container only (it reads /root/reference, which does not exist on
the GPU box):   python tests/golden/make_fixtures.py

Sources (all relative to /root/reference):
  secec/testdata/wycheproof/ecdsa_secp256k1_sha256_test.json   (463 cases)
  secec/testdata/wycheproof/ecdsa_secp256k1_sha512_test.json   (533 cases)
      harness semantics: secec/wycheproof_test.go:317-334 (mustFail = result != "valid",
      digest = full hash output, Verify(hBytes, sig, nil))
  secec/testdata/wycheproof/ecdh_secp256k1_test.json  (valid cases; shared = x(d*Q),
      secec/secec.go:53-56, wycheproof_test.go:303-305)
  secec/bitcoin/testdata/bip-0340-test-vectors.csv              (schnorr_test.go:149-245)
  secec/testdata/secp256k1_rfc6979_sha256.csv                   (ecdsa_k_test.go:244-278)
  internal/gentable/point_mul_table.bin                         (hash + sampled entries)
  secec/bitcoin/testdata/bip-0066-test-vectors.json             (asn1_shitcoin_test.go:43-112)
  in-test KATs quoted as data: point_test.go:39,49,242-261; point_mul_glv_test.go:18,25-45;
  point_mul_glv.go:40-56; scalar_test.go:27-41,76-95; field_test.go:29-41; ecdsa_k_test.go:49-70
Only data is emitted: inputs and expected outputs.  No reference source text is stored.
"""
import csv
import hashlib
import json
import os

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

SPKI_UNCOMP = "3056301006072a8648ce3d020106052b8104000a034200"
SPKI_COMP = "3036301006072a8648ce3d020106052b8104000a032200"


def dump(name, obj):
    p = os.path.join(OUT, name)
    with open(p, "w") as f:
        json.dump(obj, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print(f"{name}: {os.path.getsize(p)} bytes")


def wycheproof_ecdsa(fn, hashname):
    d = json.load(open(os.path.join(REF, "secec/testdata/wycheproof", fn)))
    cases = []
    for g in d["testGroups"]:
        pub = g["publicKey"]["uncompressed"]
        assert g["sha"] == hashname
        for t in g["tests"]:
            digest = hashlib.new(hashname.replace("-", "").lower(), bytes.fromhex(t["msg"])).hexdigest()
            cases.append({"tcId": t["tcId"], "pub": pub, "digest": digest, "sig": t["sig"],
                          "valid": t["result"] == "valid", "flags": t["flags"]})
    assert len(cases) == d["numberOfTests"]
    return {"source": f"secec/testdata/wycheproof/{fn}", "generatorVersion": d["generatorVersion"], "cases": cases}


def wycheproof_ecdh():
    fn = "ecdh_secp256k1_test.json"
    d = json.load(open(os.path.join(REF, "secec/testdata/wycheproof", fn)))
    cases = []
    for g in d["testGroups"]:
        for t in g["tests"]:
            pub = t["public"]
            if pub.startswith(SPKI_UNCOMP) and len(pub) == len(SPKI_UNCOMP) + 130:
                point = pub[len(SPKI_UNCOMP):]
            elif pub.startswith(SPKI_COMP) and len(pub) == len(SPKI_COMP) + 66:
                point = pub[len(SPKI_COMP):]
            else:
                continue
            # valid, or the one "acceptable" compressed-point case the harness accepts (wycheproof_test.go:225-231)
            ok = t["result"] == "valid" or (t["tcId"] == 2 and t["result"] == "acceptable")
            if not ok:
                continue
            priv = int(t["private"], 16)
            cases.append({"tcId": t["tcId"], "point": point, "private": "%064x" % priv, "shared": t["shared"],
                          "flags": t["flags"]})
    return {"source": f"secec/testdata/wycheproof/{fn}", "cases": cases}


def bip340():
    rows = list(csv.DictReader(open(os.path.join(REF, "secec/bitcoin/testdata/bip-0340-test-vectors.csv"))))
    cases = [{"index": int(r["index"]), "secret_key": r["secret key"], "public_key": r["public key"],
              "aux_rand": r["aux_rand"], "message": r["message"], "signature": r["signature"],
              "valid": r["verification result"] == "TRUE", "comment": r["comment"]} for r in rows]
    return {"source": "secec/bitcoin/testdata/bip-0340-test-vectors.csv", "cases": cases}


def rfc6979():
    cases = []
    for line in open(os.path.join(REF, "secec/testdata/secp256k1_rfc6979_sha256.csv")):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        priv, rest = line.split(",", 1)
        msg, sig = rest.rsplit(",", 1)
        cases.append({"private": "%064x" % int(priv), "message": msg, "digest": hashlib.sha256(msg.encode()).hexdigest(),
                      "sig": sig.lower()})
    return {"source": "secec/testdata/secp256k1_rfc6979_sha256.csv", "cases": cases}


def gentable():
    blob = open(os.path.join(REF, "internal/gentable/point_mul_table.bin"), "rb").read()
    assert len(blob) == 32 * 255 * 64
    samples = []
    for i in (0, 1, 2, 15, 16, 30, 31):
        for j in (0, 1, 2, 14, 15, 127, 128, 253, 254):
            off = (i * 255 + j) * 64
            samples.append({"i": i, "j": j, "xy": blob[off:off + 64].hex()})
    return {"source": "internal/gentable/point_mul_table.bin", "layout": "tbl[i][j] = (j+1)*2^(8i)*G as X||Y big-endian",
            "sha256": hashlib.sha256(blob).hexdigest(), "size": len(blob), "samples": samples}


def bip0066():
    d = json.load(open(os.path.join(REF, "secec/bitcoin/testdata/bip-0066-test-vectors.json")))
    return {"source": "secec/bitcoin/testdata/bip-0066-test-vectors.json",
            "valid": [{"der": v["DER"], "r": v["r"], "s": v["s"]} for v in d["valid"]],
            "invalid_decode": [{"der": v["DER"], "exception": v["exception"]} for v in d["invalid"]["decode"]]}


def kats():
    return {
        "generator": {  # point_test.go:39,49 ; point.go:18-21
            "compressed": "0279BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798".lower(),
            "uncompressed": ("0479BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"
                             "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8").lower()},
        "libsecp256k1_ecmult_const": {  # point_test.go:242-261
            "a": "04" + "6d98654457ff52b8cf1b81265b802a5ba97f9263b1e880449335132591bc450a535c59f7325e5d2bc391fbe83c12787c337e4a98e82a90110123ba37dd769c7d",
            "xn": "649d4f77c4242df77f2079c914530327a31b876ad2d8ce2a2236d5c6d7b2029b",
            "b": "04" + "237736844d209dc7098a786f20d06fcd070a38bfc11ac651030043191e2a8786ed8c3b8ec06dd57bd06ea66e45492b0fb84e4e1bfb77e21f96baae2a63dec956"},
        "glv": {  # point_mul_glv_test.go:18,25-45 ; point_mul_glv.go:40-56
            "lambda": "5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72",
            "neg_lambda": "ac9c52b33fa3cf1f5ad9e3fd77ed9ba4a880b9fc8ec739c2e0cfc810b51283cf",
            "beta": "7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee",
            "boundary_scalars": [
                "d938a5667f479e3eb5b3c7faefdb37493aa0585cc5ea2367e1b660db0209e6fc",
                "d938a5667f479e3eb5b3c7faefdb37493aa0585cc5ea2367e1b660db0209e6fd",
                "d938a5667f479e3eb5b3c7faefdb37493aa0585cc5ea2367e1b660db0209e6fe",
                "d938a5667f479e3eb5b3c7faefdb37493aa0585cc5ea2367e1b660db0209e6ff",
                "2c9c52b33fa3cf1f5ad9e3fd77ed9ba5b294b8933722e9a500e698ca4cf7632d",
                "2c9c52b33fa3cf1f5ad9e3fd77ed9ba5b294b8933722e9a500e698ca4cf7632e",
                "2c9c52b33fa3cf1f5ad9e3fd77ed9ba5b294b8933722e9a500e698ca4cf7632f",
                "2c9c52b33fa3cf1f5ad9e3fd77ed9ba5b294b8933722e9a500e698ca4cf76330",
                "7fffffffffffffffffffffffffffffffd576e73557a4501ddfe92f46681b209f",
                "7fffffffffffffffffffffffffffffffd576e73557a4501ddfe92f46681b20a0",
                "7fffffffffffffffffffffffffffffffd576e73557a4501ddfe92f46681b20a1",
                "7fffffffffffffffffffffffffffffffd576e73557a4501ddfe92f46681b20a2",
                "d363ad4cc05c30e0a5261c0288126459f85915d77825b696beebc5c2833ede11",
                "d363ad4cc05c30e0a5261c0288126459f85915d77825b696beebc5c2833ede12",
                "d363ad4cc05c30e0a5261c0288126459f85915d77825b696beebc5c2833ede13",
                "d363ad4cc05c30e0a5261c0288126459f85915d77825b696beebc5c2833ede14",
                "26c75a9980b861c14a4c38051024c8b4704d760ee95e7cd3de1bfdb1ce2c5a42",
                "26c75a9980b861c14a4c38051024c8b4704d760ee95e7cd3de1bfdb1ce2c5a43",
                "26c75a9980b861c14a4c38051024c8b4704d760ee95e7cd3de1bfdb1ce2c5a44",
                "26c75a9980b861c14a4c38051024c8b4704d760ee95e7cd3de1bfdb1ce2c5a45"]},
        "field_geq_p": {  # field_test.go:29-41 : raw -> reduced value
            "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f": 0,
            "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc30": 1,
            "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc31": 2,
            "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffc2f": 0x100000000},
        "scalar_geq_n": {  # scalar_test.go:27-41
            "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141": "0",
            "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364142": "1",
            "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364143": "2",
            "ffffffffffffffffffffffffffffffffbaaedce6af48a03bbfd25e8cd0364141": "100000000000000000000000000000000"},
        "half_n": {  # scalar_test.go:76-95
            "leq": ["7fffffffffffffffffffffffffffffff5d576e7357a4501ddfe92f46681b20a0",
                    "7fffffffffffffffffffffffffffffff5d576e7357a4501ddfe92f46681b209f"],
            "gt": ["7fffffffffffffffffffffffffffffff5d576e7357a4501ddfe92f46681b20a1",
                   "7fffffffffffffffffffffffffffffff5d576e7357a4501ddfe92f46681b20a2"]},
        "reused_k_pairs": {  # ecdsa_k_test.go:49-70 — two valid (r,s) with known key; msg hashes :38-42
            "private": "000000000000000000000000E5C4D0A8249A6F27E5E0C9D534F4DA15223F42AD".lower(),
            "sigs": [
                {"msg": "",  # filled below
                 "r": "317365e5fada9ddf645d224952c398b3bfa5dcb4d11803213ee6565639ad25be",
                 "s": "c69a9505efb9a417b5f59f62ad7cd8140947b2e2189fb7ef111a8206d2ed8aa5"},
                {"msg": "",
                 "r": "317365e5fada9ddf645d224952c398b3bfa5dcb4d11803213ee6565639ad25be",
                 "s": "14577cbf24e320e45c14efe63b4190e2e00f9936102f00d67cb5e79113ef5a9b"}]},
    }


def main():
    dump("wycheproof_ecdsa_sha256.json", wycheproof_ecdsa("ecdsa_secp256k1_sha256_test.json", "SHA-256"))
    dump("wycheproof_ecdsa_sha512.json", wycheproof_ecdsa("ecdsa_secp256k1_sha512_test.json", "SHA-512"))
    dump("wycheproof_ecdh.json", wycheproof_ecdh())
    dump("bip340.json", bip340())
    dump("rfc6979.json", rfc6979())
    dump("gentable.json", gentable())
    dump("bip0066.json", bip0066())
    k = kats()
    # messages of the reused-k pairs (ecdsa_k_test.go:38-42), hashed with SHA-256 (secec_test.go:26-29)
    msgs = ["This is Fail(TM). But it's not Epic(TM) yet...", "With private keys you can SIGN THINGS"]
    for sgn, m in zip(k["reused_k_pairs"]["sigs"], msgs):
        sgn["msg"] = m
        sgn["digest"] = hashlib.sha256(m.encode()).hexdigest()
    dump("kats.json", k)


if __name__ == "__main__":
    main()
