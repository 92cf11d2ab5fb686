"""CPU-side checks of the boundary: the HIP library was built for gfx950, loads, and
exports every symbol include/secp256k1_voi_amd.h declares.  No compute calls (no GPU here).
"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "secp256k1_voi_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(s2k_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    import secp256k1_voi_amd as S
    assert header_symbols() == sorted(S.EXPORTED_SYMBOLS)


def test_library_exports_every_symbol():
    import secp256k1_voi_amd as S
    if not os.path.exists(S.LIB_PATH):
        S.build()
    lib = ctypes.CDLL(S.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.s2k_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.s2k_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    import secp256k1_voi_amd as S
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(S.LIB_PATH):
        S.build()
    with pytest.raises(S.EngineError):
        S.Engine(0)


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under the package may reference it
    pkg = os.path.join(ROOT, "secp256k1_voi_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "secp256k1_oracle" not in txt and "from oracle" not in txt, fn
