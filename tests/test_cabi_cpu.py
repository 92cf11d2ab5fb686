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


def test_hot_kernel_register_budget():
    """The occupancy DESIGN.md states is a property of the compiled code objects: the ladder kernel must fit three
    waves per SIMD (<= 168 VGPRs) without spills or scratch, and so must the bucket pass of the multi-scalar multiplication."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objdir = os.path.join(root, "secp256k1_voi_amd", "build")
    tool = os.path.join(root, "tools", "kernel_regs.sh")
    if not (os.path.exists(os.path.join(objdir, "engine.o")) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")):
        import pytest
        pytest.skip("objects or llvm tools not available")
    out = subprocess.run(["bash", tool, objdir], capture_output=True, text=True).stdout
    info = {}
    for line in out.splitlines():
        m = re.match(r"\S+\s+(\S+)\s+vgpr=(\d+)\s+sgpr=(\d+)\s+spill=(\d+)\s+scratch=(\d+)", line)
        if m:
            info[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    fast = [v for k, v in info.items() if k.startswith("_Z13k_verify_fastILi0E")]
    assert fast, sorted(info)[:5]
    vgpr, _, spill, scratch = fast[0]
    assert vgpr <= 168 and spill == 0 and scratch == 0, fast[0]
    acc = [v for k, v in info.items() if "k_msm_accumulate" in k]
    assert acc and acc[0][0] <= 168 and acc[0][2] == 0 and acc[0][3] == 0, acc


def test_host_register_refuses_heap_blocks():
    """s2k_host_register takes whole pages only (the check precedes every HIP call, so it runs without a GPU): a numpy
    heap array is refused with S2K_ERR_ARG, and so is a page-aligned buffer whose length is not a multiple of the page.
    (Why: tests/test_gpu_round3.py::test_host_buffers_pinned_registered_pageable and the header.)"""
    import numpy as np
    import secp256k1_voi_amd as S
    lib = S.load_library()
    a = np.zeros(300 * 64 + 8, dtype=np.uint8)[8:]           # never page aligned and whole-paged at once
    assert lib.s2k_host_register(a.ctypes.data, a.nbytes) == -3          # S2K_ERR_ARG
    assert b"page" in lib.s2k_last_error(None)
    b = S.page_aligned_array((8192,))
    assert b.ctypes.data % 4096 == 0
    assert lib.s2k_host_register(b.ctypes.data, 5000) == -3
    try:
        S.host_register(a)
    except S.EngineError as e:
        assert "page" in str(e)
    else:
        raise AssertionError("heap array accepted")


def test_group_and_tickets_without_a_device():
    """The streaming / multi-device entry points on a box without a GPU: no device is counted, a group cannot be made
    (S2K_ERR_NO_DEVICE, no CPU fallback), null arguments are refused - and nothing crashes."""
    import ctypes as C

    import torch

    import secp256k1_voi_amd as S
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("box has GPUs")
    lib = S.load_library()
    assert lib.s2k_device_count() == 0 and S.device_count() == 0
    g = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert lib.s2k_group_create(devs, 2, C.byref(g)) == -1 and not g.value          # S2K_ERR_NO_DEVICE
    assert lib.s2k_group_create(None, 2, C.byref(g)) == -3                          # S2K_ERR_ARG
    assert lib.s2k_group_create(devs, 0, C.byref(g)) == -3
    assert lib.s2k_group_size(None) == 0
    assert lib.s2k_group_wait(None, 1) == -3
    lib.s2k_group_destroy(None)
    t = C.c_uint64(0)
    assert lib.s2k_ecdsa_verify_batch_submit(None, 0, None, None, None, None, 0, None, C.byref(t)) == -3
    assert lib.s2k_wait(None, 1) == -3 and lib.s2k_poll(None, 1) == -3 and lib.s2k_wait_all(None) == -3
    import pytest
    with pytest.raises(S.EngineError):
        S.Group([0])
