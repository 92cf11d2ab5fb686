"""Sanitizer runs of the CPU half (host builds only; the GPU pool has no sanitizers, and this file is listed in
.gpurunignore so that it does not travel to the GPU box): csrc/ct_cpu.cpp and the csrc/der.h parsers under
AddressSanitizer + UndefinedBehaviorSanitizer."""
import os

import pytest


def test_ct_under_sanitizers(tmp_path):
    """csrc/ct_cpu.cpp built for the host with AddressSanitizer and UndefinedBehaviorSanitizer (the GPU pool has no
    sanitizers; the CPU half gets them here) and driven by tests/c/ct_sanitize_main.cpp."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ct_san")
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(root, "include"), os.path.join(root, "secp256k1_voi_amd", "csrc", "ct_cpu.cpp"),
                           os.path.join(root, "tests", "c", "ct_sanitize_main.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr


def test_der_parsers_under_sanitizers(tmp_path):
    """csrc/der.h compiled host-only with AddressSanitizer + UBSan and fed every prefix, single-byte mutations,
    random splices of seed encodings and garbage, each in an exact-size heap buffer (tests/c/der_sanitize_main.cpp):
    the parsers the device ingest kernel shares must never read past their input."""
    import os
    import shutil
    import subprocess

    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "der_san")
    subprocess.check_call([hipcc, "--cuda-host-only", "-x", "hip", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(root, "secp256k1_voi_amd", "csrc"),
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "der_sanitize_main.cpp"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok"), p.stdout + p.stderr[-2000:]
