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


def _build_group_driver(tmp_path, sanitize, name):
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    csrc = os.path.join(root, "secp256k1_voi_amd", "csrc")
    subprocess.check_call([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitize, "-fno-sanitize-recover=all",
                           "-I", os.path.join(root, "include"), os.path.join(csrc, "group.cpp"), os.path.join(csrc, "topology.cpp"),
                           os.path.join(root, "tests", "c", "stub_ctx.cpp"), os.path.join(root, "tests", "c", "group_sanitize_main.cpp"),
                           "-o", exe, "-lpthread"])
    return exe


def test_group_threads_under_thread_sanitizer(tmp_path):
    """csrc/group.cpp (member threads, job queues, tickets, the failure ring, host blocks) and
    csrc/topology.cpp reach the GPU only through the C-ABI, so they are linked against tests/c/stub_ctx.cpp - contexts whose
    "device" is a worker thread that completes a ticket 0-2 ms later and writes the verdicts itself - and driven through
    random submit / wait / wait-out-of-order / fifth-submit / destroy-with-work-in-flight sequences of every group entry point,
    from one caller thread and from a producer / consumer pair, under ThreadSanitizer.  Any report fails the run (TSAN_OPTIONS
    halt_on_error), and every shard of every batch is checked."""
    import subprocess
    exe = _build_group_driver(tmp_path, "thread", "group_tsan")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    p = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok") and "ThreadSanitizer" not in p.stderr, p.stdout + p.stderr[-3000:]


def test_group_threads_under_address_sanitizer(tmp_path):
    """The same driver under AddressSanitizer + UndefinedBehaviorSanitizer: iterators kept across waits, jobs that outlive
    their batch, host blocks freed twice or not at all."""
    import subprocess
    exe = _build_group_driver(tmp_path, "address,undefined", "group_asan")
    p = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout + p.stderr[-3000:]
