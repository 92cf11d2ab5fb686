"""The C-ABI used from plain C, the way a cgo shim links it (no Python, no torch in the loop):
tests/c/abi_harness.c is compiled with gcc against include/secp256k1_voi_amd.h and the built library.
`cpu` part: parsers and the constant-time twins (runs anywhere); `gpu` part: a context, batch
verification of signatures made by the CT signer, and recovery of the signers' keys."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "secp256k1_voi_amd")


def build(tmp_path, with_oracle=False):
    if not os.path.exists(os.path.join(LIBDIR, "libsecp256k1_voi_amd.so")):
        pytest.skip("library not built")
    exe = str(tmp_path / "abi_harness")
    gcc = shutil.which("gcc") or "gcc"
    cmd = [gcc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_harness.c"), "-o", exe,
           "-L", LIBDIR, "-lsecp256k1_voi_amd", "-Wl,-rpath," + LIBDIR]
    if with_oracle:      # the checker, linked into the TEST program only
        import oracle
        oracle.build()
        odir = os.path.join(ROOT, "oracle")
        cmd += ["-DWITH_ORACLE", "-I", odir, "-L", odir, "-lsecp256k1_oracle", "-Wl,-rpath," + odir]
    subprocess.check_call(cmd)
    return exe


def test_c_harness_cpu(tmp_path):
    exe = build(tmp_path)
    p = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "cpu: ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_c_harness_gpu(tmp_path):
    exe = build(tmp_path)
    p = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "gpu: ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_c_harness_group_and_submit_wait(tmp_path):
    """submit / wait on one context and a two-member group (both members on device 0), driven from plain C the way the
    cgo BatchVerifier of INTEGRATION.md drives them; verdicts against the synchronous call and the CPU oracle."""
    exe = build(tmp_path, with_oracle=True)
    p = subprocess.run([exe, "group"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "group: ok" in p.stdout and "against the oracle" in p.stdout, p.stdout + p.stderr
