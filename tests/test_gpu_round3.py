"""Round-3 GPU tests: pinned / registered / pageable host buffers give the same verdicts; the memory query; the
sharded entry points (verification, multi-scalar multiplication, BIP-340 batch) with the real engine in two ranks
sharing one device; the bench's two-rank flow at the real per-rank size; the regression check of a compiler
miscompile that the product code avoids by shape.  Needs a real MI355X.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pyref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b32 = R.b32


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    return S.Engine(0)


@pytest.mark.parametrize("n", [300, 70000, 300001])
def test_host_buffers_pinned_registered_pageable(eng, oracle, n):
    """s2k_ecdsa_verify_batch from page-locked memory (s2k_host_alloc), from registered memory (s2k_host_register) and
    from pageable memory: the pinned forms take the single grouped call whose signature data arrives piece by piece
    behind the keys; all three must give the oracle's verdicts (valid and damaged signatures, repeated and lone keys),
    with the grouping on and off."""
    import secp256k1_voi_amd as S
    from secp256k1_voi_amd.synth import synth_batch
    arrs = [np.array(a) for a in synth_batch(eng, n, max(n // 9, 3), seed=300 + n)]
    rng = np.random.default_rng(n)
    for i in range(0, n, 7):                      # every seventh item damaged somewhere (key, digest, r or s)
        a = arrs[int(rng.integers(0, 4))]
        a[i, int(rng.integers(0, a.shape[1]))] ^= 1 << int(rng.integers(0, 8))
    m = min(n, 4096)
    exp_head = oracle.ecdsa_verify_batch(*(a[:m] for a in arrs), nthreads=os.cpu_count() or 1)
    pinned = [S.pinned_array(a.shape) for a in arrs]
    for d, a in zip(pinned, arrs):
        d[...] = a
    registered = [a.copy() for a in arrs]
    for a in registered:
        S.host_register(a)
    try:
        for mode in (S.KEYS_AUTO, S.KEYS_OFF):
            eng.set_key_grouping(mode)
            ref = eng.ecdsa_verify_batch(*arrs)
            assert np.array_equal(ref[:m], exp_head)
            assert 0 < int(ref.sum()) < n
            assert np.array_equal(eng.ecdsa_verify_batch(*pinned), ref)
            assert np.array_equal(eng.ecdsa_verify_batch(*registered), ref)
    finally:
        eng.set_key_grouping(S.KEYS_AUTO)
        for a in registered:
            S.host_unregister(a)


def test_device_bytes_counts_grouping_buffers(eng):
    """s2k_ctx_device_bytes: what a context holds for a batch of n - generator tables + per-signature workspace, and with
    the grouping on the grouping arrays and the table buffer (s2k_ecdsa_workspace_bytes alone understated it)."""
    import secp256k1_voi_amd as S
    n = 1 << 20
    ws = eng.workspace_bytes(n)
    eng.set_key_grouping(S.KEYS_OFF)
    off = eng.device_bytes(n)
    eng.set_key_grouping(S.KEYS_AUTO)
    auto = eng.device_bytes(n)
    assert off >= ws + (3 << 30)
    assert auto - off >= (n // 6) * 72 * 128          # the per-key table buffer alone
    assert eng.device_bytes(100) == eng.workspace_bytes(100) + off - ws   # below 256 signatures nothing is grouped
