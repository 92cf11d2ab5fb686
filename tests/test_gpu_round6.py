"""Round 6 on the GPU.  One generator-table view per call: a call whose ladder kernel ran on one table
and whose worklist kernel is launched after the background build has published another must give the verdicts of the
reference's pure function (secec/ecdsa.go:436-465) — the lanes that show a difference are the ones the ladder leaves for
the worklist kernel with the tag "twice the last generator-table entry" (engine.hip: WL_DOUBLE_LAST)."""
import random

import numpy as np
import pytest

import pyref as R
from test_gpu_hotpath import b32
from test_gpu_parity import sig_from_u

pytestmark = pytest.mark.gpu


def last_addition_doubles(rnd, bits, count):
    """Signatures whose plain ladder over a `bits`-wide generator table meets P + P in its LAST addition:
    u2 Q + u1 G - T_last = T_last, T_last = (top digit of u1 + 1) 2^(bits (W - 1)) G   (engine.hip: table layout)."""
    W = (256 + bits - 1) // bits
    items = []
    for _ in range(count):
        d = rnd.randrange(1, R.N)
        u1 = rnd.randrange(1, R.N)
        t_last = ((u1 >> (bits * (W - 1))) + 1) << (bits * (W - 1))
        u2 = (2 * t_last - u1) * pow(d, -1, R.N) % R.N
        items.append(sig_from_u(d, u1, u2))
    return items


def test_table_swap_inside_a_call_changes_no_verdict(oracle):
    """The hook publishes another table width between the ladder launch and the worklist launch of one call (what the
    background build did at a moment of its own choosing until round 5).  Lanes crafted for BOTH widths, so that whichever
    table the ladder ran on, some of them carry the tag; ordinary valid / invalid signatures around them."""
    import secp256k1_voi_amd as S
    e = S.Engine(0)                                     # an automatic context of its own (the session's contexts share the registry)
    try:
        e.set_small_batch_max(0)
        e.set_mid_batch_max(0)
        e.set_key_grouping(S.KEYS_OFF)                  # the plain ladder: generator part inside k_verify_fast<MODE_ECDSA>
        e.gt_wait()                                     # (the background build publishes when IT likes: let it finish, then the hook decides)
        first = e.gt_info()["bits"]
        other = 22 if first != 22 else 20
        rnd = random.Random(600)
        items = last_addition_doubles(rnd, first, 24) + last_addition_doubles(rnd, other, 24)
        for i in range(200):                            # ordinary lanes, a third of them damaged
            d = rnd.randrange(1, R.N)
            it = list(sig_from_u(d, rnd.randrange(1, R.N), rnd.randrange(1, R.N)))
            if i % 3 == 0:
                it[1] = b32(int.from_bytes(it[1], "big") ^ 1)
            items.append(tuple(it))
        rnd.shuffle(items)
        pub, dig, rr, ss, _ = zip(*items)
        exp = oracle.ecdsa_verify_batch(b"".join(pub), b"".join(dig), b"".join(rr), b"".join(ss), nthreads=8)
        assert 0 < int(exp.sum()) < len(items)
        for a, b in ((first, other), (other, first)):   # ladder on a, worklist kernel launched after b was published; and back
            assert e.gt_info()["bits"] == a
            e.debug_gt_swap_in_call(b)
            got = e.ecdsa_verify_batch(pub, dig, rr, ss)
            assert e.key_grouping_stats()["complete"] >= 24, "the crafted lanes did not reach the worklist kernel"
            assert e.gt_info()["bits"] == b, "the hook did not publish the other table"
            assert got.tolist() == exp.tolist(), f"verdicts changed when the {b}-bit table appeared inside a call on the {a}-bit one"
        # and without the hook, on either table
        assert e.ecdsa_verify_batch(pub, dig, rr, ss).tolist() == exp.tolist()
    finally:
        e.close()


def test_explicit_width_is_inherited_by_child_contexts_and_groups(oracle):
    """the child contexts of submit / wait and the members of a group run on the width their parent asked for -
    no 40 GiB build behind the back of a caller who chose a narrow table."""
    import secp256k1_voi_amd as S
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 4096, seed=601, corrupt_every=5)
    exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], nthreads=8)
    arrs = [np.ascontiguousarray(w[k]) for k in ("pub", "digest", "r", "s")]
    e = S.Engine(0, gt_bits=18)
    try:
        before = e.gt_info()
        t1, t2 = e.ecdsa_verify_batch_submit(*arrs), e.ecdsa_verify_batch_submit(*arrs)
        assert np.array_equal(t1.wait(), exp) and np.array_equal(t2.wait(), exp)
        after = e.gt_info()
        assert after["bits"] == 18 and not after["building"]
        assert after["bytes"] == before["bytes"], "a child context of an explicit-width parent built or started another table"
    finally:
        e.close()
    g = S.Group([0], gt_bits=18)
    try:
        assert np.array_equal(g.ecdsa_verify_batch(*arrs), exp)
        assert g.gt_wait() == 18
    finally:
        g.close()


def test_context_lifecycle_with_a_build_in_flight(oracle):
    """the last context of a device is destroyed while the background build runs - the builder is cancelled and
    joined (never detached), and the next context starts from a clean registry.  Run in a process of its own so that no other
    context holds the registry."""
    import subprocess
    import sys
    import os
    code = r'''
import time, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import secp256k1_voi_amd as S
import oracle
from workload import make_ecdsa_batch
w = make_ecdsa_batch(oracle, 2048, seed=602, corrupt_every=3)
exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], nthreads=4)
for delay in (0.0, 0.3, 0.8, 1.5):
    t0 = time.time()
    e = S.Engine(0)
    got = e.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])      # (kicks the builder)
    assert np.array_equal(got, exp)
    time.sleep(delay)
    info = e.gt_info()
    e.close()                                                              # cancels + joins the builder
    print("delay", delay, "info", info["bits"], info["building"], "close after", round(time.time() - t0, 2), flush=True)
e = S.Engine(0)                                                            # exit WITHOUT destroying: the atexit handler joins
e.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
print("leaving with a live context", flush=True)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "leaving with a live context" in r.stdout


def test_table_registry_under_concurrent_contexts(oracle):
    """The per-device table registry from several host threads at once (never two tables of one width, a width
    that is being built is waited for, explicit widths beside the automatic build): contexts of widths 0 / 18 / 0 / 22 / 18
    are created, used and destroyed concurrently, twice; every verdict array equals the oracle's and the registry ends with
    one table per width."""
    import threading
    import secp256k1_voi_amd as S
    from workload import make_ecdsa_batch
    w = make_ecdsa_batch(oracle, 3000, seed=603, corrupt_every=4)
    exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"], nthreads=8)
    keep = S.Engine(0)                                  # (holds the registry while the others come and go)
    errors, infos = [], []

    def run(bits, rounds):
        try:
            for _ in range(rounds):
                e = S.Engine(0, gt_bits=bits)
                try:
                    e.set_small_batch_max(0)
                    e.set_mid_batch_max(0)
                    for _ in range(3):
                        got = e.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
                        if not np.array_equal(got, exp):
                            errors.append(("verdicts", bits))
                    infos.append((bits, e.gt_info()))
                finally:
                    e.close()
        except Exception as ex:   # noqa: BLE001 - reported below
            errors.append((bits, repr(ex)))

    try:
        th = [threading.Thread(target=run, args=(b, 2)) for b in (0, 18, 0, 22, 18)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in th), "a context creation is stuck in the registry"
        assert not errors, errors
        for bits, info in infos:
            assert info["bits"] == bits or (bits == 0 and info["bits"] in (20, 22, 24, 26)), (bits, info)
        keep.gt_wait()
        held = keep.gt_info()["bytes"]
        # bytes held = a sum over DISTINCT widths (the session's other tests may have left explicit widths behind) that contains
        # 18, 20, 22 and the width in use: a table built twice would not fit any such sum
        import itertools
        size = {b: ((256 + b - 1) // b) * (64 << b) for b in range(16, 27)}
        now = keep.gt_info()["bits"]
        must = {18, 20, 22, now}
        rest = [b for b in size if b not in must]
        base = sum(size[b] for b in must)
        assert any(base + sum(size[b] for b in extra) == held for k in range(len(rest) + 1) for extra in itertools.combinations(rest, k)), (held, now)
    finally:
        keep.close()


def test_group_create_ex_arguments():
    """s2k_group_create_ex: widths outside 16 .. 26 and unknown flags are refused; the wait flag returns a group whose members
    are on their final tables."""
    import ctypes as C
    import secp256k1_voi_amd as S
    lib = S.load_library()
    devs = (C.c_int * 1)(0)
    h = C.c_void_p()
    for bits, flags in ((15, 0), (27, 0), (-1, 0), (0, 2), (18, 0x80)):
        assert lib.s2k_group_create_ex(devs, 1, bits, flags, C.byref(h)) != 0 and not h.value, (bits, flags)
    assert lib.s2k_group_create_ex(devs, 0, 0, 0, C.byref(h)) != 0
    g = S.Group([0], wait_tables=True)
    try:
        assert g.gt_wait() in (20, 22, 24, 26)
    finally:
        g.close()
