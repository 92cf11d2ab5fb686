"""The roofline's instruction counts are tied to the binary that ships: the trip-weighted static VALU count of the two
ladder kernels is re-taken from the code object of the BUILT library (tools/isa_count.py: disassembly, loops from backward
branches) and must agree with the committed figures bench.py prices the kernels with (profiles/*_valu_counts.json: static
recount and PMC, SQ_INSTS_VALU) within 0.5 %.  Changing a kernel without refreshing the counts fails here.  No GPU needed."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def newest_counts():
    prof = os.path.join(ROOT, "profiles")
    names = sorted((n for n in os.listdir(prof) if n.endswith("_valu_counts.json")), reverse=True)
    with open(os.path.join(prof, names[0])) as f:
        return json.load(f), names[0]


@pytest.fixture(scope="module")
def live():
    import isa_count
    import secp256k1_voi_amd as S
    if not os.path.exists(isa_count.OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    S.build()
    return isa_count.static_counts(S.LIB_PATH)


def test_loop_structure_and_magnitudes(live):
    g, k = live["k_verify_fast"], live["k_verify_fast_keyed"]
    # 128 doublings + 64 additions dominate the general ladder, 12 + 64 the keyed one
    assert 900 < g["valu_per_trip"]["doubling"] < 1100 and 1400 < g["valu_per_trip"]["addition"] < 1700
    assert 900 < k["valu_per_trip"]["doubling"] < 1100 and 1300 < k["valu_per_trip"]["addition"] < 1600
    assert g["valu_instr_static"] > 128 * g["valu_per_trip"]["doubling"] + 64 * g["valu_per_trip"]["addition"]
    assert k["valu_instr_static"] > 12 * k["valu_per_trip"]["doubling"] + 64 * k["valu_per_trip"]["addition"]
    assert 0.55 < g["mad_u64_u32_per_verify"] / g["valu_instr_static"] < 0.70


def test_committed_counts_describe_the_shipped_kernels(live):
    counts, name = newest_counts()
    for kname, skey in (("k_verify_fast", "static"), ("k_verify_fast_keyed", "static_keyed")):
        got = live[kname]["valu_instr_static"]
        assert kname in counts, "%s has no PMC entry for %s (bench.py cannot price it)" % (name, kname)
        pmc = counts[kname]["valu_instr_per_signature"]
        ref = counts[skey]["valu_instr_static"]
        assert abs(got - ref) <= 0.005 * ref, "%s: static count %d, %s says %d - refresh the counts (tools/collect_profiles_r04.sh)" % (kname, got, name, ref)
        assert abs(got - pmc) <= 0.005 * pmc, "%s: static count %d, PMC in %s %.0f - refresh the counts" % (kname, got, name, pmc)
        mad = live[kname]["mad_u64_u32_per_verify"]
        assert abs(mad - counts[skey]["mad_u64_u32_per_verify"]) <= 0.005 * mad
    assert counts.get("head"), "the counts do not name the commit they were taken at"


def test_bench_reports_staleness(live):
    sys.path.insert(0, ROOT)
    import bench
    import secp256k1_voi_amd as S
    counts, _ = bench.committed_counts()
    rep = bench.recount_shipped_binary(S.LIB_PATH, counts)
    assert rep["counts_stale"] is False and rep["static_recount"]["k_verify_fast_keyed"] == live["k_verify_fast_keyed"]["valu_instr_static"]
    doctored = json.loads(json.dumps(counts))
    doctored["k_verify_fast_keyed"]["valu_instr_per_signature"] *= 1.02
    assert bench.recount_shipped_binary(S.LIB_PATH, doctored)["counts_stale"] is True
