#!/usr/bin/env python3
"""Trip-weighted static instruction counts of the two ladder kernels, taken from the SHIPPED code object.

    python tools/isa_count.py [path/to/libsecp256k1_voi_amd.so]      -> JSON on stdout

The library's gfx950 code objects are unbundled with `llvm-objdump --offloading`, the kernel is disassembled with
`llvm-objdump -d`, and the loops are recovered from the BACKWARD BRANCHES of the disassembly (a branch to a lower
address closes a loop [target, branch]; nesting is containment).  Trip counts are those of the source:

  k_verify_fast<ECDSA>        table forward 7 | table backward 7 | ladder 32 x (doubling loop 4, addition loop 2) | generator part GT_WINDOWS
  k_verify_fast<ECDSA_KEYED>  main loop of 4 rounds, entered past its first blocks (the change of form and the doubling loop
                              run between rounds only: 3 times; doubling loop 4 per time), chunk loop 8, addition loop 2

This is what ties the roofline's instruction counts (profiles/r04_valu_counts.json, PMC) to the binary that is measured:
tests/test_counts_cpu.py recounts the built library and compares, bench.py recounts the library it loaded.
(tools/isa_mix.py / isa_mix_keyed.py do the same on the compiler's assembly listing, with its loop comments; they need a
two-minute compile, this needs none.)"""
import json
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "..", "secp256k1_voi_amd", "libsecp256k1_voi_amd.so")


def code_objects(lib):
    """Unbundles the gfx950 code objects of `lib` into a temporary directory; returns (tmpdir, [paths])."""
    tmp = tempfile.mkdtemp(prefix="s2k_isa_")
    work = os.path.join(tmp, "lib.so")
    os.symlink(os.path.abspath(lib), work)
    subprocess.run([OBJDUMP, "--offloading", work], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return tmp, sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def disassemble(lib, mangled_prefix):
    """[(offset, opcode, branch_target_offset or None)] of the kernel whose mangled name starts with `mangled_prefix`."""
    tmp, objs = code_objects(lib)
    try:
        for obj in objs:
            syms = subprocess.run([OBJDUMP, "-t", obj], capture_output=True, text=True, check=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if l.split() and l.split()[-1].startswith(mangled_prefix) and " F " in l]
            names = [n for n in names if not n.endswith(".kd")]
            if not names:
                continue
            out = subprocess.run([OBJDUMP, "-d", "--disassemble-symbols=" + names[0], obj], capture_output=True, text=True, check=True).stdout
            ins, base = [], None
            for l in out.splitlines():
                m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-Fa-f]+):", l) or re.match(r"^\s+(\S+)\s*//\s*([0-9A-Fa-f]+):", l)
                if not m:
                    continue
                addr = int(m.group(2), 16)
                if base is None:
                    base = addr
                tgt = None
                if m.group(1).startswith(("s_cbranch", "s_branch")):
                    t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", l)
                    tgt = int(t.group(1), 16) if t else 0      # (no offset: the kernel's first instruction)
                ins.append((addr - base, m.group(1), tgt))
            return ins
        raise RuntimeError("kernel %s* not found in %s" % (mangled_prefix, lib))
    finally:
        subprocess.run(["rm", "-rf", tmp])


def loops_of(ins):
    """Loops as (lo, hi) offset intervals from backward branches, merged per header, sorted by lo."""
    by_head = {}
    for off, op, tgt in ins:
        if tgt is not None and tgt <= off:
            by_head[tgt] = max(by_head.get(tgt, 0), off)
    lp = sorted(by_head.items())
    # cold blocks laid out behind the kernel's end branch BACK into the straight-line code: such "loops" cross each
    # other (overlap without containment), which loops proper never do
    crossing = lambda a, b: a[0] < b[0] <= a[1] < b[1] or b[0] < a[0] <= b[1] < a[1]
    return [l for l in lp if not any(crossing(l, o) for o in lp)]


def count(ins, weight_of):
    valu = mad = 0
    for off, op, _ in ins:
        if op.startswith("v_"):
            w = weight_of(off)
            valu += w
            mad += w if op.startswith("v_mad_u64_u32") else 0
    return valu, mad


def general(lib, gt_windows=12):
    ins = disassemble(lib, "_Z13k_verify_fastILi0EE")
    lp = loops_of(ins)
    outer = [l for l in lp if not any(o[0] <= l[0] and l[1] <= o[1] and o != l for o in lp)]
    inner = [l for l in lp if l not in outer]
    assert len(outer) == 4 and len(inner) == 2, ("unexpected loop structure of k_verify_fast<ECDSA>", outer, inner)
    assert all(outer[2][0] <= l[0] and l[1] <= outer[2][1] for l in inner), "inner loops are not inside the ladder loop"
    trips = {outer[0]: 7, outer[1]: 7, outer[2]: 32, outer[3]: gt_windows, inner[0]: 32 * 4, inner[1]: 32 * 2}

    def weight(off):
        for l in inner + outer:                       # innermost first
            if l[0] <= off <= l[1]:
                return trips[l]
        return 1
    valu, mad = count(ins, weight)
    regions = {}
    for name, l in (("table_fwd", outer[0]), ("table_bwd", outer[1]), ("doubling", inner[0]), ("addition", inner[1]), ("generator", outer[3])):
        regions[name] = sum(1 for off, op, _ in ins if op.startswith("v_") and l[0] <= off <= l[1])
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad, "valu_per_trip": regions, "instructions": len(ins)}


def keyed(lib):
    ins = disassemble(lib, "_Z13k_verify_fastILi4EE")
    lp = loops_of(ins)
    outer = [l for l in lp if not any(o[0] <= l[0] and l[1] <= o[1] and o != l for o in lp)]
    assert len(outer) == 1, ("unexpected loop structure of k_verify_fast<ECDSA_KEYED>", lp)
    main = outer[0]
    inside = [l for l in lp if l != main]
    depth2 = [l for l in inside if not any(o[0] <= l[0] and l[1] <= o[1] and o != l for o in inside)]
    depth3 = [l for l in inside if l not in depth2]
    assert len(depth2) == 2 and len(depth3) == 1, ("unexpected loop structure of k_verify_fast<ECDSA_KEYED>", lp)
    dbl, chunk = depth2
    add = depth3[0]
    assert chunk[0] <= add[0] and add[1] <= chunk[1], "the addition loop is not inside the chunk loop"
    # the main loop is entered past its first blocks: the forward branch from before the loop into it names the header of
    # a ROUND; what lies before that inside the loop (change of form, doubling loop, change back) runs between rounds only
    # (a second branch from the prologue, to the change of form itself, is the compiler's copy of the `if (round)` test:
    # never taken, the first round is round 0)
    entry = sorted(set(tgt for off, op, tgt in ins if tgt is not None and off < main[0] and dbl[1] < tgt <= chunk[0]))
    assert len(entry) == 1, ("main loop entry not found", entry, main, dbl, chunk)
    entry = entry[0]

    def weight(off):
        if add[0] <= off <= add[1]:
            return 64
        if chunk[0] <= off <= chunk[1]:
            return 32
        if dbl[0] <= off <= dbl[1]:
            return 12
        if main[0] <= off <= main[1]:
            return 3 if off < entry else 4
        return 1
    valu, mad = count(ins, weight)
    per = lambda l: sum(1 for off, op, _ in ins if op.startswith("v_") and l[0] <= off <= l[1])
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad,
            "valu_per_trip": {"doubling": per(dbl), "addition": per(add)}, "instructions": len(ins)}


def static_counts(lib=DEFAULT_LIB, gt_windows=12):
    return {"k_verify_fast": general(lib, gt_windows), "k_verify_fast_keyed": keyed(lib)}


if __name__ == "__main__":
    print(json.dumps(static_counts(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB), indent=1))
