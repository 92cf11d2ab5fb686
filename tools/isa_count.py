#!/usr/bin/env python3
"""Trip-weighted static instruction counts of the two ladder kernels, taken from the SHIPPED code object.

    python tools/isa_count.py [path/to/libsecp256k1_voi_amd.so]      -> JSON on stdout

The library's gfx950 code objects are unbundled with `llvm-objdump --offloading`, the kernel is disassembled with
`llvm-objdump -d`, and the loops are recovered from the control-flow graph of the disassembly (basic blocks; a loop is a
strongly connected component, the loops inside it are the components left when the edges into its entries are cut - the
code layout does not matter, nor does a second entry into a loop).  Trip counts are
those of the source:

  k_verify_fast<ECDSA>        table forward 7 | table backward 7 | ladder 32 x (doubling loop 4, addition loop 2) | generator part GT_WINDOWS - 1
                              (the last generator addition stands outside the loop)
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


class Cfg:
    """Basic blocks, dominators and natural loops of one kernel's disassembly."""

    def __init__(self, ins):
        self.ins = ins
        offs = [o for o, _, _ in ins]
        index = {o: i for i, o in enumerate(offs)}
        leaders = {0}
        for i, (off, op, tgt) in enumerate(ins):
            if tgt is not None:
                leaders.add(index[tgt])
                if i + 1 < len(ins):
                    leaders.add(i + 1)
            elif op.startswith(("s_endpgm", "s_setpc", "s_swappc")) and i + 1 < len(ins):
                leaders.add(i + 1)
        starts = sorted(leaders)
        self.blocks = [(starts[k], starts[k + 1] if k + 1 < len(starts) else len(ins)) for k in range(len(starts))]
        self.block_of = {}
        for b, (lo, hi) in enumerate(self.blocks):
            for i in range(lo, hi):
                self.block_of[i] = b
        nb = len(self.blocks)
        self.succ = [[] for _ in range(nb)]
        for b, (lo, hi) in enumerate(self.blocks):
            off, op, tgt = ins[hi - 1]
            if tgt is not None:
                self.succ[b].append(self.block_of[index[tgt]])
            if not op.startswith(("s_branch", "s_endpgm", "s_setpc")) and hi < len(ins):
                self.succ[b].append(self.block_of[hi])
        self.pred = [[] for _ in range(nb)]
        for b in range(nb):
            for t in self.succ[b]:
                self.pred[t].append(b)
        # Loop forest from strongly connected components (works for loops with two entries too - the keyed ladder's round
        # loop is entered past its first blocks AND, by a never-taken guard, at them): a nontrivial SCC is a loop; its
        # entries are the blocks with a predecessor outside; with the edges into the entries cut, the SCCs inside are the
        # nested loops.
        self.loops = []                                   # dicts: blocks (set), entries (sorted list), parent (index or None)
        self._nest(set(range(nb)), set(), None)

    def _sccs(self, nodes, cut):
        """nontrivial SCCs of the graph on `nodes` without the edges into `cut` (Tarjan, iterative)"""
        idx, low, on, stack, out, counter = {}, {}, set(), [], [], [0]
        for root in sorted(nodes):
            if root in idx:
                continue
            work = [(root, iter([t for t in self.succ[root] if t in nodes and t not in cut]))]
            idx[root] = low[root] = counter[0]
            counter[0] += 1
            stack.append(root)
            on.add(root)
            while work:
                v, it = work[-1]
                advanced = False
                for t in it:
                    if t not in idx:
                        idx[t] = low[t] = counter[0]
                        counter[0] += 1
                        stack.append(t)
                        on.add(t)
                        work.append((t, iter([x for x in self.succ[t] if x in nodes and x not in cut])))
                        advanced = True
                        break
                    elif t in on:
                        low[v] = min(low[v], idx[t])
                if advanced:
                    continue
                work.pop()
                if work:
                    low[work[-1][0]] = min(low[work[-1][0]], low[v])
                if low[v] == idx[v]:
                    comp = set()
                    while True:
                        x = stack.pop()
                        on.discard(x)
                        comp.add(x)
                        if x == v:
                            break
                    if len(comp) > 1 or v in [t for t in self.succ[v] if t not in cut]:
                        out.append(comp)
        return out

    def _nest(self, nodes, cut, parent):
        for comp in sorted(self._sccs(nodes, cut), key=min):
            entries = sorted(b for b in comp if b == 0 or any(p not in comp for p in self.pred[b]))
            self.loops.append({"blocks": comp, "entries": entries, "parent": parent})
            self._nest(comp, cut | set(entries), len(self.loops) - 1)

    def children(self, k):
        return [i for i, l in enumerate(self.loops) if l["parent"] == k]

    def top_level(self):
        return self.children(None)

    def weights(self, trips, headers=None):
        """block -> product of the trip counts of the loops around it.  A loop whose exit test sits in the middle (the blocks
        behind the test run one time fewer) counts trip - 1 for the blocks reached from the exiting block's in-loop successor
        before the loop's header comes round again.  `headers`: loop index -> header block, for loops with several entries."""
        w = [1] * len(self.blocks)
        for k, l in enumerate(self.loops):
            body, t = l["blocks"], trips[k]
            head = (headers or {}).get(k, l["entries"][0] if len(l["entries"]) == 1 else None)
            assert head is not None, "loop with several entries and no header given"
            exiting = [b for b in body if any(s not in body for s in self.succ[b])]
            late = set()
            if len(exiting) == 1:
                stack = [s_ for s_ in self.succ[exiting[0]] if s_ in body and s_ != head]
                while stack:
                    x = stack.pop()
                    if x in late or x == head:
                        continue
                    late.add(x)
                    stack.extend(y for y in self.succ[x] if y in body)
            for b in body:
                w[b] *= (t - 1) if b in late else t
        return w

    def count(self, w, blocks=None):
        valu = mad = 0
        for b, (lo, hi) in enumerate(self.blocks):
            if blocks is not None and b not in blocks:
                continue
            for i in range(lo, hi):
                op = self.ins[i][1]
                if op.startswith("v_"):
                    valu += w[b]
                    mad += w[b] if op.startswith("v_mad_u64_u32") else 0
        return valu, mad

    def valu_in(self, body):
        return sum(1 for b in body for i in range(*self.blocks[b]) if self.ins[i][1].startswith("v_"))


def general(lib, gt_windows=None):
    if gt_windows is None:
        gt_windows = generator_windows(lib)
    g = Cfg(disassemble(lib, "_Z13k_verify_fastILi0EE"))
    top = g.top_level()
    assert len(top) == 4, ("unexpected loop structure of k_verify_fast<ECDSA>", [g.loops[k]["entries"] for k in top])
    inner = g.children(top[2])
    assert len(inner) == 2 and not g.children(top[0]) and not g.children(top[1]) and not g.children(top[3]), ("unexpected inner loops", inner)
    trips = {top[0]: 7, top[1]: 7, top[2]: 32, top[3]: gt_windows - 1, inner[0]: 4, inner[1]: 2}   # (the last generator addition stands apart)
    assert len(trips) == len(g.loops), "loops without a trip count"
    valu, mad = g.count(g.weights(trips))
    regions = {name: g.valu_in(g.loops[k]["blocks"]) for name, k in (("table_fwd", top[0]), ("table_bwd", top[1]), ("doubling", inner[0]),
                                                                      ("addition", inner[1]), ("generator", top[3]))}
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad, "valu_per_trip": regions, "instructions": len(g.ins)}


def keyed(lib):
    g = Cfg(disassemble(lib, "_Z13k_verify_fastILi4EE"))
    top = g.top_level()
    assert len(top) == 1, ("unexpected loop structure of k_verify_fast<ECDSA_KEYED>", [g.loops[k]["entries"] for k in top])
    kids = g.children(top[0])
    assert len(kids) == 2, ("unexpected loops inside the round loop", kids)
    chunk = [k for k in kids if g.children(k)]
    dbl = [k for k in kids if not g.children(k)]
    assert len(chunk) == 1 and len(dbl) == 1, "cannot tell the doubling loop from the chunk loop"
    add = g.children(chunk[0])
    assert len(add) == 1 and not g.children(add[0])
    trips = {top[0]: 4, dbl[0]: 4, chunk[0]: 8, add[0]: 2}
    assert len(trips) == len(g.loops), "loops without a trip count"
    # a ROUND starts where the chunk loop is entered from: the block in front of the chunk loop's header
    ch = g.loops[chunk[0]]
    round_head = [p for e in ch["entries"] for p in g.pred[e] if p not in ch["blocks"]]
    assert len(set(round_head)) == 1, ("round header not found", round_head)
    w = g.weights(trips, headers={top[0]: round_head[0]})
    # the doubling loop runs between rounds only: 3 x 4
    assert {w[b] for b in g.loops[dbl[0]]["blocks"]} == {12} and {w[b] for b in g.loops[add[0]]["blocks"]} == {64}, "trip weights are not the source's"
    valu, mad = g.count(w)
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad,
            "valu_per_trip": {"doubling": g.valu_in(g.loops[dbl[0]]["blocks"]), "addition": g.valu_in(g.loops[add[0]]["blocks"])},
            "instructions": len(g.ins)}


def keyset(lib):
    """k_verify_fast<ECDSA_KEYSET>: the ladder over a key set's 32-chunk tables - one chunk loop (32) around the addition loop (2)"""
    g = Cfg(disassemble(lib, "_Z13k_verify_fastILi8EE"))
    top = g.top_level()
    assert len(top) == 1, ("unexpected loop structure of k_verify_fast<ECDSA_KEYSET>", [g.loops[k]["entries"] for k in top])
    add = g.children(top[0])
    assert len(add) == 1 and not g.children(add[0])
    trips = {top[0]: 32, add[0]: 2}
    assert len(trips) == len(g.loops), "loops without a trip count"
    valu, mad = g.count(g.weights(trips))
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad, "valu_per_trip": {"addition": g.valu_in(g.loops[add[0]]["blocks"])},
            "instructions": len(g.ins)}


def keyset_joint(lib):
    """k_verify_fast<ECDSA_KEYSET_JOINT>: one loop over the 32 digit positions, one table addition each"""
    g = Cfg(disassemble(lib, "_Z13k_verify_fastILi9EE"))
    top = g.top_level()
    assert len(top) == 1 and not g.children(top[0]), ("unexpected loop structure of k_verify_fast<ECDSA_KEYSET_JOINT>", [g.loops[k]["entries"] for k in top])
    valu, mad = g.count(g.weights({top[0]: 32}))
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad, "valu_per_trip": {"addition": g.valu_in(g.loops[top[0]]["blocks"])},
            "instructions": len(g.ins)}


def keyset_joint_wide(lib, w):
    """k_verify_fast<ECDSA_KEYSET_JOINT5 / JOINT6>: one loop over the 26 / 22 digit positions, one table addition each"""
    g = Cfg(disassemble(lib, "_Z13k_verify_fastILi%dEE" % (10 if w == 5 else 11)))
    top = g.top_level()
    assert len(top) == 1 and not g.children(top[0]), ("unexpected loop structure of the wide joint ladder", [g.loops[k]["entries"] for k in top])
    valu, mad = g.count(g.weights({top[0]: (128 + w - 1) // w}))
    return {"valu_instr_static": valu, "mad_u64_u32_per_verify": mad, "valu_per_trip": {"addition": g.valu_in(g.loops[top[0]]["blocks"])},
            "instructions": len(g.ins)}


def generator_windows(lib):
    """ceil(256 / window bits) of THIS library (s2k_generator_window_bits: a host function, no GPU needed)"""
    import ctypes
    bits = int(ctypes.CDLL(os.path.abspath(lib)).s2k_generator_window_bits())
    return (256 + bits - 1) // bits


def static_counts(lib=DEFAULT_LIB, gt_windows=None):
    if gt_windows is None:
        gt_windows = generator_windows(lib)
    return {"k_verify_fast": general(lib, gt_windows), "k_verify_fast_keyed": keyed(lib), "k_verify_fast_keyset": keyset(lib),
            "k_verify_fast_keyset_joint": keyset_joint(lib), "k_verify_fast_keyset_joint5": keyset_joint_wide(lib, 5),
            "k_verify_fast_keyset_joint6": keyset_joint_wide(lib, 6)}


if __name__ == "__main__":
    print(json.dumps(static_counts(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB), indent=1))
