#!/usr/bin/env python3
"""Static instruction mix of k_verify_fast<ECDSA>, weighted by loop trip counts.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only secp256k1_voi_amd/csrc/engine.hip -o /tmp/engine.s
    python tools/isa_mix.py /tmp/engine.s

Regions are cut at the kernel's loop headers (the compiler's "Loop Header" comments); trip counts are
those of the source: table forward loop 7, table backward loop 7, ladder 32 x (4 doublings, 2 additions),
generator part GT_WINDOWS.  Issue cost per class from tools/valu_rates2.hip (MI355X): 4 cycles per
wave64 instruction for v_mad_u64_u32, 64-bit shifts, VOP3 integer ops; 2 cycles for VOP1/VOP2 add / and /
xor / mov in a pure stream - but 4 in a stream mixed with multiplies, which is what this kernel is.
"""
import re
import sys
from collections import Counter

KERNEL = "_Z13k_verify_fastILi0EE"


def main():
    path = sys.argv[1]
    gt_windows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(":") or (l.startswith(KERNEL) and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    # Loop membership of every basic block from the compiler's block comments.
    weights = [1] * len(body)
    names = ["straight"] * len(body)
    cur = None                     # loop (header block name) the current block belongs to
    loops1, loops2 = [], []        # depth-1 / depth-2 loops in order of appearance
    block_loop = []
    for j, l in enumerate(body):
        is_block = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
        if is_block:
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d)", l)
            if m:
                cur = m.group(1)
            elif "Loop Header: Depth=1" in l:
                cur = l.split(":")[0].strip().lstrip(".L")
                loops1.append(cur)
            elif "Parent Loop" in l and j + 1 < len(body) and "Loop Header: Depth=2" in body[j + 1]:
                cur = l.split(":")[0].strip().lstrip(".L")
                loops2.append(cur)
            else:
                cur = None
        block_loop.append(cur)
    assert len(loops1) == 4 and len(loops2) == 2, (loops1, loops2)
    trip = {loops1[0]: (7, "table_fwd"), loops1[1]: (7, "table_bwd"), loops1[2]: (32, "ladder_outer"),
            loops1[3]: (gt_windows, "generator"), loops2[0]: (32 * 4, "doubling"), loops2[1]: (32 * 2, "addition")}
    for j, c in enumerate(block_loop):
        if c in trip:
            weights[j], names[j] = trip[c]
    hist = Counter()
    per_region = Counter()
    for j, l in enumerate(body):
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"^(v_|s_|global_|buffer_|ds_|flat_|scratch_)", op):
            continue
        hist[op] += weights[j]
        per_region[names[j]] += weights[j] if op.startswith("v_") else 0
    valu = {k: v for k, v in hist.items() if k.startswith("v_")}
    total = sum(valu.values())
    mad = sum(v for k, v in valu.items() if k.startswith("v_mad_u64_u32"))
    fast2 = sum(v for k, v in valu.items() if re.match(r"v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|mov_b32|not_b32)(_e32)?$", k))
    print("VALU instructions per signature (static, trip-weighted): %d" % total)
    print("  v_mad_u64_u32 %d (%.1f%%)   2-cycle-capable VOP1/VOP2 %d (%.1f%%)   other 4-cycle VALU %d (%.1f%%)"
          % (mad, 100 * mad / total, fast2, 100 * fast2 / total, total - mad - fast2, 100 * (total - mad - fast2) / total))
    print("  issue cycles per signature-wave: all at 4 cycles %d; with VOP1/VOP2 at 2 cycles %d" % (4 * total, 4 * (total - fast2) + 2 * fast2))
    print("per region:", dict(per_region))
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:25]:
        print("  %-22s %8d  %5.1f%%" % (k, v, 100 * v / total))
    other = {k: v for k, v in hist.items() if not k.startswith("v_")}
    print("non-VALU:", {k: v for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]})


if __name__ == "__main__":
    main()
