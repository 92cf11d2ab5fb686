#!/usr/bin/env python3
"""Static instruction mix of k_verify_fast<ECDSA_KEYED> (the ladder over per-key tables), weighted by the loop trip counts
of the source: prologue once (lead point and its XYZZ addition), between rounds (x3) the accumulator's change of form
XYZZ -> Jacobian, four doublings (the doubling loop 3 x 4 = 12 times) and back, the table-addition loop 4 rounds x 8 chunks x
2 halves = 64 times, its chunk-loop tail 32 times, the epilogue (XYZZ -> Jacobian, generator part, verdict) once.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only secp256k1_voi_amd/csrc/engine.hip -o /tmp/engine.s
    python tools/isa_mix_keyed.py /tmp/engine.s

The regions are found from the compiler's loop comments: the depth-1 inner loop laid out before the main loop header is
the doubling loop (the block before it and the block after it are the two conversions), the depth-3 loop the addition.
Agrees with SQ_INSTS_VALU (112 104 per signature) to 0.3 %."""
import re
import sys

KERNEL = "_Z13k_verify_fastILi4EE"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]

    def idx(pred, lo=0):
        return next(i for i in range(lo, len(body)) if pred(body[i]))
    label = lambda l: re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
    dbl = idx(lambda l: "This Inner Loop Header: Depth=1" in l)
    conv_in = max(i for i in range(dbl) if label(body[i]))                      # XYZZ -> Jacobian, laid out right before the loop
    dbl_end = idx(label, dbl + 1)                                               # Jacobian -> XYZZ follows the loop
    main_hdr = idx(lambda l: "This Loop Header: Depth=1" in l, dbl)
    d3 = idx(lambda l: "Inner Loop Header: Depth=3" in l, main_hdr)
    d3_end = idx(lambda l: label(l) and "Depth=2" in l and "Parent" not in l, d3 + 1)
    d1_tail = idx(lambda l: label(l) and "Header=" in l and "Depth=1" in l, d3_end)
    post = idx(lambda l: label(l) and "Loop" not in l, d1_tail + 1)
    pro_end = min(i for i in range(conv_in + 1) if label(body[i]) and all("in Loop" in body[j] or j == conv_in for j in range(i, conv_in + 1) if label(body[j])))
    regions = [("prologue", 0, pro_end, 1), ("to_jacobian", conv_in, dbl, 3), ("doubling", dbl, dbl_end, 12), ("to_xyzz", dbl_end, main_hdr, 3),
               ("addition", d3 - 1, d3_end, 64), ("chunk_tail", d3_end, d1_tail, 32), ("epilogue", post, len(body), 1)]
    tot = mad = 0
    for name, a, b, w in regions:
        v = m = 0
        for l in body[a:b]:
            s = l.strip()
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                continue
            op = s.split()[0]
            if op.startswith("v_"):
                v += 1
                m += op.startswith("v_mad_u64_u32")
        print("%-11s lines %5d-%5d  x%-3d  %5d VALU, %5d v_mad_u64_u32" % (name, a, b, w, v, m))
        tot += v * w
        mad += m * w
    print("per signature: %d VALU instructions, %d v_mad_u64_u32 (%.1f %%)" % (tot, mad, 100.0 * mad / tot))


if __name__ == "__main__":
    main()
