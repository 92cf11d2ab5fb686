#!/usr/bin/env python3
"""Static instruction mix of k_verify_fast<ECDSA_KEYED> (the ladder over per-key tables), weighted by the loop trip counts
of the source: prologue once, the doubling loop 4 x 3 = 12 times, the table-addition loop 4 rounds x 8 chunks x 2 halves =
64 times, its chunk-loop tail 32 times, the epilogue (generator part, verdict) once.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only secp256k1_voi_amd/csrc/engine.hip -o /tmp/engine.s
    python tools/isa_mix_keyed.py /tmp/engine.s

The regions are found from the compiler's loop comments: the depth-1 inner loop laid out before the main loop header is
the doubling loop, the depth-3 loop the addition.  Agrees with SQ_INSTS_VALU (117 012 per signature) to 0.3 %."""
import re
import sys

KERNEL = "_Z13k_verify_fastILi4EE"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]

    def idx(pred, lo=0):
        return next(i for i in range(lo, len(body)) if pred(body[i]))
    dbl = idx(lambda l: "This Inner Loop Header: Depth=1" in l)
    main_hdr = idx(lambda l: "This Loop Header: Depth=1" in l, dbl)
    pre_end = max(i for i in range(dbl) if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", body[i]) and "in Loop" in body[i]) if any(
        "in Loop" in body[i] for i in range(dbl)) else dbl
    d3 = idx(lambda l: "Inner Loop Header: Depth=3" in l, main_hdr)
    d3_end = idx(lambda l: re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l) and "Depth=2" in l and "Parent" not in l, d3 + 1)
    d1_tail = idx(lambda l: re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l) and "Header=" in l and "Depth=1" in l, d3_end)
    post = idx(lambda l: re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l) and "Loop" not in l, d1_tail + 1)
    regions = [("prologue", 0, pre_end, 1), ("doubling", pre_end, main_hdr, 12), ("addition", d3 - 1, d3_end, 64),
               ("chunk_tail", d3_end, d1_tail, 32), ("epilogue", post, len(body), 1)]
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
        print("%-10s lines %5d-%5d  x%-3d  %5d VALU, %5d v_mad_u64_u32" % (name, a, b, w, v, m))
        tot += v * w
        mad += m * w
    print("per signature: %d VALU instructions, %d v_mad_u64_u32 (%.1f %%)" % (tot, mad, 100.0 * mad / tot))


if __name__ == "__main__":
    main()
