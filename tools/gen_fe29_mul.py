#!/usr/bin/env python3
"""Generates secp256k1_voi_amd/csrc/fe29_mul_gen.h: the 9 x 29-bit field multiply, square and
fused multiply-adds for gfx950 (v_mad_u64_u32 accumulation chains, one asm statement per
column tools/gen_chain.py), plus the 8x32 <-> 9x29 limb conversions.

Schedule (the 10x26 one with L = 9 limbs of W = 29 bits): column L-1 first, then for
k = 0..L-2 the high column L+k is reduced into the low column k with
2^261 = 2^5 (2^32 + 977) = R1 * 2^29 + R0, R0 = 0x7A20, R1 = 0x100.  81 products instead of
100, 8 reduction steps instead of 9; the price is headroom: a column holds at most 8 products
of full-size limbs, so (limb bound of a) * (limb bound of b) must stay below 2^61 (fe29.h).

    python tools/gen_fe29_mul.py > secp256k1_voi_amd/csrc/fe29_mul_gen.h
"""
from gen_chain import chain

L, W = 9, 29


def conversions():
    o = []
    o.append("// 8 x 32-bit little-endian words (value < 2^256) -> limbs\n")
    o.append("S2K_DEV fe29 fe29_from_words(const uint32_t w[8]) {\n  fe29 r;\n")
    for i in range(L):
        lo = W * i
        j, s = lo // 32, lo % 32
        if s == 0:
            e = f"w[{j}]"
        elif s + W <= 32 or j + 1 >= 8:
            e = f"(w[{j}] >> {s})"
        else:
            e = f"((w[{j}] >> {s}) | (w[{j + 1}] << {32 - s}))"
        if i < L - 1:
            e += " & F29_M"
        o.append(f"  r.n[{i}] = {e};\n")
    o.append("  return r;\n}\n")
    o.append("// canonical (fully normalised) limbs -> 8 x 32-bit little-endian words\n")
    o.append("S2K_DEV void fe29_to_words(uint32_t w[8], const fe29& a) {\n")
    for j in range(8):
        parts = []
        for i in range(L):
            lo = W * i - 32 * j          # position of limb i's bit 0 inside word j
            if lo >= 32 or lo + W <= 0:
                continue
            if lo == 0:
                parts.append(f"a.n[{i}]")
            elif lo > 0:
                parts.append(f"(a.n[{i}] << {lo})")
            else:
                parts.append(f"(a.n[{i}] >> {-lo})")
        o.append(f"  w[{j}] = {' | '.join(parts)};\n")
    o.append("}\n")
    return "".join(o)


def gen(name, kind, addend=False):
    """kind: 'mul' (A*B), 'sqr' (A^2), 'mulmul' (A*B + C*D), 'mulsqr' (A*B + C^2);
    addend: + E, a lazy value whose limbs are added to the columns (one multiply-add by 1 each),
    so that a sum leaves the reduction carry-propagated instead of needing its own pass"""
    first_sq = kind == "sqr"
    second = {"mul": None, "sqr": None, "mulmul": "mul", "mulsqr": "sqr"}[kind]
    o = []
    args = {"mul": "const fe29& A, const fe29& B", "sqr": "const fe29& A",
            "mulmul": "const fe29& A, const fe29& B, const fe29& C, const fe29& D",
            "mulsqr": "const fe29& A, const fe29& B, const fe29& C"}[kind]
    if addend:
        args += ", const fe29& E"
    o.append(f"S2K_DEV fe29 {name}({args}) {{\n")
    o.append("  const uint32_t* a = A.n;\n  uint32_t t[9];\n")
    if first_sq:
        o.append("  uint32_t a2[9];\n#pragma unroll\n  for (int i = 0; i < 9; ++i) a2[i] = a[i] * 2;\n")
    else:
        o.append("  const uint32_t* b = B.n;\n")
    if second:
        o.append("  const uint32_t* c_ = C.n;\n")
        if second == "sqr":
            o.append("  uint32_t c2[9];\n#pragma unroll\n  for (int i = 0; i < 9; ++i) c2[i] = c_[i] * 2;\n")
        else:
            o.append("  const uint32_t* d_ = D.n;\n")
    o.append("  const uint32_t R0 = F29_R0, R1 = F29_R1;\n")

    def terms(K, x, y, sq):
        out = []
        for i in range(L):
            j = K - i
            if not (0 <= j < L):
                continue
            if sq:
                if i < j:
                    out.append((f"{x}2[{i}]", f"{x}[{j}]", "v"))
                elif i == j:
                    out.append((f"{x}[{i}]", f"{x}[{i}]", "v"))
            else:
                out.append((f"{x}[{i}]", f"{y}[{j}]", "v"))
        return out

    def col(K):
        t = terms(K, "a", "b", first_sq)
        if addend and K < L:
            t.append((f"E.n[{K}]", "1", "n"))
        if second:
            t += terms(K, "c_" if second == "mul" else "c", "d_", second == "sqr")
        return t

    # the squared second operand uses c2[] / c_[] names
    def fix(ts):
        return [(x.replace("c2[", "c2[").replace("c[", "c_["), y.replace("c[", "c_["), k) for x, y, k in ts]

    # every product, including the first of each accumulator, is inside an asm chain: no C-level
    # 64-bit multiply is left for the compiler to narrow (see fe29_mul_small_norm in pt29.h)
    o.append("  uint64_t d, c;\n")
    o.append(chain("d", fix(col(L - 1)), init=True))
    o.append(f"  t[{L - 1}] = (uint32_t)d & F29_M;\n  d >>= {W};\n")
    o.append("  uint32_t u, uprev = 0;\n")
    for k in range(L - 1):
        o.append(chain("d", fix(col(L + k))))
        o.append(f"  u = (uint32_t)d & F29_M;\n  d >>= {W};\n")
        lo = []
        if k > 0:
            lo.append(("uprev", "R1", "s"))
        lo += fix(col(k))
        lo.append(("u", "R0", "s"))
        o.append(chain("c", lo, init=(k == 0)))
        o.append(f"  t[{k}] = (uint32_t)c & F29_M;\n  c >>= {W};\n  uprev = u;\n")
    o.append(chain("c", [("uprev", "R1", "s")]))
    o.append("  return fe29_mul_tail(t, c, d);\n}\n")
    return "".join(o)


def main():
    print("// fe29_mul_gen.h — GENERATED by tools/gen_fe29_mul.py; do not edit by hand.")
    print("// Included by fe29.h inside namespace s2k (needs fe29, F29_*, fe29_mul_tail).")
    print("// clang-format off")
    print(conversions())
    print(gen("fe29_mul", "mul"))
    print(gen("fe29_sqr", "sqr"))
    print(gen("fe29_mul_add_mul", "mulmul"))
    print(gen("fe29_mul_add_sqr", "mulsqr"))
    print(gen("fe29_mul_plus", "mul", addend=True))
    print(gen("fe29_sqr_plus", "sqr", addend=True))
    print("// clang-format on")


if __name__ == "__main__":
    main()
