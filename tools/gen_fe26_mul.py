#!/usr/bin/env python3
"""Generates secp256k1_voi_amd/csrc/fe26_mul_gen.h: fully unrolled 10x26 field multiply and
square for gfx950, as serial v_mad_u64_u32 accumulation chains grouped into one asm
statement per column (hipcc pads one wait state after every asm statement, so fewer,
longer statements are cheaper; left to itself it also splits columns into partial sums
joined by extra 64-bit adds).

Algorithm: the classic 10x26 schedule (column 9 first, then low column k together with
high column 10+k, folding with R0 = 0x3D10 and R1 = 0x400; see fe26.h).  The carry-out of
each v_mad_u64_u32 goes to VCC and is never read: the magnitude bounds guarantee that no
64-bit column sum overflows.

    python tools/gen_fe26_mul.py > secp256k1_voi_amd/csrc/fe26_mul_gen.h
"""


def chain(acc, prods, indent="  ", maxn=13):
    """asm statement(s): acc += sum(x*y) for (x, y, kind) in prods; kind 'v' or 's' for y.
    At most `maxn` products per statement (inline asm takes at most 30 operands)."""
    if not prods:
        return ""
    if len(prods) > maxn:
        return chain(acc, prods[:maxn], indent, maxn) + chain(acc, prods[maxn:], indent, maxn)
    lines = []
    ops = []
    n = 1
    for x, y, kind in prods:
        lines.append(f'"v_mad_u64_u32 %0, vcc, %{n}, %{n + 1}, %0')
        ops.append(f'"v"({x})')
        ops.append(f'"{kind}"({y})')
        n += 2
    body = "\\n\\t\"\n      ".join(lines) + '"'
    return f'{indent}asm({body}\n      : "+&v"({acc})\n      : {", ".join(ops)}\n      : "vcc");\n'


def gen_mul():
    o = []
    o.append("S2K_DEV fe26 fe26_mul(const fe26& A, const fe26& B) {\n")
    o.append("  const uint32_t* a = A.n;\n  const uint32_t* b = B.n;\n  uint32_t t[10];\n")
    o.append("  const uint32_t R0 = F26_R0, R1 = F26_R1;\n")
    o.append("  uint64_t d = (uint64_t)a[0] * b[9];\n")
    o.append(chain("d", [(f"a[{i}]", f"b[{9 - i}]", "v") for i in range(1, 10)]))
    o.append("  t[9] = (uint32_t)d & F26_M;\n  d >>= 26;\n")
    o.append("  uint64_t c = (uint64_t)a[0] * b[0];\n  uint32_t u, uprev = 0;\n")
    for k in range(9):
        hi = [(f"a[{i}]", f"b[{10 + k - i}]", "v") for i in range(k + 1, 10)]
        o.append(chain("d", hi))
        o.append("  u = (uint32_t)d & F26_M;\n  d >>= 26;\n")
        lo = []
        if k > 0:
            lo.append(("uprev", "R1", "s"))
        lo += [(f"a[{i}]", f"b[{k - i}]", "v") for i in range(0, k + 1) if not (k == 0 and i == 0)]
        lo.append(("u", "R0", "s"))
        o.append(chain("c", lo))
        o.append(f"  t[{k}] = (uint32_t)c & F26_M;\n  c >>= 26;\n  uprev = u;\n")
    o.append(chain("c", [("uprev", "R1", "s")]))
    o.append("  return fe26_mul_tail(t, c, d);\n}\n")
    return "".join(o)


def gen_sqr():
    o = []
    o.append("S2K_DEV fe26 fe26_sqr(const fe26& A) {\n")
    o.append("  const uint32_t* a = A.n;\n  uint32_t a2[10], t[10];\n")
    o.append("#pragma unroll\n  for (int i = 0; i < 10; ++i) a2[i] = a[i] * 2;\n")
    o.append("  const uint32_t R0 = F26_R0, R1 = F26_R1;\n")
    o.append("  uint64_t d = (uint64_t)a2[0] * a[9];\n")
    o.append(chain("d", [(f"a2[{i}]", f"a[{9 - i}]", "v") for i in range(1, 5)]))
    o.append("  t[9] = (uint32_t)d & F26_M;\n  d >>= 26;\n")
    o.append("  uint64_t c = (uint64_t)a[0] * a[0];\n  uint32_t u, uprev = 0;\n")
    for k in range(9):
        K = 10 + k
        hi = [(f"a2[{i}]", f"a[{K - i}]", "v") for i in range(k + 1, 10) if 2 * i < K]
        if K % 2 == 0:
            hi.append((f"a[{K // 2}]", f"a[{K // 2}]", "v"))
        o.append(chain("d", hi))
        o.append("  u = (uint32_t)d & F26_M;\n  d >>= 26;\n")
        lo = []
        if k > 0:
            lo.append(("uprev", "R1", "s"))
        lo += [(f"a2[{i}]", f"a[{k - i}]", "v") for i in range(0, k + 1) if 2 * i < k]
        if k % 2 == 0 and k > 0:
            lo.append((f"a[{k // 2}]", f"a[{k // 2}]", "v"))
        lo.append(("u", "R0", "s"))
        o.append(chain("c", lo))
        o.append(f"  t[{k}] = (uint32_t)c & F26_M;\n  c >>= 26;\n  uprev = u;\n")
    o.append(chain("c", [("uprev", "R1", "s")]))
    o.append("  return fe26_mul_tail(t, c, d);\n}\n")
    return "".join(o)


def gen_muladd(name, second):
    """a*b + (c*d | c^2) with ONE reduction: both products accumulate into the same column sums.
    Bound: m_a*m_b + m_c*m_d <= 64 keeps every 64-bit column sum below 2^64."""
    sq = second == "sqr"
    o = []
    if sq:
        o.append(f"S2K_DEV fe26 {name}(const fe26& A, const fe26& B, const fe26& C) {{\n")
    else:
        o.append(f"S2K_DEV fe26 {name}(const fe26& A, const fe26& B, const fe26& C, const fe26& D) {{\n")
    o.append("  const uint32_t* a = A.n;\n  const uint32_t* b = B.n;\n  const uint32_t* c_ = C.n;\n  uint32_t t[10];\n")
    if sq:
        o.append("  uint32_t c2[10];\n#pragma unroll\n  for (int i = 0; i < 10; ++i) c2[i] = c_[i] * 2;\n")
    else:
        o.append("  const uint32_t* d_ = D.n;\n")
    o.append("  const uint32_t R0 = F26_R0, R1 = F26_R1;\n")

    def second_terms(K):
        """products of the second operand pair landing in column K"""
        out = []
        if sq:
            for i in range(0, 10):
                j = K - i
                if 0 <= j <= 9 and i < j:
                    out.append((f"c2[{i}]", f"c_[{j}]", "v"))
            if K % 2 == 0 and K // 2 <= 9:
                out.append((f"c_[{K // 2}]", f"c_[{K // 2}]", "v"))
        else:
            for i in range(0, 10):
                j = K - i
                if 0 <= j <= 9:
                    out.append((f"c_[{i}]", f"d_[{j}]", "v"))
        return out

    o.append("  uint64_t d = (uint64_t)a[0] * b[9];\n")
    o.append(chain("d", [(f"a[{i}]", f"b[{9 - i}]", "v") for i in range(1, 10)]))
    o.append(chain("d", second_terms(9)))
    o.append("  t[9] = (uint32_t)d & F26_M;\n  d >>= 26;\n")
    o.append("  uint64_t c = (uint64_t)a[0] * b[0];\n  uint32_t u, uprev = 0;\n")
    for k in range(9):
        hi = [(f"a[{i}]", f"b[{10 + k - i}]", "v") for i in range(k + 1, 10)]
        o.append(chain("d", hi))
        o.append(chain("d", second_terms(10 + k)))
        o.append("  u = (uint32_t)d & F26_M;\n  d >>= 26;\n")
        lo = []
        if k > 0:
            lo.append(("uprev", "R1", "s"))
        lo += [(f"a[{i}]", f"b[{k - i}]", "v") for i in range(0, k + 1) if not (k == 0 and i == 0)]
        o.append(chain("c", lo + second_terms(k) + [("u", "R0", "s")]))
        o.append(f"  t[{k}] = (uint32_t)c & F26_M;\n  c >>= 26;\n  uprev = u;\n")
    o.append(chain("c", [("uprev", "R1", "s")]))
    o.append("  return fe26_mul_tail(t, c, d);\n}\n")
    return "".join(o)


def main():
    print("// fe26_mul_gen.h — GENERATED by tools/gen_fe26_mul.py; do not edit by hand.")
    print("// Included by fe26.h inside namespace s2k (needs fe26, F26_*, fe26_mul_tail).")
    print("// clang-format off")
    print(gen_mul())
    print(gen_sqr())
    print(gen_muladd("fe26_mul_add_mul", "mul"))
    print(gen_muladd("fe26_mul_add_sqr", "sqr"))
    print("// clang-format on")


if __name__ == "__main__":
    main()
