#!/usr/bin/env python3
"""Shared piece of the multiplication generators (gen_fe29_mul.py, gen_sc26_mul.py): serial
v_mad_u64_u32 accumulation chains as ONE inline-asm statement per column.  hipcc pads a wait
state after every asm statement and, left to itself, splits a column into partial sums joined
by extra 64-bit adds; one statement per chain avoids both.  The carry-out of v_mad_u64_u32 goes
to VCC and is never read: the limb bounds guarantee that no 64-bit column sum overflows."""


def chain(acc, prods, indent="  ", maxn=13, init=False):
    """asm statement(s): acc += sum(x*y) for (x, y, kind) in prods; kind 'v', 's' or 'n' for y.
    At most `maxn` products per statement (inline asm takes at most 30 operands).
    init=True: the first product initialises acc (addend 0) instead of accumulating."""
    if not prods:
        return ""
    if len(prods) > maxn:
        return chain(acc, prods[:maxn], indent, maxn, init) + chain(acc, prods[maxn:], indent, maxn)
    lines = []
    ops = []
    n = 1
    for k, (x, y, kind) in enumerate(prods):
        addend = "0" if (init and k == 0) else "%0"
        lines.append(f'"v_mad_u64_u32 %0, vcc, %{n}, %{n + 1}, {addend}')
        ops.append(f'"v"({x})')
        ops.append(f'"{kind}"({y})')
        n += 2
    body = "\\n\\t\"\n      ".join(lines) + '"'
    out = "=&v" if init else "+&v"
    return f'{indent}asm({body}\n      : "{out}"({acc})\n      : {", ".join(ops)}\n      : "vcc");\n'
