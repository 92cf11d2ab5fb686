"""Lane-level model of csrc/fe29r.h / pt29r.h: the 9x29 lazy field with ONE LIMB PER LANE of a 16-lane DPP row, and the
complete projective group law (Renes-Costello-Batina, a = 0, b3 = 21; the reference's addComplete / doubleComplete,
point_projective.go:24,208) on it, one product per row, four rows per wavefront.

A "register" is a list of 64 integers (one per lane).  The data-movement primitives are modelled exactly as the ISA defines
them (row_shr / row_shl / row_newbcast with bound_ctrl:1 inside 16-lane rows; v_permlane16_swap / v_permlane32_swap), and
every 32- / 64-bit width the device code relies on is asserted, so running the model on extremal lazy inputs doubles as the
overflow proof of the schedule.  The device code is a transcription of these functions (same names)."""

P = 2**256 - 2**32 - 977
W = 29
M = (1 << W) - 1
M8 = (1 << 24) - 1
R0, R1 = 0x7A20, 0x100
P_LIMBS = [0x1FFFFC2F, 0x1FFFFFF7] + [M] * 6 + [M8]
U32, U64 = 1 << 32, 1 << 64
LANES = 64


def lane_const(f):
    return [f(l & 15, l >> 4) for l in range(LANES)]


# ---- data movement -------------------------------------------------------------------------------
def row_shr(v, n):      # lane j of a row reads lane j - n of the row; 0 when out of the row (bound_ctrl:1)
    return [v[l - n] if (l & 15) >= n else 0 for l in range(LANES)]


def row_shl(v, n):      # lane j reads lane j + n
    return [v[l + n] if (l & 15) + n <= 15 else 0 for l in range(LANES)]


def row_newbcast(v, n):  # every lane of a row reads lane n of the row
    return [v[(l & ~15) | n] for l in range(LANES)]


def permlane16_swap(a, b):   # odd rows of a <-> even rows of b
    a, b = list(a), list(b)
    for base in (0, 32):
        for j in range(16):
            a[base + 16 + j], b[base + j] = b[base + j], a[base + 16 + j]
    return a, b


def permlane32_swap(a, b):   # rows 2, 3 of a <-> rows 0, 1 of b
    a, b = list(a), list(b)
    for j in range(32):
        a[32 + j], b[j] = b[j], a[32 + j]
    return a, b


def bcast_rows(v):
    """the four rows of v, each replicated into all four rows: 3 moves + 3 swaps"""
    e, o = permlane16_swap(v, v)          # (v0, v0, v2, v2), (v1, v1, v3, v3)
    r0, r2 = permlane32_swap(e, e)
    r1, r3 = permlane32_swap(o, o)
    return r0, r1, r2, r3


def by_row(r0, r1, r2, r3):
    rows = (r0, r1, r2, r3)
    return [rows[l >> 4][l] for l in range(LANES)]


# ---- per-lane constants (registers set up once per kernel from the lane id) ----------------------
K_R0 = lane_const(lambda j, r: R0 if j <= 6 else 0)          # fold of column j + 9 (<= 15) into column j
K_R1 = lane_const(lambda j, r: R1 if 1 <= j <= 7 else 0)     # fold of column j + 8 (9 .. 15) into column j
K_16 = lane_const(lambda j, r: R0 if j == 7 else (R1 if j == 8 else 0))    # column 16 (row-uniform) into columns 7, 8
K_17 = lane_const(lambda j, r: R0 if j == 8 else (R1 if j == 9 else 0))    # column 17 into columns 8, 9 (column 9 is folded again)
K_FOLD2 = lane_const(lambda j, r: R0 if j == 0 else (R1 if j == 1 else 0))
K_TOP = lane_const(lambda j, r: 0x3D1 if j == 0 else (8 if j == 1 else 0))   # 2^256 = 0x3D1 + 8 * 2^29
K_WMASK = lane_const(lambda j, r: M if j < 8 else (M8 if j == 8 else 0))
K_SHIFT = lane_const(lambda j, r: 29 if j < 8 else 24)
K_LE8 = lane_const(lambda j, r: U32 - 1 if j <= 8 else 0)
K_LT8 = lane_const(lambda j, r: U32 - 1 if j < 8 else 0)
K_P = lane_const(lambda j, r: P_LIMBS[j] if j <= 8 else 0)


def fer_from_int(x):
    """the same element in all four rows; lanes 9..15 of a row are zero (invariant of every fer value)"""
    return lane_const(lambda j, r: ((x >> (W * j)) & (M if j < 8 else (1 << 32) - 1)) if j <= 8 else 0)


def fer_value(v, row=0):
    assert all(v[16 * row + j] == 0 for j in range(9, 16)), "lanes 9..15 must stay zero"
    return sum(v[16 * row + j] << (W * j) for j in range(9))


def fer_add(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert max(r) < U32
    return r


def fer_negate(a, w):
    """-a for a whose limbs are at most (w + 1) * p's limbs; (w + 1) * p - a limb by limb"""
    r = [K_P[l] * (w + 1) - a[l] for l in range(LANES)]
    assert min(r) >= 0 and max(r) < U32, "negate bias too small"
    return r


def fer_shl(a, sh):     # limb-wise shift by a per-lane amount (v_lshlrev_b32)
    r = [x << s for x, s in zip(a, sh)]
    assert max(r) < U32
    return r


def _tail(acc, h):
    """columns 0..15 in the lanes (64-bit), column 16 in h (row-uniform, 64-bit) -> 9 limbs"""
    assert max(acc) < U64 and max(h) < U64
    lo = [x & M for x in acc]
    c1 = [(x >> W) & M for x in acc]          # v_alignbit_b32 + v_and
    c2 = [x >> (2 * W) for x in acc]          # acc_hi >> 26
    assert max(c2) < 64
    t = [a + b + c for a, b, c in zip(lo, row_shr(c1, 1), row_shr(c2, 2))]            # < 2^30 + 64
    hl = [x & M for x in h]
    hc = [x >> W for x in h]
    assert max(hc) < U32
    t16 = [a + b + c for a, b, c in zip(hl, row_newbcast(c1, 15), row_newbcast(c2, 14))]
    t17 = [a + b for a, b in zip(hc, row_newbcast(c2, 15))]
    assert max(t17) < (1 << 24)
    # fold: column k >= 9 goes to column k - 9 times R0 and column k - 8 times R1
    hi9 = row_shl(t, 9)
    hi8 = row_shl(t, 8)
    r = [(t[l] & K_LE8[l]) + hi9[l] * K_R0[l] + hi8[l] * K_R1[l] + t16[l] * K_16[l] + t17[l] * K_17[l] for l in range(LANES)]
    assert max(r) < U64
    r9 = row_newbcast([x & (U32 - 1) for x in r], 9)    # column 9 = R1 * t17 (32 bits), folded once more
    assert all(r[l] < U32 for l in range(LANES) if (l & 15) == 9)
    r = [r[l] + r9[l] * K_FOLD2[l] for l in range(LANES)]
    assert max(r) < U64
    return fer_carry(r)


def fer_carry(r):
    """carry pass of 64-bit lane values (lanes 0..8 meaningful) to lazy limbs: limbs 0..7 keep 29 bits, limb 8 keeps 24, the
    carries move one lane up, what leaves limb 8 comes back through 2^256 = 0x3D1 + 8 * 2^29"""
    lo = [r[l] & K_WMASK[l] for l in range(LANES)]
    cc = [r[l] >> K_SHIFT[l] for l in range(LANES)]         # v_lshrrev_b64, low word taken
    assert all(cc[l] < U32 for l in range(LANES) if (l & 15) <= 8)
    cc = [x & (U32 - 1) for x in cc]
    x = row_newbcast(cc, 8)
    assert max(x) < (1 << 22), "top carry too large for the 32-bit fold"
    up = row_shr([cc[l] & K_LT8[l] for l in range(LANES)], 1)
    s = [lo[l] + up[l] + x[l] * K_TOP[l] for l in range(LANES)]
    assert max(s) < U32
    return [s[l] & K_LE8[l] for l in range(LANES)]


def fer_mulsum(pairs, addend=None):
    """sum of the products a * b (+ a lazy addend), one reduction; every row multiplies its own operands"""
    acc = list(addend) if addend else [0] * LANES
    h = [0] * LANES
    for a, b in pairs:
        for i in range(9):
            ai = row_newbcast(a, i)
            bs = row_shr(b, i) if i else b
            acc = [x + y * z for x, y, z in zip(acc, ai, bs)]
            assert max(acc) < U64, "column sum overflows 64 bits"
        a8, b8 = row_newbcast(a, 8), row_newbcast(b, 8)
        h = [x + y * z for x, y, z in zip(h, a8, b8)]
    return _tail(acc, h)


def fer_mul(a, b):
    return fer_mulsum([(a, b)])


def fer_small_norm(d, k):
    """d * k for per-lane small k (< 2^7), carried to lazy limbs"""
    r = [x * y for x, y in zip(d, k)]
    assert max(r) < (1 << 40)
    return fer_carry(r)


def fer_norm(d):
    return fer_carry(list(d))


ROW = [l >> 4 for l in range(LANES)]


def sel(rows):
    """rows: four registers; row r of the result is row r of rows[r]"""
    return by_row(*rows)


K_DBL1 = lane_const(lambda j, r: (63, 21, 4, 63)[r])
K_DBL2 = lane_const(lambda j, r: (0, 4, 0, 21)[r])
SH_A = lane_const(lambda j, r: (0, 0, 1, 0)[r])
SH_B = lane_const(lambda j, r: (1, 1, 0, 0)[r])
K_ADD_T = lane_const(lambda j, r: (3, 1, 21, 1)[r])
K_ADD_U = lane_const(lambda j, r: (1, 1, 21, 1)[r])


def ptr_double(X, Y, Z):
    """2 (X : Y : Z); coordinates replicated in all four rows.  In: X, Z one unit, Y up to two; out the same."""
    # round 1: X Y | Y^2 | Y Z | Z^2
    Pr = fer_mul(sel((X, Y, Y, Z)), sel((Y, Y, Z, Z)))
    e, o = permlane16_swap(Pr, Pr)                 # e = (XY, XY, YZ, YZ), o = (YY, YY, ZZ, ZZ)
    YY, ZZ = permlane32_swap(o, o)
    op1 = fer_small_norm(sel((ZZ, ZZ, YY, ZZ)), K_DBL1)     # 63 ZZ | 21 ZZ | 4 YY | 63 ZZ
    op2 = fer_small_norm(sel((ZZ, YY, ZZ, ZZ)), K_DBL2)     # -     | 4 YY  | -    | 21 ZZ
    t0m = fer_add(YY, fer_negate(op1, 1))                    # rows 0, 3: YY - 63 ZZ
    y3 = fer_add(YY, op2)                                    # row 3: YY + 21 ZZ
    # round 2: X3 = (XY)(2 t0m) | (21 ZZ)(8 YY) | Z3 = (8 YY)(YZ) | t0m y3 ;  Y3 = row 1 + row 3
    A2 = fer_shl(sel((e, op1, op1, t0m)), SH_A)
    B2 = fer_shl(sel((t0m, op2, e, y3)), SH_B)
    R = fer_mul(A2, B2)
    e2, o2 = permlane16_swap(R, R)                 # (R0, R0, R2, R2), (R1, R1, R3, R3)
    X3, Z3 = permlane32_swap(e2, e2)
    ya, yb = permlane32_swap(o2, o2)
    return X3, fer_add(ya, yb), Z3


def ptr_add(P1, P2):
    """P1 + P2, complete (Algorithm 7); coordinates replicated in all rows.  In: X, Z one unit, Y up to two; out: all one."""
    X1, Y1, Z1 = P1
    X2, Y2, Z2 = P2
    Y1, Y2 = fer_norm(Y1), fer_norm(Y2)
    T = fer_mul(sel((X1, Y1, Z1, Z1)), sel((X2, Y2, Z2, Z2)))            # t0 | t1 | t2 | -
    t0, t1, t2, _ = bcast_rows(T)
    e = fer_negate(fer_add(sel((t0, t1, t0, t0)), sel((t1, t2, t2, t2))), 2)
    A = fer_add(sel((X1, Y1, X1, X1)), sel((Y1, Z1, Z1, Z1)))
    B = fer_add(sel((X2, Y2, X2, X2)), sel((Y2, Z2, Z2, Z2)))
    U = fer_mulsum([(A, B)], addend=e)                                    # t3 | t4 | y3' | -
    Ts = fer_small_norm(T, K_ADD_T)                                       # 3 t0 | t1 | 21 t2
    Us = fer_small_norm(U, K_ADD_U)                                       # t3 | t4 | 21 y3'
    t0p, t1n, t2p, _ = bcast_rows(Ts)
    t3, t4, y3, _ = bcast_rows(Us)
    V = fer_add(t1n, fer_negate(t2p, 1))                                  # t1 - t2'  [3]
    Wp = fer_add(t1n, t2p)                                                # t1 + t2'  [2]
    # X3 = t3 V - t4 y3 | Z3 = t4 W + t3 t0' | Y3 = W V + y3 t0'
    R = fer_mulsum([(sel((t3, t4, Wp, Wp)), sel((V, Wp, V, V))), (sel((fer_negate(t4, 1), t3, y3, y3)), sel((y3, t0p, t0p, t0p)))])
    X3, Z3, Y3, _ = bcast_rows(R)
    return X3, Y3, Z3
