"""The algebra of the affine-table ladder (eccoxide_amd/csrc/kernels_coz.hpp), restated in Python integers
and checked against the oracle's affine group law -- on the CPU, no kernel involved.

The kernel's steps, in its order:
  * 2P from the affine P, and P itself over the denominator of 2P (4 x y^2, 8 y^4);
  * 14 co-Z additions with update (Goundar-Joye-Miyaji ZADDU), each handing out the ratio of denominators;
  * the backward pass to one denominator: lambda_d = zeta / Z_d from the ratios, entries (X l^2, Y l^3) --
    started at 1 / zeta on the a = -3 curves (entries affine on the curve), at 1 on a = 0 (entries affine on
    the isomorphic curve y^2 = x^3 + b zeta^6; Z *= zeta brings a result back);
  * signed 5-bit windows over those entries with Jacobian doublings and mixed additions.
What is asserted: every entry is d*P, the isomorphism maps back to the curve, a full ladder equals k*P, and
a base of small order is detected by a zero ratio (the condition the kernel marks units by).
"""
import random

import pytest

from oracle import ecc_ref as R

CURVES = [R.P256R1, R.P384R1, R.P521R1, R.BLS12_381_G1]


def dblu(c, x, y):
    """(2P over Z = 2y, P over the same Z, Z) from the affine P; a = -3 or a = 0."""
    p = c.p
    xx, yy = x * x % p, y * y % p
    yyyy = yy * yy % p
    s = 4 * x * yy % p
    m = (3 * xx + c.a) % p
    x3 = (m * m - 2 * s) % p
    y3 = (m * (s - x3) - 8 * yyyy) % p
    return (x3, y3), (s, 8 * yyyy % p), 2 * y % p


def zaddu(c, x1, y1, x2, y2):
    """(x1, y1) + (x2, y2) over a common Z; returns the sum, the first operand over the new Z, and the ratio."""
    p = c.p
    d = (x1 - x2) % p
    cc = d * d % p
    w1, w2 = x1 * cc % p, x2 * cc % p
    e = (y1 - y2) % p
    a1 = y1 * (w1 - w2) % p
    x3 = (e * e - w1 - w2) % p
    y3 = (e * (w1 - x3) - a1) % p
    return (x3, y3), (w1, a1), d


def build_table(c, P):
    """rows[d] = (X_d, Y_d, ratio_d) for d = 2..16 as the kernel stores them, plus zeta = Z_16; None if degenerate."""
    x, y = P
    rows = {1: (x, y, 1)}
    T, Pz, ratio = dblu(c, x, y)
    if ratio == 0:
        return None
    z = ratio
    rows[2] = (T[0], T[1], ratio)
    for d in range(3, 17):
        T, Pz, ratio = zaddu(c, Pz[0], Pz[1], T[0], T[1])
        if ratio == 0:
            return None
        z = z * ratio % c.p
        rows[d] = (T[0], T[1], ratio)
    return rows, z


def common_denominator(c, rows, zeta, invert):
    """The backward pass: entries over zeta (invert=False: affine on the isomorphic curve) or truly affine."""
    p = c.p
    lam = pow(zeta, -1, p) if invert else 1
    out = {16: (rows[16][0] * lam * lam % p, rows[16][1] * lam * lam * lam % p)}
    nxt = rows[16][2]
    for d in range(15, 0, -1):
        lam = lam * nxt % p
        X, Y, nxt = rows[d]
        out[d] = (X * lam * lam % p, Y * lam * lam * lam % p)
    return out, lam


def jac_dbl(c, P):
    X, Y, Z = P
    p = c.p
    if Z == 0:
        return P
    delta, gamma = Z * Z % p, Y * Y % p
    beta = X * gamma % p
    alpha = (3 * X * X + c.a * delta * delta) % p
    x3 = (alpha * alpha - 8 * beta) % p
    return (x3, (alpha * (4 * beta - x3) - 8 * gamma * gamma) % p, 2 * Y * Z % p)


def jac_madd(c, P, e):
    """Jacobian + affine with the kernel's special cases (infinity, equal, opposite)."""
    X, Y, Z = P
    x2, y2 = e
    p = c.p
    if Z == 0:
        return (x2, y2, 1)
    zz = Z * Z % p
    h = (x2 * zz - X) % p
    r = (y2 * zz * Z - Y) % p
    if h == 0:
        return jac_dbl(c, P) if r == 0 else (1, 1, 0)
    hh = h * h % p
    v = X * hh % p
    x3 = (r * r - hh * h - 2 * v) % p
    return (x3, (r * (v - x3) - Y * hh * h) % p, Z * h % p)


def booth(k, w):
    """Signed 5-bit digit of window w (bit -1 is 0): (|d|, negative)."""
    w6 = ((k << 1) >> (5 * w)) & 0x3F
    s = w6 >> 5
    m = (63 - w6) if s else w6
    return (m >> 1) + (m & 1), bool(s)


def ladder(c, table, k, bits):
    nwin = (bits + 1 + 4) // 5
    q = (1, 1, 0)
    for w in range(nwin - 1, -1, -1):
        if w != nwin - 1:
            for _ in range(5):
                q = jac_dbl(c, q)
        d, neg = booth(k, w)
        if d:
            x, y = table[d]
            q = jac_madd(c, q, (x, (-y) % c.p if neg else y))
    return q


def to_affine(c, P, zscale=1):
    X, Y, Z = P
    Z = Z * zscale % c.p
    if Z == 0:
        return None
    zi = pow(Z, -1, c.p)
    return (X * zi * zi % c.p, Y * zi * zi * zi % c.p)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_table_entries_are_the_multiples(c):
    rng = random.Random(77)
    P = R.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy))
    rows, zeta = build_table(c, P)
    iso = c.a == 0
    table, lam1 = common_denominator(c, rows, zeta, invert=not iso)
    for d in range(1, 17):
        want = R.affine_mul(c, d, P)
        if iso:
            # (x', y') lies on y^2 = x^3 + b zeta^6 and maps back by (x' / zeta^2, y' / zeta^3)
            x, y = table[d]
            assert (y * y - x * x * x - c.b * pow(zeta, 6, c.p)) % c.p == 0
            assert to_affine(c, (x, y, zeta)) == want
        else:
            assert table[d] == want
    if iso:
        assert lam1 == zeta          # the product of all ratios: where the kernel takes zeta from


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_ladder_over_the_table(c):
    rng = random.Random(78)
    bits = c.n.bit_length() if c.name != "bls12_381_g1" else 256
    bits = (bits + 7) // 8 * 8
    P = R.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy))
    rows, zeta = build_table(c, P)
    iso = c.a == 0
    table, _ = common_denominator(c, rows, zeta, invert=not iso)
    ciso = c if not iso else R.WeierstrassParams(**{**c.__dict__, "b": c.b * pow(zeta, 6, c.p) % c.p})
    ks = [0, 1, 2, 15, 16, 17, 31, 32, 33, c.n - 1, c.n, c.n + 1, (1 << bits) - 1] + [rng.randrange(1 << bits) for _ in range(12)]
    for k in ks:
        got = to_affine(c, ladder(ciso, table, k, bits), zeta if iso else 1)
        assert got == R.affine_mul(c, k % c.n, P), hex(k)


def test_small_order_base_is_detected_by_a_zero_ratio():
    """bls12_381_g1 has points of order 3 (the cofactor is divisible by 3): 2P = -P, so the first co-Z addition
    P + 2P meets equal x coordinates and its ratio is zero -- what the kernel marks the unit by."""
    c = R.BLS12_381_G1
    x = -R.BLS_X_ABS
    cof = (x - 1) ** 2 // 3
    xx = 1
    P3 = None
    while P3 is None:
        xx += 1
        P = R.ref_w_decompress_xy(c, xx, False)
        if P is None:
            continue
        T = R.affine_mul(c, c.n * (cof // 3), P)
        if T is not None and R.affine_mul(c, 3, T) is None:
            P3 = T
    assert build_table(c, P3) is None
