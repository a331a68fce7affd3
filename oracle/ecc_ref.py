"""CPU ORACLE (test infrastructure, not product code) -- pure-Python big-int
restatement of eccoxide's scalar-multiplication hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (eccoxide_amd/) never does.

Two independent layers live here on purpose:

  1. `affine_*`  -- textbook chord/tangent arithmetic on affine big ints.  This is
     the mathematical ground truth (what every correct implementation must
     produce as affine bytes).
  2. `ref_*`     -- a step-by-step restatement of the reference's algorithm:
     the Renes-Costello-Batina complete projective formulas, the constant-time
     fixed 4-bit window ladder, the comb fixed-base loop and the extended
     Edwards addition, each citing the reference file:line it follows.  It
     reproduces the reference's un-normalised (X:Y:Z) residues, not only the
     affine result.

Parity pinning: tests/test_oracle_golden.py checks both layers against every
golden vector the reference's own tests hold for this path (tests/golden/*.json,
extracted by tools/extract_fixtures.py): NIST kG (src/tests/kats_data.rs),
RFC 6979 / RFC 8032 keys, BLS12-381 G1 k*G encodings and the comb tables.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Optional, Tuple

# --------------------------------------------------------------------------
# Curve parameters.  Standard published constants (SEC 2 / FIPS 186-4, the
# BLS12-381 spec, RFC 8032); cross-checked against the reference's
# src/params/*.rs by tests/test_oracle_golden.py::test_params_match_reference.
# --------------------------------------------------------------------------


@dataclass(frozen=True)
class WeierstrassParams:
    name: str
    p: int
    n: int          # order of the generator
    a: int
    b: int
    gx: int
    gy: int
    fb: int         # field bytes
    sb: int         # scalar bytes
    flavour: str    # "am3" | "a0"

    @property
    def nw(self) -> int:
        return 2 * self.sb

    @property
    def b3(self) -> int:
        return (3 * self.b) % self.p


P256R1 = WeierstrassParams(
    "p256r1",
    p=0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF,
    n=0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551,
    a=-3 % 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF,
    b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
    gx=0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
    gy=0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5,
    fb=32, sb=32, flavour="am3",
)

_P384 = 2**384 - 2**128 - 2**96 + 2**32 - 1
P384R1 = WeierstrassParams(
    "p384r1",
    p=_P384,
    n=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973,
    a=-3 % _P384,
    b=0xB3312FA7E23EE7E4988E056BE3F82D19181D9C6EFE8141120314088F5013875AC656398D8A2ED19D2A85C8EDD3EC2AEF,
    gx=0xAA87CA22BE8B05378EB1C71EF320AD746E1D3B628BA79B9859F741E082542A385502F25DBF55296C3A545E3872760AB7,
    gy=0x3617DE4A96262C6F5D9E98BF9292DC29F8F41DBD289A147CE9DA3113B5F0B8C00A60B1CE1D7E819D7A431D7C90EA0E5F,
    fb=48, sb=48, flavour="am3",
)

_P521 = 2**521 - 1
P521R1 = WeierstrassParams(
    "p521r1",
    p=_P521,
    n=2**521 - 0x5AE79787C40D069948033FEB708F65A2FC44A36477663B851449048E16EC79BF7,
    a=-3 % _P521,
    b=0x0051953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00,
    gx=0x00C6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
    gy=0x011839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650,
    fb=66, sb=66, flavour="am3",
)

BLS12_381_G1 = WeierstrassParams(
    "bls12_381_g1",
    p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    n=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    a=0,
    b=4,
    gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    fb=48, sb=32, flavour="a0",
)


@dataclass(frozen=True)
class EdwardsParams:
    name: str
    p: int
    n: int
    d: int
    gx: int
    gy: int
    fb: int = 32
    sb: int = 32

    @property
    def nw(self) -> int:
        return 2 * self.sb

    @property
    def d2(self) -> int:
        return (2 * self.d) % self.p


_P25519 = 2**255 - 19
ED25519 = EdwardsParams(
    "ed25519",
    p=_P25519,
    n=2**252 + 27742317777372353535851937790883648493,
    d=(-121665 * pow(121666, -1, _P25519)) % _P25519,
    gx=0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
    gy=0x6666666666666666666666666666666666666666666666666666666666666658,
)

WEIERSTRASS = {c.name: c for c in (P256R1, P384R1, P521R1, BLS12_381_G1)}
CURVES = dict(WEIERSTRASS)
CURVES[ED25519.name] = ED25519

# --------------------------------------------------------------------------
# Layer 1: textbook affine arithmetic (None = point at infinity)
# --------------------------------------------------------------------------
Affine = Optional[Tuple[int, int]]


def affine_add(c: WeierstrassParams, P: Affine, Q: Affine) -> Affine:
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    p = c.p
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 + c.a) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def affine_mul(c: WeierstrassParams, k: int, P: Affine) -> Affine:
    R: Affine = None
    Q = P
    while k:
        if k & 1:
            R = affine_add(c, R, Q)
        Q = affine_add(c, Q, Q)
        k >>= 1
    return R


def on_curve(c: WeierstrassParams, P: Affine) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - (x * x * x + c.a * x + c.b)) % c.p == 0


def ed_affine_add(c: EdwardsParams, P: Tuple[int, int], Q: Tuple[int, int]) -> Tuple[int, int]:
    """Unified twisted-Edwards addition, a = -1 (textbook form)."""
    x1, y1 = P
    x2, y2 = Q
    p = c.p
    t = c.d * x1 * x2 * y1 * y2 % p
    x3 = (x1 * y2 + x2 * y1) * pow(1 + t, -1, p) % p
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, p) % p
    return (x3, y3)


def ed_affine_mul(c: EdwardsParams, k: int, P: Tuple[int, int]) -> Tuple[int, int]:
    R = (0, 1)
    Q = P
    while k:
        if k & 1:
            R = ed_affine_add(c, R, Q)
        Q = ed_affine_add(c, Q, Q)
        k >>= 1
    return R


# --------------------------------------------------------------------------
# Layer 2: restatement of the reference algorithm
# --------------------------------------------------------------------------
Proj = Tuple[int, int, int]
INFINITY: Proj = (0, 1, 0)  # src/curve/projective.rs:152-156


def ref_add_am3(c: WeierstrassParams, P: Proj, Q: Proj) -> Proj:
    """RCB Algorithm 4 exactly as sequenced in src/curve/projective.rs:378-422."""
    p, b = c.p, c.b
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    t0 = X1 * X2 % p; t1 = Y1 * Y2 % p; t2 = Z1 * Z2 % p
    t3 = (X1 + Y1) % p; t4 = (X2 + Y2) % p; t3 = t3 * t4 % p
    t4 = (t0 + t1) % p; t3 = (t3 - t4) % p; t4 = (Y1 + Z1) % p
    X3 = (Y2 + Z2) % p; t4 = t4 * X3 % p; X3 = (t1 + t2) % p
    t4 = (t4 - X3) % p; X3 = (X1 + Z1) % p; Y3 = (X2 + Z2) % p
    X3 = X3 * Y3 % p; Y3 = (t0 + t2) % p; Y3 = (X3 - Y3) % p
    Z3 = b * t2 % p; X3 = (Y3 - Z3) % p; Z3 = (X3 + X3) % p
    X3 = (X3 + Z3) % p; Z3 = (t1 - X3) % p; X3 = (t1 + X3) % p
    Y3 = b * Y3 % p; t1 = (t2 + t2) % p; t2 = (t1 + t2) % p
    Y3 = (Y3 - t2) % p; Y3 = (Y3 - t0) % p; t1 = (Y3 + Y3) % p
    Y3 = (t1 + Y3) % p; t1 = (t0 + t0) % p; t0 = (t1 + t0) % p
    t0 = (t0 - t2) % p; t1 = t4 * Y3 % p; t2 = t0 * Y3 % p
    Y3 = X3 * Z3 % p; Y3 = (Y3 + t2) % p; X3 = t3 * X3 % p
    X3 = (X3 - t1) % p; Z3 = t4 * Z3 % p; t1 = t3 * t0 % p
    Z3 = (Z3 + t1) % p
    return (X3, Y3, Z3)


def ref_double_am3(c: WeierstrassParams, P: Proj) -> Proj:
    """RCB Algorithm 6 as sequenced in src/curve/projective.rs:606-639."""
    p, b = c.p, c.b
    X, Y, Z = P
    t0 = X * X % p; t1 = Y * Y % p; t2 = Z * Z % p
    t3 = X * Y % p; t3 = (t3 + t3) % p; Z3 = X * Z % p
    Z3 = (Z3 + Z3) % p; Y3 = b * t2 % p; Y3 = (Y3 - Z3) % p
    X3 = (Y3 + Y3) % p; Y3 = (X3 + Y3) % p; X3 = (t1 - Y3) % p
    Y3 = (t1 + Y3) % p; Y3 = X3 * Y3 % p; X3 = X3 * t3 % p
    t3 = (t2 + t2) % p; t2 = (t2 + t3) % p; Z3 = b * Z3 % p
    Z3 = (Z3 - t2) % p; Z3 = (Z3 - t0) % p; t3 = (Z3 + Z3) % p
    Z3 = (Z3 + t3) % p; t3 = (t0 + t0) % p; t0 = (t3 + t0) % p
    t0 = (t0 - t2) % p; t0 = t0 * Z3 % p; Y3 = (Y3 + t0) % p
    t0 = Y * Z % p; t0 = (t0 + t0) % p; Z3 = t0 * Z3 % p
    X3 = (X3 - Z3) % p; Z3 = t0 * t1 % p; Z3 = (Z3 + Z3) % p
    Z3 = (Z3 + Z3) % p
    return (X3, Y3, Z3)


def ref_add_a0(c: WeierstrassParams, P: Proj, Q: Proj) -> Proj:
    """RCB complete addition for a = 0, src/curve/projective.rs:300-337."""
    p, b3 = c.p, c.b3
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    t0 = X1 * X2 % p; t1 = Y1 * Y2 % p; t2 = Z1 * Z2 % p
    t3 = (X1 + Y1) % p; t4 = (X2 + Y2) % p; t3 = t3 * t4 % p
    t4 = (t0 + t1) % p; t3 = (t3 - t4) % p; t4 = (Y1 + Z1) % p
    X3 = (Y2 + Z2) % p; t4 = t4 * X3 % p; X3 = (t1 + t2) % p
    t4 = (t4 - X3) % p; X3 = (X1 + Z1) % p; Y3 = (X2 + Z2) % p
    X3 = X3 * Y3 % p; Y3 = (t0 + t2) % p; Y3 = (X3 - Y3) % p
    X3 = (t0 + t0) % p; t0 = (X3 + t0) % p; t2 = b3 * t2 % p
    Z3 = (t1 + t2) % p; t1 = (t1 - t2) % p; Y3 = b3 * Y3 % p
    X3 = t4 * Y3 % p; t2 = t3 * t1 % p; X3 = (t2 - X3) % p
    Y3 = Y3 * t0 % p; t1 = t1 * Z3 % p; Y3 = (t1 + Y3) % p
    t0 = t0 * t3 % p; Z3 = Z3 * t4 % p; Z3 = (Z3 + t0) % p
    return (X3, Y3, Z3)


def ref_double_a0(c: WeierstrassParams, P: Proj) -> Proj:
    """RCB doubling for a = 0, src/curve/projective.rs:558-582."""
    p, b3 = c.p, c.b3
    X, Y, Z = P
    t0 = Y * Y % p; Z3 = (t0 + t0) % p; Z3 = (Z3 + Z3) % p
    Z3 = (Z3 + Z3) % p; t1 = Y * Z % p; t2 = Z * Z % p
    t2 = b3 * t2 % p; X3 = t2 * Z3 % p; Y3 = (t0 + t2) % p
    Z3 = t1 * Z3 % p; t1 = (t2 + t2) % p; t2 = (t1 + t2) % p
    t0 = (t0 - t2) % p; Y3 = t0 * Y3 % p; Y3 = (X3 + Y3) % p
    t1 = X * Y % p; X3 = t0 * t1 % p; X3 = (X3 + X3) % p
    return (X3, Y3, Z3)


def ref_add(c: WeierstrassParams, P: Proj, Q: Proj) -> Proj:
    return ref_add_am3(c, P, Q) if c.flavour == "am3" else ref_add_a0(c, P, Q)


# ---- BLS12-381 G1 subgroup membership, the reference's endomorphism test ----------------------
# src/curve/bls12_381/g1.rs:62-109; BETA: src/params/bls12_381.rs:100 (tests/golden/params.json "beta")
BLS_X_ABS = 0xD201000000010000   # |x| of the (negative) seed
BLS_BETA = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE


def ref_g1_mul_by_abs_x(P: Proj) -> Proj:
    """g1.rs:69-78: [|x|]P by double-and-add over the bits of |x| below the top one, complete formulas."""
    c = BLS12_381_G1
    acc = P
    for i in range(62, -1, -1):
        acc = ref_double_a0(c, acc)
        if (BLS_X_ABS >> i) & 1:
            acc = ref_add_a0(c, acc, P)
    return acc


def ref_g1_is_in_subgroup(P: Affine) -> bool:
    """g1.rs:105-109: sigma(P) == -[x^2]P with sigma(x, y) = (beta x, y); projective equality as
    is_equivalent (projective.rs:133-139).  The identity is in the subgroup."""
    if P is None:
        return True
    c = BLS12_381_G1
    p = c.p
    Q = ref_g1_mul_by_abs_x(ref_g1_mul_by_abs_x((P[0], P[1], 1)))
    mx, my, mz = Q[0], (-Q[1]) % p, Q[2]
    sx, sy, sz = BLS_BETA * P[0] % p, P[1], 1
    return (sx * mz - mx * sz) % p == 0 and (sy * mz - my * sz) % p == 0 and (mz % p != 0)


def ref_double(c: WeierstrassParams, P: Proj) -> Proj:
    return ref_double_am3(c, P) if c.flavour == "am3" else ref_double_a0(c, P)


def ref_scalar_mul_fixed_window(c: WeierstrassParams, P: Proj, n_be: bytes) -> Proj:
    """src/curve/projective.rs:871-896 (am3) / :842-867 (a0): table[d]=d*P,
    then per big-endian byte, hi nibble then lo nibble: 4 doublings, lookup, add."""
    table: List[Proj] = [INFINITY] * 16
    table[1] = P
    table[2] = ref_double(c, P)
    for d in range(3, 16):
        table[d] = ref_add(c, table[d - 1], P)
    q = INFINITY
    for byte in n_be:
        for idx in (byte >> 4, byte & 0x0F):
            q = ref_double(c, ref_double(c, ref_double(c, ref_double(c, q))))
            q = ref_add(c, q, table[idx])   # select_from_table, projective.rs:427-434
    return q


def ref_comb_table(c: WeierstrassParams) -> List[List[Affine]]:
    """COMB_TABLE[i][j] = (j+1) * 16^i * G as affine points
    (src/params/comb/p256r1.rs:3-8; generator sage/comb.sage)."""
    out = []
    base: Affine = (c.gx, c.gy)
    for _ in range(c.nw):
        row = []
        acc: Affine = None
        for _j in range(15):
            acc = affine_add(c, acc, base)
            row.append(acc)
        out.append(row)
        base = affine_add(c, row[14], base)  # 16 * base
    return out


def ref_mul_base_table(c: WeierstrassParams, table: List[List[Affine]], n_be: bytes) -> Proj:
    """src/curve/projective.rs:965-981 (am3) / :945-961 (a0); runtime table from
    build_comb_table :451-472 (index 0 = infinity, z = 1 otherwise)."""
    assert len(table) == 2 * len(n_be), "comb table size mismatch"
    q = INFINITY
    for i in range(len(table)):
        byte = n_be[len(n_be) - 1 - i // 2]
        digit = (byte & 0x0F) if i % 2 == 0 else (byte >> 4)
        if digit == 0:
            sel = INFINITY
        else:
            x, y = table[i][digit - 1]
            sel = (x, y, 1)
        q = ref_add(c, q, sel)
    return q


def ref_to_affine(c: WeierstrassParams, P: Proj) -> Affine:
    """src/curve/projective.rs:655-682: z==0 -> not present; else (X/Z, Y/Z)."""
    X, Y, Z = P
    if Z % c.p == 0:
        return None
    zi = pow(Z, -1, c.p)
    return (X * zi % c.p, Y * zi % c.p)


# ---- Edwards (src/curve/curve25519.rs) ------------------------------------
Ext = Tuple[int, int, int, int]
ED_IDENTITY: Ext = (0, 1, 1, 0)  # curve25519.rs:623-628


def ref_ed_from_affine(c: EdwardsParams, x: int, y: int) -> Ext:
    return (x, y, 1, x * y % c.p)  # curve25519.rs:638-645


def ref_ed_add(c: EdwardsParams, P: Ext, Q: Ext) -> Ext:
    """curve25519.rs:695-710."""
    p = c.p
    X1, Y1, Z1, T1 = P
    X2, Y2, Z2, T2 = Q
    aa = (Y1 - X1) * (Y2 - X2) % p
    bb = (Y1 + X1) * (Y2 + X2) % p
    cc = (c.d2 * T1 % p) * T2 % p
    dd = 2 * (Z1 * Z2 % p) % p
    e = (bb - aa) % p; f = (dd - cc) % p; g = (dd + cc) % p; h = (bb + aa) % p
    return (e * f % p, g * h % p, f * g % p, e * h % p)


def ref_ed_double(c: EdwardsParams, P: Ext) -> Ext:
    """curve25519.rs:604-619 + :669-677."""
    p = c.p
    X, Y, Z, _ = P
    a = X * X % p; b = Y * Y % p; cc = 2 * (Z * Z % p) % p
    d = -a % p
    e = ((X + Y) ** 2 - (a + b)) % p
    g = (d + b) % p; f = (g - cc) % p; h = (d - b) % p
    return (e * f % p, g * h % p, f * g % p, e * h % p)


def ref_ed_comb_table(c: EdwardsParams) -> List[List[Tuple[int, int]]]:
    out = []
    base = (c.gx, c.gy)
    for _ in range(c.nw):
        row = []
        acc = (0, 1)
        for _j in range(15):
            acc = ed_affine_add(c, acc, base)
            row.append(acc)
        out.append(row)
        base = ed_affine_add(c, row[14], base)
    return out


def ref_ed_mul_base(c: EdwardsParams, table, n_be: bytes) -> Ext:
    """curve25519.rs:840-851 (the loop indexes the BIG-endian scalar bytes)."""
    q = ED_IDENTITY
    for i in range(len(table)):
        byte = n_be[len(n_be) - 1 - i // 2]
        digit = (byte & 0x0F) if i % 2 == 0 else (byte >> 4)
        sel = ED_IDENTITY if digit == 0 else ref_ed_from_affine(c, *table[i][digit - 1])
        q = ref_ed_add(c, q, sel)
    return q


def ref_ed_scale_bytes(c: EdwardsParams, P: Ext, k_be: bytes) -> Ext:
    """curve25519.rs:746-757: double-and-(selected)-add, MSB first."""
    q = ED_IDENTITY
    for byte in k_be:
        for i in range(7, -1, -1):
            q = ref_ed_double(c, q)
            if (byte >> i) & 1:
                q = ref_ed_add(c, q, P)
    return q


def ref_ed_to_affine(c: EdwardsParams, P: Ext) -> Tuple[int, int]:
    zi = pow(P[2], -1, c.p)  # curve25519.rs:663-666 (Z is never 0 on Edwards)
    return (P[0] * zi % c.p, P[1] * zi % c.p)


def ed_encode_point(c: EdwardsParams, aff: Tuple[int, int]) -> bytes:
    """RFC 8032 5.1.2 encoding (what protocol/ed25519.rs encode_point emits)."""
    x, y = aff
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def ed25519_secret_scalar(seed: bytes) -> int:
    """protocol/ed25519.rs:61-79: SHA-512, clamp, reduce mod l."""
    h = bytearray(hashlib.sha512(seed).digest()[:32])
    h[0] &= 248
    h[31] &= 127
    h[31] |= 64
    return int.from_bytes(bytes(h), "little") % ED25519.n


# ---- Montgomery x-only ladder (src/curve/curve25519.rs:472-513) ----------------
X25519_A24 = 121666  # MONT_A24_BYTES, curve25519.rs:374-377


def ref_ladder(base_u: int, k_be: bytes) -> int:
    """curve25519.rs:474-513: returns the u-coordinate of k*P (0 for the point at infinity)."""
    p = _P25519
    x1 = base_u % p
    x2, z2, x3, z3 = 1, 0, x1, 1
    swap = 0
    for byte in k_be:
        for i in range(7, -1, -1):
            bit = (byte >> i) & 1
            swap ^= bit
            if swap:
                x2, x3 = x3, x2
                z2, z3 = z3, z2
            swap = bit
            a = (x2 + z2) % p; aa = a * a % p
            b = (x2 - z2) % p; bb = b * b % p
            e = (aa - bb) % p
            c = (x3 + z3) % p; d = (x3 - z3) % p
            da = d * a % p; cb = c * b % p
            x3 = (da + cb) ** 2 % p
            z3 = x1 * ((da - cb) ** 2 % p) % p
            x2 = aa * bb % p
            z2 = e * ((bb + X25519_A24 * e) % p) % p
    if swap:
        x2, x3 = x3, x2
        z2, z3 = z3, z2
    return x2 * pow(z2, p - 2, p) % p  # invert_or_zero: 0 -> 0


def ref_x25519(scalar_le: bytes, u_le: bytes) -> bytes:
    """protocol/x25519.rs:14-45: clamp, reverse to big-endian, mask the top bit of u."""
    k = bytearray(scalar_le)
    k[0] &= 248
    k[31] &= 127
    k[31] |= 64
    u = bytearray(u_le)
    u[31] &= 0x7F
    r = ref_ladder(int.from_bytes(bytes(u), "little"), bytes(reversed(k)))
    return r.to_bytes(32, "little")


def montgomery_u_of_edwards(y: int) -> int:
    """u = (1 + y)/(1 - y) (curve25519.rs:790-798)"""
    p = _P25519
    return (1 + y) * pow(1 - y, -1, p) % p


# --------------------------------------------------------------------------
# Byte-level batch API (same conventions as the C ABI, include/eccx.h)
# --------------------------------------------------------------------------
def scalarmul_var_bytes(curve: str, scalars: bytes, points: bytes) -> Tuple[bytes, bytes]:
    """n x SB big-endian scalars, n x 2FB affine x||y big-endian -> (n x 2FB, n flags)."""
    c = WEIERSTRASS[curve]
    n = len(scalars) // c.sb
    out = bytearray()
    inf = bytearray()
    for i in range(n):
        k = scalars[i * c.sb:(i + 1) * c.sb]
        x = int.from_bytes(points[i * 2 * c.fb: i * 2 * c.fb + c.fb], "big")
        y = int.from_bytes(points[i * 2 * c.fb + c.fb:(i + 1) * 2 * c.fb], "big")
        q = ref_scalar_mul_fixed_window(c, (x, y, 1), k)
        a = ref_to_affine(c, q)
        if a is None:
            out += bytes(2 * c.fb)
            inf.append(1)
        else:
            out += a[0].to_bytes(c.fb, "big") + a[1].to_bytes(c.fb, "big")
            inf.append(0)
    return bytes(out), bytes(inf)


_COMB_CACHE = {}


def comb_table(curve: str):
    if curve not in _COMB_CACHE:
        c = CURVES[curve]
        _COMB_CACHE[curve] = ref_ed_comb_table(c) if isinstance(c, EdwardsParams) else ref_comb_table(c)
    return _COMB_CACHE[curve]


def scalarmul_base_bytes(curve: str, scalars: bytes) -> Tuple[bytes, bytes]:
    c = CURVES[curve]
    n = len(scalars) // c.sb
    tab = comb_table(curve)
    out = bytearray()
    inf = bytearray()
    for i in range(n):
        k = scalars[i * c.sb:(i + 1) * c.sb]
        if isinstance(c, EdwardsParams):
            x, y = ref_ed_to_affine(c, ref_ed_mul_base(c, tab, k))
            # Edwards affine bytes are little-endian in the reference (curve25519.rs:138)
            out += x.to_bytes(32, "little") + y.to_bytes(32, "little")
            inf.append(1 if (x, y) == (0, 1) else 0)
        else:
            a = ref_to_affine(c, ref_mul_base_table(c, tab, k))
            if a is None:
                out += bytes(2 * c.fb)
                inf.append(1)
            else:
                out += a[0].to_bytes(c.fb, "big") + a[1].to_bytes(c.fb, "big")
                inf.append(0)
    return bytes(out), bytes(inf)


# ---- point wire formats ----------------------------------------
# Statuses of the batched decoders: 0 point, 1 the encoding of the point at infinity, 2 rejected.
CODEC_OK, CODEC_INFINITY, CODEC_INVALID = 0, 1, 2


def ref_sqrt_p3mod4(p: int, a: int) -> Optional[int]:
    """sqrt for p = 3 (mod 4): the candidate a^((p+1)/4), present only if it squares back to a
    (sec2/p256r1.rs:68-84, p384r1.rs:71-, p521r1.rs:126-131, bls12_381/fp.rs:64-68)."""
    r = pow(a, (p + 1) // 4, p)
    return r if r * r % p == a % p else None


def ref_w_decompress_xy(c: WeierstrassParams, x: int, negative: bool) -> Optional[Tuple[int, int]]:
    """affine::Point::decompress (src/curve/affine.rs:48-58): y = sqrt(x^3 + a x + b), replaced by
    -y when its sign (low bit of the canonical value, fiat/field_macros.rs:557-565) is not the
    requested one."""
    yy = (x * x * x + c.a * x + c.b) % c.p
    y = ref_sqrt_p3mod4(c.p, yy)
    if y is None:
        return None
    if bool(y & 1) != negative:
        y = (-y) % c.p
    return (x, y)


def _bls_is_largest(c: WeierstrassParams, y: int) -> bool:
    return y > (c.p - 1) // 2  # Fp::is_largest, bls12_381/serialize.rs:143-148


def compressed_bytes(curve: str) -> int:
    if curve == "ed25519":
        return 32
    c = WEIERSTRASS[curve]
    return c.fb if curve == "bls12_381_g1" else c.fb + 1


def ref_point_compress(curve: str, P: Affine) -> bytes:
    """One affine point (None = infinity) -> its compressed encoding.
    sec2: SEC1 0x02 | (y odd) || x -- the byte form of PointAffine::compress's (x, Sign)
    (curve_macros.rs:211-213); infinity = FB + 1 zero bytes.
    bls12_381_g1: to_compressed (serialize.rs:253-262, flags :52-62,103-106); infinity
    = 0xC0 then zeros (:93-95).  ed25519: encode_point (protocol/ed25519.rs:27-36)."""
    if curve == "ed25519":
        return ed_encode_point(ED25519, P)
    c = WEIERSTRASS[curve]
    if curve == "bls12_381_g1":
        if P is None:
            return bytes([0xC0]) + bytes(c.fb - 1)
        out = bytearray(P[0].to_bytes(c.fb, "big"))
        out[0] |= 0x80 | (0x20 if _bls_is_largest(c, P[1]) else 0)
        return bytes(out)
    if P is None:
        return bytes(c.fb + 1)
    return bytes([2 | (P[1] & 1)]) + P[0].to_bytes(c.fb, "big")


def ref_ed_sqrt_div(p: int, u: int, v: int) -> Optional[int]:
    """FieldElement::sqrt_div (curve25519.rs:246-265): r = u v^3 (u v^7)^((p-5)/8); v r^2 = u:
    r; v r^2 = -u: i r; else none."""
    v3 = v * v % p * v % p
    v7 = v3 * v3 % p * v % p
    r = u * v3 % p * pow(u * v7 % p, (p - 5) // 8, p) % p
    check = v * r % p * r % p
    if check == u % p:
        return r
    if check == (-u) % p:
        return r * pow(2, (p - 1) // 4, p) % p
    return None


def ref_point_decompress(curve: str, enc: bytes, check_subgroup: bool = False) -> Tuple[Affine, int]:
    """One encoding -> (point or None, status)."""
    if curve == "ed25519":
        c = ED25519
        sign = enc[31] >> 7  # decode_point, protocol/ed25519.rs:38-59
        y = int.from_bytes(enc, "little") & ((1 << 255) - 1)
        if y >= c.p:
            return None, CODEC_INVALID
        if sign and y in (1, c.p - 1):
            return None, CODEC_INVALID
        yy = y * y % c.p  # Point::decompress, curve25519.rs:772-784
        x = ref_ed_sqrt_div(c.p, (yy - 1) % c.p, (c.d * yy + 1) % c.p)
        if x is None:
            return None, CODEC_INVALID
        if (x & 1) != sign:
            x = (-x) % c.p
        return (x, y), CODEC_OK
    c = WEIERSTRASS[curve]
    if curve == "bls12_381_g1":
        fl = enc[0] & 0xE0  # read_compressed_flags, serialize.rs:109-127
        x = int.from_bytes(enc, "big") & ((1 << (8 * c.fb - 3)) - 1)
        if not fl & 0x80:
            return None, CODEC_INVALID
        if fl & 0x40:
            return None, (CODEC_INFINITY if not (fl & 0x20) and x == 0 else CODEC_INVALID)
        if x >= c.p:
            return None, CODEC_INVALID
        P = ref_w_decompress_xy(c, x, False)  # read_compressed_affine, serialize.rs:181-202
        if P is None:
            return None, CODEC_INVALID
        if _bls_is_largest(c, P[1]) != bool(fl & 0x20):
            P = (P[0], (-P[1]) % c.p)
        if check_subgroup and not ref_g1_is_in_subgroup(P):  # from_compressed, :299-313 -> g1.rs:105-109
            return None, CODEC_INVALID
        return P, CODEC_OK
    x = int.from_bytes(enc[1:], "big")
    if enc[0] not in (2, 3):
        return None, (CODEC_INFINITY if enc[0] == 0 and x == 0 else CODEC_INVALID)
    if x >= c.p:
        return None, CODEC_INVALID
    P = ref_w_decompress_xy(c, x, enc[0] == 3)
    return (P, CODEC_OK) if P is not None else (None, CODEC_INVALID)


def point_compress_bytes(curve: str, xy: bytes, inf: Optional[bytes] = None) -> bytes:
    """Batch form over the engine's x||y records (little-endian for ed25519)."""
    fb = CURVES[curve].fb
    order = "little" if curve == "ed25519" else "big"
    out = bytearray()
    for i in range(len(xy) // (2 * fb)):
        rec = xy[i * 2 * fb:(i + 1) * 2 * fb]
        P = None if (inf is not None and inf[i]) else (int.from_bytes(rec[:fb], order), int.from_bytes(rec[fb:], order))
        out += ref_point_compress(curve, P)
    return bytes(out)


def point_decompress_bytes(curve: str, enc: bytes, check_subgroup: bool = False) -> Tuple[bytes, bytes]:
    fb = CURVES[curve].fb
    eb = compressed_bytes(curve)
    order = "little" if curve == "ed25519" else "big"
    out, flags = bytearray(), bytearray()
    for i in range(len(enc) // eb):
        P, st = ref_point_decompress(curve, enc[i * eb:(i + 1) * eb], check_subgroup)
        out += bytes(2 * fb) if P is None else P[0].to_bytes(fb, order) + P[1].to_bytes(fb, order)
        flags.append(st)
    return bytes(out), bytes(flags)


def ref_g1_to_uncompressed(P: Affine) -> bytes:
    """to_uncompressed (bls12_381/serialize.rs:269-277): x||y big-endian, flags clear; the identity
    is the infinity flag alone (infinity_uncompressed, :98-100)."""
    c = BLS12_381_G1
    if P is None:
        return bytes([0x40]) + bytes(2 * c.fb - 1)
    return P[0].to_bytes(c.fb, "big") + P[1].to_bytes(c.fb, "big")


def ref_g1_from_uncompressed(enc: bytes, check_subgroup: bool = False) -> Tuple[Affine, int]:
    """from_uncompressed[_oncurve_only] (serialize.rs:353-383; read_uncompressed_flags :129-141,
    read_uncompressed_affine :207-224)."""
    c = BLS12_381_G1
    fl = enc[0] & 0xE0
    x = int.from_bytes(enc[:c.fb], "big") & ((1 << (8 * c.fb - 3)) - 1)
    y = int.from_bytes(enc[c.fb:], "big")
    if fl & 0xA0:
        return None, CODEC_INVALID
    if fl & 0x40:
        return None, (CODEC_INFINITY if x == 0 and y == 0 else CODEC_INVALID)
    if x >= c.p or y >= c.p or not on_curve(c, (x, y)):
        return None, CODEC_INVALID
    if check_subgroup and not ref_g1_is_in_subgroup((x, y)):
        return None, CODEC_INVALID
    return (x, y), CODEC_OK
