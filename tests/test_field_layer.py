"""GPU stress test of the unsaturated field layer (eccoxide_amd/csrc/ufe.hpp).

The scalar-multiplication parity tests only ever feed the field layer the limb values real
ladders produce.  The layer's safety argument is about bounds, though: every operation is
typed with the largest limbs (K) and value (V) it may receive, and the 64-bit product columns
and 32-bit limbs must not overflow at those extremes.  This test drives single operations
through tests/hip/libfieldcheck.so with the worst inputs the types admit (all limbs at
K*2^B - 1, value just under V*p, digits of 0 / p / 2p) plus random ones, and checks every
result against Python integers: congruence modulo p, limb bound and value bound of the
output type.
"""
import ctypes
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "hip", "libfieldcheck.so")

P = {
    0: 2**256 - 2**224 + 2**192 + 2**96 - 1,
    1: 2**384 - 2**128 - 2**96 + 2**32 - 1,
    2: 2**521 - 1,
    3: 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    4: 2**255 - 19,
}
NAMES = {0: "p256r1", 1: "p384r1", 2: "p521r1", 3: "bls12_381_g1", 4: "curve25519"}
OP_MUL_TIGHT, OP_MUL_LAZY, OP_SQR_LAZY, OP_SUB_CHAIN, OP_REDUCE_MAX, OP_CANONICAL, OP_MUL_AUTO, OP_ADD_AUTO, OP_REDUCE_LAZY, OP_INVERT = range(10)
OP_MUL_ADD, OP_MUL_ADD_MAX, OP_MUL_SUB, OP_MUL_SUB_MAX, OP_MUL_SUB_2SQR = range(10, 15)


class FieldCheck:
    def __init__(self):
        import torch  # noqa: F401  (one HIP runtime in the process, as eccoxide_amd._lib does)

        self.lib = ctypes.CDLL(LIB)
        self.lib.fieldcheck_info.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        self.lib.fieldcheck_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_size_t]

    def info(self, curve):
        arr = (ctypes.c_int * 8)()
        assert self.lib.fieldcheck_info(curve, arr) == 0
        return dict(zip(["N", "B", "KMAX", "KKMAX", "KA", "KB", "KS", "L"], list(arr)))

    def run4(self, curve, op, a, b, c, d):
        arrs = [np.ascontiguousarray(x, dtype=np.uint32) for x in (a, b, c, d)]
        out = np.zeros_like(arrs[0])
        self.lib.fieldcheck_run4.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_size_t]
        rc = self.lib.fieldcheck_run4(curve, op, *[x.ctypes.data for x in arrs], out.ctypes.data, arrs[0].shape[0])
        assert rc == 0, f"fieldcheck_run4 failed: hip error {rc}"
        return out

    def run(self, curve, op, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        out = np.zeros_like(a)
        rc = self.lib.fieldcheck_run(curve, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0])
        assert rc == 0, f"fieldcheck_run failed: hip error {rc}"
        return out


@pytest.fixture(scope="module")
def fc():
    if not os.path.exists(LIB):
        pytest.fail("tests/hip/libfieldcheck.so missing: run __graft_entry__.build()")
    return FieldCheck()


def value(limbs, B):
    return sum(int(x) << (B * i) for i, x in enumerate(limbs))


def gen(rng, inf, p, K, V, count):
    """rows of N limbs, each limb < K*2^B, value < V*p: extremes first, then random"""
    N, B = inf["N"], inf["B"]
    wtop = B * (N - 1)
    top_cap = min(K * (1 << B) - 1, (V * p >> wtop) - K - 1)
    assert top_cap > 0
    lim = K * (1 << B) - 1
    rows = []

    def row(lo, top):
        r = [min(x, lim) for x in lo] + [min(top, top_cap)]
        assert value(r, B) < V * p
        return r

    rows.append(row([lim] * (N - 1), top_cap))            # everything at the bound
    rows.append(row([0] * (N - 1), 0))                     # zero
    rows.append(row([lim] * (N - 1), 0))
    rows.append(row([0] * (N - 1), top_cap))
    rows.append(row([(1 << B) - 1] * (N - 1), top_cap))    # tight all-ones
    for m in (1, 2):                                       # digits of p and 2p (zero mod p)
        d = [(m * p >> (B * i)) & ((1 << B) - 1) for i in range(N - 1)] + [m * p >> wtop]
        if d[-1] <= top_cap:
            rows.append(d)
    rows.append(row([lim if i % 2 else 0 for i in range(N - 1)], top_cap))
    rows.append(row([0 if i % 2 else lim for i in range(N - 1)], top_cap))
    while len(rows) < count:
        style = rng.random()
        if style < 0.3:   # mostly-max limbs with a few random ones
            lo = [lim if rng.random() < 0.8 else rng.randrange(lim + 1) for _ in range(N - 1)]
        else:
            lo = [rng.randrange(lim + 1) for _ in range(N - 1)]
        rows.append(row(lo, rng.randrange(top_cap + 1)))
    return rows


def mont_factor(curve, inf):
    """what one product divides by: R for the Montgomery fields, 1 for the plain one"""
    return 1 if curve in (2, 4) else 1 << (inf["B"] * inf["N"])


def check_out(out_row, inf, p, want_mod, k_out=1, v_out=3, slack=1 << 18, exact_digits=False):
    B, N = inf["B"], inf["N"]
    v = value(out_row, B)
    assert v % p == want_mod % p, "wrong residue"
    assert v < v_out * p, f"value bound exceeded: {v / p:.3f} p"
    for i, x in enumerate(out_row[:-1]):
        bound = k_out * (1 << B) + (0 if exact_digits else slack)
        assert int(x) < bound, f"limb {i} = {int(x):#x} exceeds its bound"


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3, 4], ids=lambda c: NAMES[c])
def test_products_at_the_operand_bounds(fc, curve):
    rng = random.Random(100 + curve)
    inf, p = fc.info(curve), P[curve]
    rinv = pow(mont_factor(curve, inf), -1, p)
    n = 512
    # (operation, bounds of a, bounds of b)
    cases = [(OP_MUL_TIGHT, (1, 3), (1, 3)),
             (OP_MUL_LAZY, (inf["KA"], 7), (inf["KB"], 5)),
             (OP_SQR_LAZY, (inf["KS"], 4), None),
             (OP_MUL_AUTO, (inf["KMAX"], 64), (inf["KMAX"], 64))]
    for op, (ka, va), bb in cases:
        a = gen(rng, inf, p, ka, va, n)
        b = gen(rng, inf, p, bb[0], bb[1], n) if bb else a
        rng.shuffle(b) if bb else None
        out = fc.run(curve, op, a, b)
        for ra, rb, ro in zip(a, b, out):
            va_, vb_ = value(ra, inf["B"]), value(rb, inf["B"])
            check_out(ro, inf, p, va_ * vb_ * rinv)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3, 4], ids=lambda c: NAMES[c])
def test_sums_differences_and_reductions(fc, curve):
    rng = random.Random(200 + curve)
    inf, p = fc.info(curve), P[curve]
    n = 512
    a = gen(rng, inf, p, 1, 3, n)
    b = gen(rng, inf, p, 1, 3, n)
    rng.shuffle(b)
    out = fc.run(curve, OP_SUB_CHAIN, a, b)
    for ra, rb, ro in zip(a, b, out):
        check_out(ro, inf, p, value(ra, inf["B"]) - 3 * value(rb, inf["B"]), exact_digits=True)
    a = gen(rng, inf, p, inf["KMAX"], 64, n)
    out = fc.run(curve, OP_REDUCE_MAX, a, a)
    for ra, ro in zip(a, out):
        check_out(ro, inf, p, value(ra, inf["B"]), exact_digits=True)
    # limbs one step below the 32-bit limit, value below 48p: the shift-and-add reduction of the
    # Solinas primes (the other fields take their usual chain)
    a2 = gen(rng, inf, p, inf["KMAX"] - 1, 48, n)
    out = fc.run(curve, OP_REDUCE_LAZY, a2, a2)
    for ra, ro in zip(a2, out):
        check_out(ro, inf, p, value(ra, inf["B"]), exact_digits=True)
    b = gen(rng, inf, p, inf["KMAX"], 64, n)
    rng.shuffle(b)
    out = fc.run(curve, OP_ADD_AUTO, a, b)
    for ra, rb, ro in zip(a, b, out):
        check_out(ro, inf, p, 2 * (value(ra, inf["B"]) + value(rb, inf["B"])), exact_digits=True)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3, 4], ids=lambda c: NAMES[c])
def test_canonical_output_is_the_unique_residue(fc, curve):
    rng = random.Random(300 + curve)
    inf, p = fc.info(curve), P[curve]
    rinv = pow(mont_factor(curve, inf), -1, p)
    a = gen(rng, inf, p, inf["KMAX"], 64, 512)
    out = fc.run(curve, OP_CANONICAL, a, a)
    for ra, ro in zip(a, out):
        got = sum(int(x) << (32 * i) for i, x in enumerate(ro[: inf["L"]]))
        assert got == value(ra, inf["B"]) * rinv % p


def _hd_divsteps_needed(p, a, limit):
    """Division steps started at delta = 1/2 (zeta = -(delta + 1/2) = -1), in Python integers: the
    number of steps until g = 0, or None if `limit` does not suffice."""
    zeta, f, g = -1, p, a
    for i in range(limit):
        if g == 0:
            assert abs(f) == 1 or a == 0
            return i
        if zeta < 0 and (g & 1):
            zeta, f, g = -zeta - 2, g, (g - f) >> 1
        else:
            zeta, g = zeta - 1, (g + (g & 1) * f) >> 1
    return i + 1 if g == 0 else None


@pytest.mark.parametrize("curve", [0, 4], ids=lambda c: NAMES[c])
def test_half_delta_division_steps_stay_inside_the_proven_bound(curve):
    """inv_gcd.hpp runs 20 batches of 30 division steps for the 256-bit fields (INV30_HD): 590 are proven
    to suffice for every odd modulus below 2^256 when delta starts at 1/2 (the bound libsecp256k1's
    modinv32 relies on).  A model of the same step rule in Python integers, over edge values and random
    ones: no input comes near the 600 steps the kernel runs."""
    rng = random.Random(4400 + curve)
    p = P[curve]
    assert p.bit_length() <= 256
    vals = [1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 255 if p >> 255 else 1 << 254, (1 << 254) - 1]
    vals += [1 << k for k in range(1, 255, 7)] + [p - (1 << k) for k in range(0, 254, 11)]
    vals += [rng.randrange(1, p) for _ in range(4000)]
    worst = 0
    for v in vals:
        need = _hd_divsteps_needed(p, v % p, 600)
        assert need is not None and need <= 590, hex(v)
        worst = max(worst, need)
    assert worst > 300          # the model really iterates


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [0, 1, 2, 3, 4], ids=lambda c: NAMES[c])
def test_division_step_inversion(fc, curve):
    """inv_gcd.hpp against Python: a * a^-1 = 1 (mod p) for edge values (1, 2, p-1, p-2, powers of
    two, all-ones patterns below p), random values, and 0 -> 0 (the reference checks its own
    safegcd against Fermat the same way, src/curve/fiat/field_macros.rs:1203-1215)."""
    rng = random.Random(400 + curve)
    inf, p = fc.info(curve), P[curve]
    N, L = inf["N"], inf["L"]
    vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << (p.bit_length() - 1), (1 << (p.bit_length() - 1)) - 1]
    vals += [1 << k for k in range(1, p.bit_length() - 1, 29)]
    vals += [(1 << k) - 1 for k in range(2, p.bit_length() - 1, 31)]
    vals += [p - (1 << k) for k in range(0, p.bit_length() - 2, 37)]
    vals += [rng.randrange(1, p) for _ in range(300)]
    if p.bit_length() <= 256:
        # the two fields that run 20 batches (600 half-delta steps, 590 proven): feed the kernel the inputs that
        # need the MOST steps in the Python model of the same rule, so the budget is exercised where it is tightest
        cand = [rng.randrange(1, p) for _ in range(3000)] + [p - 1, (p + 1) // 2, (p - 1) // 2, (1 << 254) - 1]
        vals += sorted(cand, key=lambda v: _hd_divsteps_needed(p, v, 600), reverse=True)[:24]
    vals = [v % p for v in vals]
    rows = [[(v >> (32 * i)) & 0xFFFFFFFF if i < L else 0 for i in range(N)] for v in vals]
    out = fc.run(curve, OP_INVERT, rows, rows)
    for v, ro in zip(vals, out):
        got = sum(int(x) << (32 * i) for i, x in enumerate(ro[:L]))
        assert got < p
        assert got == (pow(v, -1, p) if v else 0), hex(v)


@pytest.mark.gpu
def test_merged_products_at_the_operand_bounds(fc):
    """One Montgomery reduction for two products (round 2): u_mul_add on the general Montgomery field
    (BLS12-381: unsigned columns, the subtraction as (4p - c)*d) and u_mul_sub / u_mul_sub_2sqr on the
    sparse signed columns of P-384, with the worst limbs their column budgets admit -- every limb at
    K*2^B - 1 on both sides, zero against maximal (the most negative columns), digits of p and 2p."""
    rng = random.Random(500)
    n = 768
    # bls12_381: curve 3
    inf, p = fc.info(3), P[3]
    rinv = pow(mont_factor(3, inf), -1, p)
    B = inf["B"]
    a, b = gen(rng, inf, p, 1, 3, n), gen(rng, inf, p, 3, 6, n)
    c, d = gen(rng, inf, p, 1, 3, n), gen(rng, inf, p, 1, 3, n)
    for x in (b, c, d):
        rng.shuffle(x)
    out = fc.run4(3, OP_MUL_ADD, a, b, c, d)
    for ra, rb, rc, rd, ro in zip(a, b, c, d, out):
        check_out(ro, inf, p, (value(ra, B) * value(rb, B) - value(rc, B) * value(rd, B)) * rinv)
    a, b = gen(rng, inf, p, 3, 6, n), gen(rng, inf, p, 3, 6, n)
    c, d = gen(rng, inf, p, 2, 4, n), gen(rng, inf, p, 4, 8, n)
    for x in (b, d):
        rng.shuffle(x)
    out = fc.run4(3, OP_MUL_ADD_MAX, a, b, c, d)
    for ra, rb, rc, rd, ro in zip(a, b, c, d, out):
        check_out(ro, inf, p, (value(ra, B) * value(rb, B) + value(rc, B) * value(rd, B)) * rinv)
    # signed columns: p384r1 (sparse reduction) and bls12_381 (general reduction)
    for curve in (1, 3):
        _signed_merged_products(fc, rng, curve, n)


def _signed_merged_products(fc, rng, curve, n):
    inf, p = fc.info(curve), P[curve]
    rinv = pow(mont_factor(curve, inf), -1, p)
    B = inf["B"]
    for op, ka, kb, kc, kd in ((OP_MUL_SUB, (1, 3), (3, 6), (1, 3), (1, 3)), (OP_MUL_SUB_MAX, (2, 4), (4, 8), (2, 4), (4, 8))):
        for order in range(3):
            a, b = gen(rng, inf, p, ka[0], ka[1], n), gen(rng, inf, p, kb[0], kb[1], n)
            c, d = gen(rng, inf, p, kc[0], kc[1], n), gen(rng, inf, p, kd[0], kd[1], n)
            if order == 1:      # extremes of a*b against shuffled c*d, and the other way round
                rng.shuffle(c), rng.shuffle(d)
            elif order == 2:
                rng.shuffle(a), rng.shuffle(b)
            out = fc.run4(curve, op, a, b, c, d)
            for ra, rb, rc, rd, ro in zip(a, b, c, d, out):
                check_out(ro, inf, p, (value(ra, B) * value(rb, B) - value(rc, B) * value(rd, B)) * rinv)
    for order in range(3):
        a, b, c = gen(rng, inf, p, 2, 4, n), gen(rng, inf, p, 4, 8, n), gen(rng, inf, p, 2, 4, n)
        if order == 1:
            rng.shuffle(c)
        elif order == 2:
            rng.shuffle(a), rng.shuffle(b)
        out = fc.run4(curve, OP_MUL_SUB_2SQR, a, b, c, c)
        for ra, rb, rc, ro in zip(a, b, c, out):
            check_out(ro, inf, p, (value(ra, B) * value(rb, B) - 2 * value(rc, B) ** 2) * rinv)
