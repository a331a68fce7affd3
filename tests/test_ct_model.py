"""CPU checks of the window bounds the secret-scalar kernels rely on (tests/ct_model.py).

Exhaustive on small parameters: for every scalar of SB bytes, every collision accumulator == +-entry
falls inside the windows that carry the collision selects -- ct_unsafe_windows() for the fixed-base
comb (kernels_ct.hpp), coz_full_windows() for the variable-base ladder on a point of prime order
(kernels_coz.hpp).  And on the real curves: the constructed collision scalars exist (so the GPU tests
have something to bite on) and respect the same bounds.
"""
import json
import os

import pytest

from tests import ct_model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIMES = [257, 263, 509, 521, 1021, 2039, 4093, 8191, 16381, 32749, 65521, 61, 127]


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6])
def test_booth_digits_reconstruct_the_scalar(w):
    for k in list(range(0, 65536, 7)) + [65535, 65534, 32768]:
        ds = M.booth_digits(k, 2, w)
        assert all(abs(d) <= 1 << (w - 1) for d in ds)


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("n", PRIMES)
def test_fixed_base_collisions_stay_in_the_unsafe_windows(w, n):
    sb = 2
    if n <= 1 << w:  # the model needs entries d G distinct from each other and from infinity
        pytest.skip("order below the window's digit range")
    nwin = (8 * sb + 1 + w - 1) // w
    unsafe = M.fixed_base_unsafe_windows(sb, w, n.bit_length())
    assert 1 <= unsafe <= nwin
    hits = 0
    for k in range(1 << (8 * sb)):
        for win, _ in M.fixed_base_events(k, sb, w, n):
            assert win >= nwin - unsafe, (k, win, nwin, unsafe)
            hits += 1
    # the bound is not vacuous: collisions do occur for orders that are small against 2^(8 SB)
    if n.bit_length() <= 8 * sb - 2:
        assert hits > 0


@pytest.mark.parametrize("wb", [3, 4, 5])
@pytest.mark.parametrize("n", PRIMES)
def test_var_base_collisions_stay_in_the_full_windows(wb, n):
    sb = 2
    if n <= 1 << wb:
        pytest.skip("order below the table's range")
    full = M.var_base_full_windows(sb, wb, n.bit_length())
    hits = 0
    for k in range(1 << (8 * sb)):
        for win, _ in M.var_base_events(k, sb, wb, n):
            assert win < full, (k, win, full)
            hits += 1
    assert hits > 0  # k = n + 2 d always exists for n < 2^(8 SB) - 32


CURVES = {"p256r1": (32, 256), "p384r1": (48, 384), "p521r1": (66, 521), "bls12_381_g1": (32, 255)}


@pytest.mark.parametrize("curve", sorted(CURVES))
def test_real_curve_collision_scalars(curve):
    sb, nbits = CURVES[curve]
    n = int(json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))[curve]["order"], 16)
    assert n.bit_length() == nbits
    w = 6
    nwin = (8 * sb + 1 + w - 1) // w
    unsafe = M.fixed_base_unsafe_windows(sb, w, nbits)
    for k, ev in M.collision_scalars_fixed_base(sb, w, n, limit=16).items():
        assert all(win >= nwin - unsafe for win, _ in ev), (hex(k), ev)
    for wb in (4, 5):
        full = M.var_base_full_windows(sb, wb, nbits)
        found = M.collision_scalars_var_base(sb, wb, n, limit=16)
        assert found, "k = n + 2 d must collide in the last window"
        for k, ev in found.items():
            assert all(win < full for win, _ in ev), (hex(k), ev)


@pytest.mark.parametrize("sb,w", [(32, 6), (48, 6), (66, 6), (32, 5), (32, 7), (48, 5), (66, 4)])
def test_every_reachable_digit_has_a_table_scalar(sb, w):
    """The secret-scalar comb's table is built from SB-byte scalars d 2^(w i) (ensure_comb_ct, eccx_api.cpp): every
    digit the recoding can produce must name a scalar below 2^(8 SB) -- except the single entry 2^(8 SB) itself
    (the top window's digit 2^(8 SB - w i_top); a scalar of all ones recodes to it), which the build makes by a
    doubling of 2^(8 SB - 1) G."""
    top = 1 << (8 * sb)
    nwin = (8 * sb + 1 + w - 1) // w
    shift = 8 * sb - w * (nwin - 1)
    assert 0 <= shift < w
    seen_special = False
    for k in (top - 1, top - 2, top >> 1, (top >> 1) - 1, (top >> 1) + 1, 1, 0, int("f" * (2 * sb - 1), 16)) + tuple(
            (top - 1) ^ (1 << j) for j in range(0, 8 * sb, 5)):
        for i, d in enumerate(M.booth_digits(k, sb, w)):
            e = abs(d) << (w * i)
            assert e <= top
            if e == top:
                assert i == nwin - 1 and d == 1 << shift
                seen_special = True
    assert seen_special


def test_tagged_limbs_survive_the_floating_point_select():
    """kernels_ct.hpp ct_scan_lds_pk selects table words with v_pk_fma_f32: acc = e * m + acc, m = 1.0 for the lane's
    digit and 0.0 otherwise, on limbs tagged with bit 30.  What the kernel relies on, checked here in IEEE binary32
    (numpy): a tagged limb is a positive normal number; e * 1 + 0 and e * 0 + acc reproduce bit patterns exactly; the
    accumulator of a digit that matches no entry stays +0.  Limbs: every value below 2^29 that matters at the edges,
    top limbs up to 3 * 2^28 - 1 (P-521's, the largest), and random ones."""
    import numpy as np

    TAG = np.uint32(0x40000000)
    rng = np.random.default_rng(5)
    limbs = np.concatenate([
        np.array([0, 1, 2, (1 << 23) - 1, 1 << 23, (1 << 28) - 1, 1 << 28, (1 << 29) - 1, 1 << 29, 3 * (1 << 28) - 1], dtype=np.uint32),
        rng.integers(0, 3 << 28, size=200000, dtype=np.uint32)])
    tagged = limbs | TAG
    f = tagged.view(np.float32)
    assert np.all(np.isfinite(f)) and np.all(f >= np.float32(2.0))         # normal, positive: exponent field 10xxxxxx / 110xxxxx
    expo = (tagged >> np.uint32(23)) & np.uint32(0xFF)
    assert expo.min() >= 128 and expo.max() < 255
    one, zero = np.float32(1.0), np.float32(0.0)
    with np.errstate(all="raise"):                                          # no overflow, underflow or invalid operation
        kept = f * one + zero                                               # the lane's entry arrives on a +0 accumulator
        assert np.array_equal(kept.view(np.uint32), tagged)
        other = np.roll(f, 1)
        after = other * zero + kept                                         # the entries after it leave it alone
        assert np.array_equal(after.view(np.uint32), tagged)
        nothing = f * zero + zero                                           # digit 0: nothing selected
        assert np.all(nothing.view(np.uint32) == 0)
    assert np.array_equal(after.view(np.uint32) & ~TAG, limbs)              # the tag comes off
    assert np.all((nothing.view(np.uint32) & ~TAG) == 0)


@pytest.mark.parametrize("x", [10, 12])
@pytest.mark.parametrize("wb", [3, 4, 5])
def test_glv_ladder_collisions_stay_in_the_bottom_window(x, wb):
    """The endomorphism ladder under ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP (kernels_coz.hpp, GLV and CT) resolves
    accumulator == +-entry in window 0 only.  Exhaustive on two small members of the BLS12 family (r = x^4 - x^2 + 1
    prime, the scalar range 3.2-3.3 r where the real curve has 2.2 r): every scalar, every addition."""
    r = x**4 - x * x + 1
    assert all(r % q for q in range(2, int(r**0.5) + 1))
    x2 = x * x
    sbits = r.bit_length() + 1
    kbits = (((1 << sbits) - 1) // x2).bit_length()
    seen = set()
    for k in range(1 << sbits):
        for win, half, kind in M.glv_events(k, x2, r, kbits, wb):
            assert win == 0, (k, win, half, kind)
            seen.add(kind)
    assert seen == {"twice", "cancel"}


def test_real_bls12_381_glv_collision_scalars():
    found = M.collision_scalars_glv(M.BLS_X2, M.BLS_R, 256, 129, 4)
    assert M.BLS_R in found and 2 * M.BLS_R in found                     # k = r, 2 r: the last addition cancels
    kinds = {kind for ev in found.values() for _, _, kind in ev}
    assert kinds == {"twice", "cancel"}
    assert all(win == 0 for ev in found.values() for win, _, _ in ev)
