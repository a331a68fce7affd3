"""GPU parity tests proper: the HIP path, called through the C ABI (include/eccx.h), against
the CPU oracle on the same seeded inputs, against the committed golden fixtures, and — at
BASELINE.json's full batch sizes — through size-independent properties.  Bar: bit-exact
(bytes and flags; un-normalised X:Y:Z too where checked).
"""
import hashlib
import random

import numpy as np
import pytest

from eccoxide_amd import workload as W
from oracle import ecc_ref as R
from tests.oracle_lib import golden

pytestmark = pytest.mark.gpu

WEI = ["p256r1", "p384r1", "p521r1", "bls12_381_g1"]
ALL = WEI + ["ed25519"]


_SIZES = {"p256r1": (32, 32), "p384r1": (48, 48), "p521r1": (66, 66), "bls12_381_g1": (48, 32), "ed25519": (32, 32)}


def sizes(curve):
    """(field bytes, scalar bytes)"""
    return _SIZES[curve]


def gen_point_bytes(oracle, curve):
    one = (1).to_bytes(sizes(curve)[1], "big")
    return oracle.base(curve, one)[0]


def bases(oracle, curve, n, seed):
    """r_i * G, guaranteed in the prime-order subgroup."""
    rs = W.random_scalars(curve, n, seed=seed).tobytes()
    return oracle.base(curve, rs, threads=16)[0]


# ---- seeded random batches vs the oracle, incl. ragged batch sizes -----------------------
@pytest.mark.parametrize("curve", ALL)
@pytest.mark.parametrize("n", [1, 63, 257, 1500])
def test_var_matches_oracle(engine, oracle, curve, n):
    ks = W.random_scalars(curve, n, seed=100 + n).tobytes()
    pts = bases(oracle, curve, n, seed=200 + n)
    want_out, want_inf, want_proj = oracle.var(curve, ks, pts, threads=16)
    # reference-mirroring kernels: bytes, flags and the un-normalised (X:Y:Z[:T]) residues
    out, flags, proj = engine.scalarmul_var(curve, ks, pts, want_proj=True)
    assert out == want_out
    assert flags == want_inf
    assert proj == want_proj
    # default (fast, Jacobian) kernels: same affine bytes and flags
    out, flags = engine.scalarmul_var(curve, ks, pts)
    assert out == want_out
    assert flags == want_inf


@pytest.mark.parametrize("curve", ALL)
@pytest.mark.parametrize("n", [1, 65, 1500])
def test_base_matches_oracle(engine, oracle, curve, n):
    ks = W.random_scalars(curve, n, seed=300 + n).tobytes()
    want_out, want_inf, want_proj = oracle.base(curve, ks, threads=16)
    out, flags, proj = engine.scalarmul_base(curve, ks, want_proj=True)   # reference-mirroring kernels
    assert out == want_out and flags == want_inf and proj == want_proj
    out, flags = engine.scalarmul_base(curve, ks)                         # default (fast) kernels
    assert out == want_out and flags == want_inf
    out, flags = engine.scalarmul_base(curve, ks, mirror=True)            # mirror + batched normalisation
    assert out == want_out and flags == want_inf


def test_empty_batch(engine):
    for curve in ALL:
        out, flags = engine.scalarmul_var(curve, b"", b"")
        assert out == b"" and flags == b""
        out, flags = engine.scalarmul_base(curve, b"")
        assert out == b"" and flags == b""


# ---- edge scalars: 0, 1, n-1, n, n+1, all-ones ------------------------------------------
@pytest.mark.parametrize("curve", ALL)
def test_edge_scalars(engine, oracle, curve):
    fb, sb = sizes(curve)
    order = W.order(curve)
    top = (1 << 521) - 1 if curve == "p521r1" else (1 << (8 * sb)) - 1
    vals = [0, 1, 2, 15, 16, 17, order - 1, order, order + 1, top, 1 << (8 * sb - 8)]
    if curve == "p521r1":   # the 66-byte string has 7 bits above the field size: still k*P for the integer k
        vals += [(1 << 528) - 1, 1 << 527, (1 << 522) + 12345, (1 << 528) - order]
    vals = [v for v in vals if v < (1 << (8 * sb))]
    ks = b"".join(v.to_bytes(sb, "big") for v in vals)
    n = len(vals)
    g = gen_point_bytes(oracle, curve)
    pts = g + bases(oracle, curve, n - 1, seed=5)
    want = oracle.var(curve, ks, pts)
    got = engine.scalarmul_var(curve, ks, pts, want_proj=True)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    fast = engine.scalarmul_var(curve, ks, pts)
    assert fast[0] == want[0] and fast[1] == want[1]
    # k = 0 and k = n give the neutral element
    assert got[1][0] == 1 and got[1][vals.index(order)] == 1
    if curve != "ed25519":
        assert got[0][: 2 * fb] == bytes(2 * fb)
    want = oracle.base(curve, ks)
    got = engine.scalarmul_base(curve, ks, want_proj=True)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    fast = engine.scalarmul_base(curve, ks)
    assert fast[0] == want[0] and fast[1] == want[1]
    # 1 * G == G
    assert got[0][2 * fb: 4 * fb] == g


# ---- inputs that hit the special cases of the fast (Jacobian) ladder ---------------------------
@pytest.mark.parametrize("curve", WEI)
def test_fast_ladder_exceptional_cases(engine, oracle, curve):
    """The complete formulas of the reference have no special cases; the Jacobian ladder
    does.  Scalars built so that the accumulator meets the table entry:
      k = n + 2d with d = -n mod 16: before the last addition the accumulator is
          ((n+d)/16)*16*P = d*P and the digit is d         -> P + P (doubling) case
      k = n: accumulator (n-d)*P, digit d                   -> P + (-P) case
      k = 16^j * small, k with zero nibbles                 -> infinity / digit-0 cases
    """
    fb, sb = sizes(curve)
    order = W.order(curve)
    d = (-order) % 16
    assert d != 0
    # the default kernel uses signed 5-bit windows: the same construction with radix 32 and a
    # signed last digit e in [-15, 16]: 32*a = n + e, k = 32*a + e = n + 2e
    e = (-order) % 32
    if e > 16:
        e -= 32
    vals = [order + 2 * d, order + 2 * e, order - 2 * e, order, order + 1, order - 1, order + 16, order + 32,
            1 << 4, 1 << 5, 1 << 8, 1 << 10, (1 << 12) + 1, (1 << 15) + 17, 0x1001 << 64,
            31, 32, 33, 16, 17, 48, 0x210 << 40,
            2 * order + 4 * d if 2 * order + 4 * d < (1 << (8 * sb)) else order + 2 * d]
    if curve == "p521r1":
        vals += [(order + 2 * d) * 16 + 5, (order + 2 * d) * 256 + 0x50]
    vals = [v for v in vals if v < (1 << (8 * sb))]
    n = len(vals)
    ks = b"".join(v.to_bytes(sb, "big") for v in vals)
    pts = bases(oracle, curve, n, seed=31)
    want = oracle.var(curve, ks, pts)
    got = engine.scalarmul_var(curve, ks, pts)
    assert got[0] == want[0] and got[1] == want[1]
    got = engine.scalarmul_var(curve, ks, pts, mirror=True)
    assert got[0] == want[0] and got[1] == want[1]
    # fixed-base comb (Jacobian mixed additions): k = n hits P + (-P) at the top window;
    # k = 2^256 - n (BLS) / 2^522 - n (P-521) makes the running sum equal the top window's
    # entry, i.e. the P + P case of the comb
    extra = {"bls12_381_g1": [(1 << 256) - order], "p521r1": [(1 << 522) - order]}.get(curve, [])
    ks = b"".join(v.to_bytes(sb, "big") for v in vals + extra)
    want = oracle.base(curve, ks)
    got = engine.scalarmul_base(curve, ks)
    assert got[0] == want[0] and got[1] == want[1]


def test_fast_ladder_small_order_point_bls(engine, oracle):
    """(0, 2) is a point of order 3 on y^2 = x^3 + 4 (outside G1): 2P = -P, 3P = infinity, so
    the window-table build itself runs into P + (-P), infinity + P and P + P."""
    n = 64
    ks = W.random_scalars("bls12_381_g1", n, seed=41).tobytes()
    pt = (0).to_bytes(48, "big") + (2).to_bytes(48, "big")
    want = oracle.var("bls12_381_g1", ks, pt * n)
    got = engine.scalarmul_var("bls12_381_g1", ks, pt * n)
    assert got[0] == want[0] and got[1] == want[1]
    assert set(want[1]) <= {0, 1} and 1 in set(want[1])   # k = 0 mod 3 gives infinity


def _ed25519_points_outside_the_subgroup(count):
    """curve points (x, y) lifted from small y values: 7 in 8 of them carry an 8-torsion part"""
    p = 2**255 - 19
    d = (-121665 * pow(121666, -1, p)) % p
    out, y = [], 2
    while len(out) < count:
        y += 1
        xx = (y * y - 1) * pow(d * y * y + 1, -1, p) % p
        x = pow(xx, (p + 3) // 8, p)
        if (x * x - xx) % p:
            x = x * pow(2, (p - 1) // 4, p) % p
        if (x * x - xx) % p == 0:
            out.append((x, y))
    return out


def test_edwards_window_ladder_on_torsion_points(engine, oracle):
    """The windowed Edwards ladder relies on the unified formulas being complete: run it on
    points of order 1, 2, 4, 8 and on points of order 8*l, with edge and random scalars, against
    the reference's double-and-add (mirror kernel and oracle)."""
    order = W.order("ed25519")
    pts = _ed25519_points_outside_the_subgroup(24)
    le = lambda v: v.to_bytes(32, "little")
    mixed = b"".join(le(x) + le(y) for x, y in pts)
    # l * P leaves the torsion part: points of order dividing 8
    tors, fl = oracle.var("ed25519", order.to_bytes(32, "big") * len(pts), mixed)[:2]
    p = 2**255 - 19
    special = le(0) + le(1) + le(0) + le(p - 1)          # the neutral element and the point of order 2
    cases = mixed + tors + special
    n = len(cases) // 64
    rnd = W.random_scalars("ed25519", n, seed=77).tobytes()
    edge = [0, 1, 2, 7, 8, 9, 16, 31, 32, order - 1, order, order + 1, 8 * order, (1 << 256) - 1, 1 << 255]
    for ks in (rnd, b"".join(edge[i % len(edge)].to_bytes(32, "big") for i in range(n))):
        want = oracle.var("ed25519", ks, cases)
        got = engine.scalarmul_var("ed25519", ks, cases)
        assert got[0] == want[0] and got[1] == want[1]
        got = engine.scalarmul_var("ed25519", ks, cases, mirror=True)
        assert got[0] == want[0] and got[1] == want[1]
    assert len(set(tors[i:i + 64] for i in range(0, len(tors), 64))) >= 4  # several distinct torsion points


@pytest.mark.parametrize("curve", ALL)
def test_default_and_mirror_kernels_agree_on_a_large_batch(engine, oracle, curve):
    """Two different routes to the same points -- Jacobian coordinates, signed 5-bit windows and
    16-bit combs (default) against the reference's complete formulas in homogeneous coordinates,
    4-bit windows and 4-bit comb on saturated canonical limbs (mirror; its variable-base ladder
    shares the unsaturated field layer) -- must produce the same bytes and flags for 2^16 random
    units, variable and fixed base; a sample is checked against the CPU oracle as well."""
    import torch

    n = 1 << 16
    dev = torch.device("cuda", 0)
    ks = torch.from_numpy(W.random_scalars(curve, n, seed=501)).to(dev)
    rs = torch.from_numpy(W.random_scalars(curve, n, seed=502)).to(dev)
    pts, fl = engine.scalarmul_base_t(curve, rs)
    pts_m, fl_m = engine.scalarmul_base_t(curve, rs, mirror=True)
    assert torch.equal(pts, pts_m) and torch.equal(fl, fl_m)
    out, f = engine.scalarmul_var_t(curve, ks, pts)
    out_m, f_m = engine.scalarmul_var_t(curve, ks, pts, mirror=True)
    assert torch.equal(out, out_m) and torch.equal(f, f_m)
    idx = torch.arange(0, n, 257, device=dev)
    want = oracle.var(curve, ks[idx].cpu().numpy().tobytes(), pts[idx].cpu().numpy().tobytes(), threads=4)
    assert out[idx].cpu().numpy().tobytes() == want[0] and f[idx].cpu().numpy().tobytes() == want[1]


@pytest.mark.parametrize("curve", ALL)
def test_reference_layout_comb_variants(engine, oracle, curve):
    """ECCX_TABLE_IN_L2 / ECCX_TABLE_IN_LDS select the reference's 4-bit comb (table read through
    L1/L2, or -- edwards25519 -- staged in LDS, BASELINE.json configs[2]) instead of the default
    16-bit-window path: same bytes and flags, also for a batch large enough to engage the
    1024-thread LDS workgroups on every CU."""
    import torch

    dev = torch.device("cuda", 0)
    for n in (300, 1 << 17):
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=601 + n)).to(dev)
        ref_out, ref_fl = engine.scalarmul_base_t(curve, ks)
        for lds in (False, True):
            out, fl = engine.scalarmul_base_t(curve, ks, table_in_lds=lds)
            assert torch.equal(out, ref_out) and torch.equal(fl, ref_fl), (curve, n, lds)
        idx = torch.arange(0, n, max(1, n // 64), device=dev)
        want = oracle.base(curve, ks[idx].cpu().numpy().tobytes(), threads=4)
        assert ref_out[idx].cpu().numpy().tobytes() == want[0] and ref_fl[idx].cpu().numpy().tobytes() == want[1]


@pytest.mark.parametrize("curve", ["p256r1", "ed25519"])
def test_host_buffer_path_in_chunks(engine, curve):
    """eccx_scalarmul_var on host buffers sends batches of 2^17 units and more through in four
    chunks whose PCIe copies overlap the neighbouring chunk's kernels; a ragged size must give
    exactly what the device-resident path gives for the same inputs."""
    import torch

    n = (1 << 17) + 777
    ks = W.random_scalars(curve, n, seed=701)
    d_k = torch.from_numpy(ks).cuda()
    pts, _ = engine.scalarmul_base_t(curve, torch.from_numpy(W.random_scalars(curve, n, seed=702)).cuda())
    want_out, want_fl = engine.scalarmul_var_t(curve, d_k, pts)
    got = engine.scalarmul_var(curve, ks.tobytes(), pts.cpu().numpy().tobytes())
    assert got[0] == want_out.cpu().numpy().tobytes() and got[1] == want_fl.cpu().numpy().tobytes()


# ---- the reference's own known-answer vectors, on the GPU ---------------------------------
@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1"])
def test_nist_kg_on_gpu(engine, oracle, curve):
    """src/tests/kats.rs:3-40: &Point::GENERATOR * &k for the NIST vectors."""
    fb, sb = sizes(curve)
    kats = golden("nist_kg.json")[curve]
    ks = b"".join(bytes.fromhex(kv["k"]).rjust(sb, b"\0") for kv in kats)
    want = b"".join(bytes.fromhex(kv["x"]).rjust(fb, b"\0") + bytes.fromhex(kv["y"]).rjust(fb, b"\0") for kv in kats)
    out, flags = engine.scalarmul_var(curve, ks, gen_point_bytes(oracle, curve) * len(kats))
    assert out == want and flags == bytes(len(kats))
    out, flags = engine.scalarmul_base(curve, ks)
    assert out == want and flags == bytes(len(kats))


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1"])
def test_rfc6979_on_gpu(engine, curve):
    """src/protocol/ecdsa.rs:700-707 (public key = mul_base(secret)), :174 (r = x(kG) mod n)."""
    fb, sb = sizes(curve)
    v = golden("rfc6979.json")[curve]
    ks = bytes.fromhex(v["secret"].rjust(2 * sb, "0")) + b"".join(bytes.fromhex(k["k"].rjust(2 * sb, "0")) for k in v["sign_kats"])
    out, flags = engine.scalarmul_base(curve, ks)
    assert out[: 2 * fb] == bytes.fromhex(v["ux"].rjust(2 * fb, "0")) + bytes.fromhex(v["uy"].rjust(2 * fb, "0"))
    for i, kat in enumerate(v["sign_kats"], start=1):
        x = int.from_bytes(out[i * 2 * fb: i * 2 * fb + fb], "big")
        assert x % W.order(curve) == int(kat["r"], 16)


def test_rfc8032_on_gpu(engine):
    """src/protocol/ed25519.rs:271-315: seed -> clamp/reduce -> mul_base -> encoded public key."""
    order = W.order("ed25519")
    for v in golden("rfc8032.json"):
        h = bytearray(hashlib.sha512(bytes.fromhex(v["seed"])).digest()[:32])
        h[0] &= 248
        h[31] &= 127
        h[31] |= 64
        a = int.from_bytes(bytes(h), "little") % order
        out, flags = engine.scalarmul_base("ed25519", a.to_bytes(32, "big"))
        x, y = int.from_bytes(out[:32], "little"), int.from_bytes(out[32:], "little")
        assert (y | ((x & 1) << 255)).to_bytes(32, "little").hex() == v["public"]
        assert flags == b"\0"


def test_bls_g1_kats_on_gpu(engine, oracle):
    """src/curve/bls12_381/g1.rs:605-692."""
    v = golden("bls_g1.json")
    g = gen_point_bytes(oracle, "bls12_381_g1")
    for e in v["uncompressed"]:
        k = e["k"].to_bytes(32, "big")
        assert engine.scalarmul_var("bls12_381_g1", k, g)[0].hex() == e["bytes"]
        assert engine.scalarmul_base("bls12_381_g1", k)[0].hex() == e["bytes"]
    p = int(golden("params.json")["bls12_381_g1"]["p"], 16)
    for e in v["compressed"]:
        out, _ = engine.scalarmul_base("bls12_381_g1", e["k"].to_bytes(32, "big"))
        enc = bytearray(bytes.fromhex(e["bytes"]))
        sort_flag = bool(enc[0] & 0x20)
        enc[0] &= 0x1F
        assert bytes(enc) == out[:48]
        assert sort_flag == (int.from_bytes(out[48:], "big") > (p - 1) // 2)


@pytest.mark.parametrize("curve", ALL)
def test_comb_table_matches_reference(engine, curve):
    """The engine-built fixed-base table equals the reference's embedded COMB_TABLE
    (src/params/comb/<curve>.rs): SHA-256 over the whole table + sampled windows."""
    cs = golden("comb_samples.json")[curve]
    raw = engine.comb_table(curve)
    fb = cs["field_bytes"]
    assert len(raw) == cs["windows"] * 15 * 2 * fb
    assert hashlib.sha256(raw).hexdigest() == cs["sha256_xy_concat"]
    for w, entries in cs["samples"].items():
        for j, (xs, ys) in enumerate(entries):
            off = (int(w) * 15 + j) * 2 * fb
            assert raw[off: off + fb].hex() == xs and raw[off + fb: off + 2 * fb].hex() == ys


# ---- input validation (PointAffine::from_coordinate, src/curve/affine.rs:90-119) ------------
@pytest.mark.parametrize("curve", ALL)
def test_point_validation(engine, oracle, curve):
    fb, sb = sizes(curve)
    n = 8
    ks = W.random_scalars(curve, n, seed=77).tobytes()
    pts = bytearray(bases(oracle, curve, n, seed=78))
    good = bytes(pts)
    pts[2 * fb * 1 + fb - 1 if curve != "ed25519" else 2 * fb * 1] ^= 1     # unit 1: x off the curve
    pts[2 * fb * 3: 2 * fb * 3 + fb] = b"\xff" * fb                          # unit 3: x >= p (non-canonical)
    if curve == "p521r1":
        pts[2 * fb * 3] = 0x03
    out, flags = engine.scalarmul_var(curve, ks, bytes(pts), validate=True)
    want_out, want_inf, _ = oracle.var(curve, ks, good)
    for i in range(n):
        if i in (1, 3):
            assert flags[i] == 2 and out[i * 2 * fb:(i + 1) * 2 * fb] == bytes(2 * fb)
        else:
            assert flags[i] == want_inf[i] and out[i * 2 * fb:(i + 1) * 2 * fb] == want_out[i * 2 * fb:(i + 1) * 2 * fb]


@pytest.mark.parametrize("curve", ALL)
@pytest.mark.parametrize("validate", [True, False])
def test_bad_points_do_not_disturb_units_normalised_beside_them(engine, oracle, curve, validate):
    """The batched normalisation shares one inversion between the 16 (8) units of a lane, 256
    units apart.  A unit fed something that is not a curve point can reach Z = 0 (mod p) with
    non-zero limbs -- an Edwards record with x = +-y doubles to Z = 0, a Weierstrass record with
    y = p doubles to Z = 2*p*Z' -- and must neither be read as a finite point nor zero the
    inverse of its neighbours: every well-formed unit still equals the oracle, with and without
    ECCX_VALIDATE_POINTS (found by tools/soak.py, seed 777)."""
    fb, sb = sizes(curve)
    n = 4096 + 300
    ks = W.random_scalars(curve, n, seed=91).tobytes()
    good = bases(oracle, curve, n, seed=92)
    pts = bytearray(good)
    p = R.CURVES[curve].p
    order = "little" if curve == "ed25519" else "big"
    pb = 2 * fb
    patterns = [b"\xff" * pb,                                             # x = y (mod p) / not canonical
                good[:fb] + p.to_bytes(fb, order),                          # y = p: zero with non-zero digits
                bytes(pb),                                                  # (0, 0)
                (7).to_bytes(fb, order) + (7).to_bytes(fb, order),          # x = y, canonical
                (7).to_bytes(fb, order) + (p - 7).to_bytes(fb, order)]      # x = -y
    bad = list(range(3, n, 37))
    for j, i in enumerate(bad):
        pts[i * pb:(i + 1) * pb] = patterns[j % len(patterns)]
    want_out, want_inf, _ = oracle.var(curve, ks, good)
    isbad = set(bad)
    for mirror in (False, True):
        out, flags = engine.scalarmul_var(curve, ks, bytes(pts), validate=validate, mirror=mirror)
        for i in range(n):
            if i in isbad:
                if validate:
                    assert flags[i] == 2 and out[i * pb:(i + 1) * pb] == bytes(pb), (i, mirror)
            else:
                assert flags[i] == want_inf[i] and out[i * pb:(i + 1) * pb] == want_out[i * pb:(i + 1) * pb], (i, mirror)
    out, flags = engine.scalarmul_var(curve, ks, bytes(pts), validate=validate)
    if validate:
        assert (out, flags) == engine.scalarmul_var(curve, ks, bytes(pts), validate=True, mirror=True)
        # the fused double-scalar kernel shares the normalisation
        d_out, d_flags = engine.double_scalarmul(curve, bytes(len(ks)), ks, bytes(pts), validate=True)
        assert (d_out, d_flags) == (out, flags)


# ---- reference properties at test sizes (src/tests/completeness.rs:60-117) -------------------
@pytest.mark.parametrize("curve", ALL)
def test_mul_base_matches_generic(engine, oracle, curve):
    n = 512
    ks = W.random_scalars(curve, n, seed=9).tobytes()
    g = gen_point_bytes(oracle, curve)
    a = engine.scalarmul_base(curve, ks)
    b = engine.scalarmul_var(curve, ks, g * n)
    assert a == b


# ---- BASELINE.json full sizes: size-independent properties --------------------------------------
def _full_size_check(engine, oracle, curve, n, seed):
    """k_i * (r_i * G) == (k_i * r_i mod order) * G for every unit: crosses the variable-base
    ladder, the comb path and (host big-int) scalar-field arithmetic; plus a sampled
    comparison against the oracle."""
    import torch

    fb, sb = sizes(curve)
    order = W.order(curve)
    ks = W.random_scalars(curve, n, seed=seed)
    rs = W.random_scalars(curve, n, seed=seed + 1)
    d_k, d_r = torch.from_numpy(ks).cuda(), torch.from_numpy(rs).cuda()
    pts, f0 = engine.scalarmul_base_t(curve, d_r)
    out, flags = engine.scalarmul_var_t(curve, d_k, pts)
    prod = np.empty((n, sb), dtype=np.uint8)
    kb, rb = ks.tobytes(), rs.tobytes()
    buf = bytearray(n * sb)
    for i in range(n):
        v = int.from_bytes(kb[i * sb:(i + 1) * sb], "big") * int.from_bytes(rb[i * sb:(i + 1) * sb], "big") % order
        buf[i * sb:(i + 1) * sb] = v.to_bytes(sb, "big")
    d_p = torch.frombuffer(buf, dtype=torch.uint8).reshape(n, sb).cuda()
    out2, flags2 = engine.scalarmul_base_t(curve, d_p)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    assert torch.equal(flags, flags2)
    assert int(f0.max()) == 0
    # checksum of everything + sampled oracle comparison
    sample = np.random.Generator(np.random.PCG64(seed)).choice(n, size=512, replace=False)
    sample.sort()
    s_k = ks[sample].tobytes()
    s_p = pts[torch.from_numpy(sample).cuda()].cpu().numpy().tobytes()
    want_out, want_inf, _ = oracle.var(curve, s_k, s_p, threads=16)
    got = out[torch.from_numpy(sample).cuda()].cpu().numpy().tobytes()
    assert got == want_out
    assert flags[torch.from_numpy(sample).cuda()].cpu().numpy().tobytes() == want_inf


def test_full_size_p256r1(engine, oracle):
    """BASELINE.json configs[1]: p256r1 variable-base, batch = 2^20."""
    _full_size_check(engine, oracle, "p256r1", 1 << 20, seed=1000)


def test_full_size_ed25519_mul_base(engine, oracle):
    """BASELINE.json configs[2]: Ed25519 mul_base, batch = 2^20.  Properties: linearity
    (a+b)*B == a*B + b*B checked through the variable-base kernel on a slice, and a
    sampled oracle comparison."""
    import torch

    n = 1 << 20
    order = W.order("ed25519")
    ks = W.random_scalars("ed25519", n, seed=2000)
    d_k = torch.from_numpy(ks).cuda()
    out, flags = engine.scalarmul_base_t("ed25519", d_k)
    torch.cuda.synchronize()
    assert int(flags.max()) == 0
    sample = np.random.Generator(np.random.PCG64(5)).choice(n, size=1024, replace=False)
    sample.sort()
    want_out, want_inf, _ = oracle.base("ed25519", ks[sample].tobytes(), threads=16)
    assert out[torch.from_numpy(sample).cuda()].cpu().numpy().tobytes() == want_out
    # k*B via comb == k*B via double-and-add for a 2^16 slice (curve25519.rs:1373-1387)
    m = 1 << 16
    g = oracle.base("ed25519", (1).to_bytes(32, "big"))[0]
    d_g = torch.frombuffer(bytearray(g * m), dtype=torch.uint8).reshape(m, 64).cuda()
    out2, flags2 = engine.scalarmul_var_t("ed25519", d_k[:m].contiguous(), d_g)
    torch.cuda.synchronize()
    assert torch.equal(out[:m], out2)


@pytest.mark.parametrize("curve,n", [("p384r1", 1 << 19), ("p521r1", 1 << 19), ("bls12_381_g1", 1 << 20)])
def test_full_size_other_curves(engine, oracle, curve, n):
    """BASELINE.json configs[3]/[4] at their per-GPU sizes: p384r1 / p521r1 2^22 over 8 GPUs =
    2^19 per GPU, BLS12-381 G1 2^20 on one GPU."""
    _full_size_check(engine, oracle, curve, n, seed=3000 + n)


def test_whole_config_3_batch_on_one_gpu(engine, oracle):
    """BASELINE.json configs[3] is 2^22 p384r1 units over 8 GPUs; the whole batch also fits ONE GPU (1 GB of
    inputs and outputs, 0.6 GB of window tables): the same size-independent check at 2^22, eight times the per-GPU share."""
    _full_size_check(engine, oracle, "p384r1", 1 << 22, seed=7000)


def test_two_contexts_on_two_streams(oracle):
    """Contexts are independent (own scratch slab and tables): two of them driven from two
    HIP streams at once give the same bytes as one after the other."""
    import torch

    import eccoxide_amd as E

    n = 1 << 14
    ks = torch.from_numpy(W.random_scalars("p256r1", n, seed=81)).cuda()
    rs = torch.from_numpy(W.random_scalars("p256r1", n, seed=82)).cuda()
    with E.Engine(0) as e1, E.Engine(0) as e2:
        pts, _ = e1.scalarmul_base_t("p256r1", rs)
        torch.cuda.synchronize()
        ref_a, _ = e1.scalarmul_var_t("p256r1", ks, pts)
        ref_b, _ = e1.scalarmul_var_t("p256r1", rs, pts)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s1):
            out_a, _ = e1.scalarmul_var_t("p256r1", ks, pts, stream=s1.cuda_stream)
        with torch.cuda.stream(s2):
            out_b, _ = e2.scalarmul_var_t("p256r1", rs, pts, stream=s2.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)
    want = oracle.var("p256r1", ks[:64].cpu().numpy().tobytes(), pts[:64].cpu().numpy().tobytes())[0]
    assert ref_a[:64].cpu().numpy().tobytes() == want


def test_sharded_entry_point_single_gpu(engine, oracle):
    """eccx_scalarmul_var_sharded with one context == the plain call."""
    import ctypes

    lib = engine._lib
    n = 300
    ks = W.random_scalars("p256r1", n, seed=4).tobytes()
    pts = bases(oracle, "p256r1", n, seed=6)
    out = ctypes.create_string_buffer(n * 64)
    fl = ctypes.create_string_buffer(n)
    arr = (ctypes.c_void_p * 1)(engine._ctx.value)
    rc = lib.eccx_scalarmul_var_sharded(arr, 1, 0, n, ks, pts, out, fl, 0)
    assert rc == 0
    want = oracle.var("p256r1", ks, pts)
    assert out.raw == want[0] and fl.raw == want[1]


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "ed25519"])
def test_device_buffers_at_any_alignment(engine, oracle, curve):
    """Field elements move as 16-byte words, 4-byte words or bytes depending on the address
    (fe.hpp byte I/O): tensor views at offsets 0, 1, 2, 4, 8 and 12 give the same bytes."""
    import torch

    fb, sb = sizes(curve)
    n = 700
    ks = W.random_scalars(curve, n, seed=5).tobytes()
    pts = bases(oracle, curve, n, seed=6)
    want_v = oracle.var(curve, ks, pts)
    want_b = oracle.base(curve, ks)
    dev = torch.device("cuda", 0)

    def view(data, off, fill=0):
        buf = torch.full((len(data) + 64,), fill, dtype=torch.uint8, device=dev)
        buf[off:off + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        return buf, buf[off:off + len(data)]

    for off in (0, 1, 2, 4, 8, 12):
        _, k_t = view(ks, off)
        _, p_t = view(pts, (off * 3) % 16)
        obuf, o_t = view(bytes(n * 2 * fb), off, fill=0xAA)
        fbuf, f_t = view(bytes(n), (off + 1) % 16, fill=0xAA)
        engine.scalarmul_var_t(curve, k_t, p_t, o_t, f_t)
        torch.cuda.synchronize()
        assert o_t.cpu().numpy().tobytes() == want_v[0] and f_t.cpu().numpy().tobytes() == want_v[1], off
        assert bool((obuf[:off] == 0xAA).all()) and bool((obuf[off + n * 2 * fb:] == 0xAA).all())
        engine.scalarmul_base_t(curve, k_t, o_t, f_t)
        torch.cuda.synchronize()
        assert o_t.cpu().numpy().tobytes() == want_b[0] and f_t.cpu().numpy().tobytes() == want_b[1], off
        s_t, sf_t = engine.point_add_t(curve, p_t, o_t)
        torch.cuda.synchronize()
        assert (s_t.cpu().numpy().tobytes(), sf_t.cpu().numpy().tobytes()) == engine.point_add(curve, pts, want_b[0])
