"""GPU parity tests of the secret-scalar kernels (ECCX_CT_SCAN, round 3): the fixed-base combs of
kernels_ct.hpp and the scanning affine-table ladder of kernels_coz.hpp (CT = true), through the C ABI,
bit for bit against the oracle.

What the reference pins here: Point::mul_base and &Point * &Scalar are ONE function each -- the
constant-time one (src/curve/projective.rs:871-896, :965-981; src/curve/curve25519.rs:840-869) -- so
every vector of tests/golden applies (NIST kG, RFC 6979 / 8032 public keys, G1 encodings).  On top of
them: the scalars that drive the kernels' select-only special cases (tests/ct_model.py finds them),
whole wavefronts of identical scalars (where a compiler-made branch on a lane mask would be uniform
and could skip work), low-order BLS12-381 bases, and the full-size batches of BASELINE.json.
"""
import json
import os

import numpy as np
import pytest

from eccoxide_amd import workload as W
from tests import ct_model as M
from tests.oracle_lib import ROOT

pytestmark = pytest.mark.gpu

WEI = ["p256r1", "p384r1", "p521r1", "bls12_381_g1"]
ALL = WEI + ["ed25519"]
_SIZES = {"p256r1": (32, 32), "p384r1": (48, 48), "p521r1": (66, 66), "bls12_381_g1": (48, 32), "ed25519": (32, 32)}
CT_BASE_BITS = 6    # ECCX_CT_BASE_BITS (edwards25519: 5; it has no collision cases)
CT_GATHER_BITS = 7  # ECCX_CT_GATHER_BITS
CT_VAR_BITS = 4     # ECCX_CT_VAR_BITS
LOOKUPS = [{"ct_scan": True}, {"ct_gather": True}]  # the strict scan and the opt-in cross-lane gather


def _bases(oracle, curve, n, seed):
    return oracle.base(curve, W.random_scalars(curve, n, seed=seed).tobytes(), threads=16)[0]


def _pack(vals, sb):
    return b"".join((v % (1 << (8 * sb))).to_bytes(sb, "big") for v in vals)


def _edge_scalars(curve):
    fb, sb = _SIZES[curve]
    order = W.order(curve)
    top = 1 << (8 * sb)
    vals = [0, 1, 2, 3, order - 1, order, order + 1, order - 2, top - 1, top - 2, top >> 1, (top >> 1) - 1, (top >> 1) + 1]
    # every window alone at its extreme digits, for both window widths in use
    for w in (CT_BASE_BITS, CT_GATHER_BITS, CT_VAR_BITS, 5):
        for i in range(0, (8 * sb + w) // w):
            for d in (1, (1 << (w - 1)) - 1, 1 << (w - 1), (1 << (w - 1)) + 1, (1 << w) - 1):
                v = d << (w * i)
                if v < top:
                    vals.append(v)
    # runs of ones (long borrow chains in the Booth recoding) and alternating patterns
    vals += [(1 << j) - 1 for j in range(1, 8 * sb + 1, 7)] + [int("a5" * sb, 16), int("5a" * sb, 16), int("f0" * sb, 16)]
    return [v for v in vals if 0 <= v < top]


@pytest.mark.parametrize("lookup", LOOKUPS, ids=["scan", "gather"])
@pytest.mark.parametrize("curve", ALL)
def test_ct_fixed_base_edge_scalars(engine, oracle, curve, lookup):
    fb, sb = _SIZES[curve]
    ks = _pack(_edge_scalars(curve), sb)
    want = oracle.base(curve, ks, threads=16)
    got = engine.scalarmul_base(curve, ks, **lookup)
    assert got[0] == want[0] and got[1] == want[1]


@pytest.mark.parametrize("curve", WEI)
def test_ct_fixed_base_collision_scalars(engine, oracle, curve):
    """Scalars for which the comb's accumulator meets +-(its next table entry): the select-only doubling
    (from the entry's affine coordinates) and cancellation of the top windows -- found for both window widths."""
    fb, sb = _SIZES[curve]
    order = W.order(curve)
    vals = set()
    for bits in (CT_BASE_BITS, CT_GATHER_BITS):
        found = M.collision_scalars_fixed_base(sb, bits, order, limit=24)
        kinds = {kind for ev in found.values() for _, kind in ev}
        assert "cancel" in kinds  # k = n is always there
        vals |= set(found)
    ks = _pack(sorted(vals), sb)
    want = oracle.base(curve, ks, threads=16)
    for lookup in LOOKUPS:
        got = engine.scalarmul_base(curve, ks, **lookup)
        assert got[0] == want[0] and got[1] == want[1]
    assert engine.scalarmul_base(curve, ks) == got


@pytest.mark.parametrize("curve", WEI)
def test_ct_var_base_edge_and_collision_scalars(engine, oracle, curve):
    fb, sb = _SIZES[curve]
    order = W.order(curve)
    found = M.collision_scalars_var_base(sb, CT_VAR_BITS, order, limit=48)
    kinds = {kind for ev in found.values() for _, kind in ev}
    assert kinds == {"twice", "cancel"}
    vals = sorted(found) + _edge_scalars(curve)
    ks = _pack(vals, sb)
    pts = _bases(oracle, curve, len(vals), seed=911)
    want = oracle.var(curve, ks, pts, threads=16)
    got = engine.scalarmul_var(curve, ks, pts, ct_scan=True)
    assert got[0] == want[0] and got[1] == want[1]


@pytest.mark.parametrize("curve", ALL)
def test_ct_whole_wavefronts_of_one_scalar(engine, oracle, curve):
    """Every lane of a wavefront holds the same scalar: any lane mask derived from a digit is then all-ones
    or all-zeros.  (The scan written with ?: compiled to EXEC-masked regions that a uniform mask skips --
    results stay right either way, this pins that the asm selects kept them right.)"""
    fb, sb = _SIZES[curve]
    order = W.order(curve)
    pt1 = _bases(oracle, curve, 1, seed=921)
    for v in (0, 1, order - 1, order, (1 << (8 * sb)) - 1, int("a5" * sb, 16) % order, 1 << (8 * sb - 3)):
        ks = _pack([v], sb) * 192
        want = oracle.base(curve, ks[:sb], threads=1)
        for lookup in LOOKUPS:
            got = engine.scalarmul_base(curve, ks, **lookup)
            assert got[0] == want[0] * 192 and got[1] == want[1] * 192
        wantv = oracle.var(curve, ks[:sb], pt1, threads=1)
        gotv = engine.scalarmul_var(curve, ks, pt1 * 192, ct_scan=True)
        assert gotv[0] == wantv[0] * 192 and gotv[1] == wantv[1] * 192


def test_ct_golden_vectors(engine):
    """The reference's own known answers through the secret-scalar kernels: NIST kG as fixed base and as
    variable base with base = G (src/tests/kats_data.rs), RFC 6979 public keys (src/protocol/ecdsa.rs:808-915)."""
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "nist_kg.json")))
    params = json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))
    for curve in ("p256r1", "p384r1", "p521r1"):
        fb, sb = _SIZES[curve]
        vec = kats[curve]
        ks = _pack([int(v["k"], 16) if isinstance(v["k"], str) else int(v["k"]) for v in vec], sb)
        want = b"".join(int(v["x"], 16).to_bytes(fb, "big") + int(v["y"], 16).to_bytes(fb, "big") for v in vec)
        for lookup in LOOKUPS:
            got = engine.scalarmul_base(curve, ks, **lookup)
            assert got[0] == want and set(got[1]) == {0}
        g = int(params[curve]["gx"], 16).to_bytes(fb, "big") + int(params[curve]["gy"], 16).to_bytes(fb, "big")
        got = engine.scalarmul_var(curve, ks, g * len(vec), ct_scan=True)
        assert got[0] == want and set(got[1]) == {0}


def test_ct_bls_low_order_bases(engine, oracle):
    """BLS12-381 has curve points outside G1, of order 3, 11, 33, ...: order <= 8 degenerates the table build
    (marked from the base point, redone by the reference-mirroring scan kernel), the others meet
    accumulator == +-entry in the middle of the ladder, which is why every window of this curve carries
    the collision selects."""
    curve = "bls12_381_g1"
    p = int(json.load(open(os.path.join(ROOT, "tests", "golden", "params.json")))[curve]["p"], 16)
    h = 0x396C8C005555E1568C00AAAB0000AAAB
    r = W.order(curve)

    def lift(x):
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        return (x, y) if y * y % p == rhs else None

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(k, P):
        acc = None
        for bit in bin(k)[2:]:
            acc = add(acc, acc)
            if bit == "1":
                acc = add(acc, P)
        return acc

    pts = []
    x = 1
    while len(pts) < 6:
        x += 1
        P = lift(x)
        if P is None:
            continue
        Q = mul(r, P)  # kills the G1 part: what is left has order dividing h
        for small in (3, 11, 33, 121, 363):
            T = mul(h // small, Q) if h % small == 0 else None
            if T is not None and mul(small, T) is None:
                pts.append(T)
    pts = pts[:12] + [(0, 2)]  # (0, 2): order 3
    enc = [a.to_bytes(48, "big") + b.to_bytes(48, "big") for a, b in pts]
    n = 256
    ks = W.random_scalars(curve, n, seed=931).tobytes()
    g1 = _bases(oracle, curve, n, seed=932)
    mixed = b"".join(enc[i % len(enc)] if i % 3 == 0 else g1[i * 96:(i + 1) * 96] for i in range(n))
    want = oracle.var(curve, ks, mixed, threads=16)
    got = engine.scalarmul_var(curve, ks, mixed, ct_scan=True)
    assert got[0] == want[0] and got[1] == want[1]
    assert 1 in set(want[1])  # some k is a multiple of a small order


def test_ct_bls_subgroup_bases_narrow_ladder(engine, oracle):
    """ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP on bls12_381_g1 (sk * H(m)): the secret-scalar ladder that handles
    accumulator == +-entry only where a base of prime order r can reach it.  Bases in G1, every edge scalar and the
    scalars that do collide there (k = r, k = 2^256 - 1, ...), whole wavefronts of them, against the oracle."""
    curve = "bls12_381_g1"
    fb, sb = _SIZES[curve]
    vals = _edge_scalars(curve) + sorted(M.collision_scalars_var_base(sb, CT_VAR_BITS, W.order(curve), limit=48))
    vals += [W.order(curve) * j + d for j in (1, 2) for d in (-2, -1, 0, 1, 2, 8, 16)]
    glv = M.collision_scalars_glv(M.BLS_X2, M.BLS_R, 256, 129, CT_VAR_BITS)  # the two-half ladder's own collisions
    assert {kind for ev in glv.values() for _, _, kind in ev} == {"twice", "cancel"}
    vals += sorted(glv) + [k + d * M.BLS_X2 for k in (M.BLS_R, 2 * M.BLS_R) for d in (-8, -1, 1, 7, 8)] + [M.BLS_X2 - 1, M.BLS_X2, M.BLS_X2 + 1, 8 * M.BLS_X2, (M.BLS_X2 << 128) % (1 << 256)]
    vals = [v for v in vals if 0 <= v < (1 << (8 * sb))]
    ks = _pack(vals, sb)
    pts = _bases(oracle, curve, len(vals), seed=21)
    want = oracle.var(curve, ks, pts, threads=16)
    got = engine.scalarmul_var(curve, ks, pts, ct_scan=True, assume_subgroup=True)
    assert got == (want[0], want[1])
    # one scalar per wavefront, 64 different G1 bases each
    pick = [W.order(curve), W.order(curve) - 1, (1 << 256) - 1, 0, 1] + sorted(glv)
    ks = b"".join(_pack([v], sb) * 64 for v in pick)
    pts = _bases(oracle, curve, 64 * len(pick), seed=22)
    want = oracle.var(curve, ks, pts, threads=16)
    assert engine.scalarmul_var(curve, ks, pts, ct_scan=True, assume_subgroup=True) == (want[0], want[1])
    # 2^16 random units: same bytes as the general secret-scalar ladder
    n = 1 << 16
    ks = W.random_scalars(curve, n, seed=23).tobytes()
    pts = _bases(oracle, curve, n, seed=24)
    assert engine.scalarmul_var(curve, ks, pts, ct_scan=True, assume_subgroup=True) == engine.scalarmul_var(curve, ks, pts, ct_scan=True)
    # on a cofactor-1 curve the option changes nothing
    ks = W.random_scalars("p256r1", 200, seed=25).tobytes()
    pts = _bases(oracle, "p256r1", 200, seed=26)
    assert engine.scalarmul_var("p256r1", ks, pts, ct_scan=True, assume_subgroup=True) == engine.scalarmul_var("p256r1", ks, pts, ct_scan=True)


@pytest.mark.parametrize("curve", WEI)
def test_ct_rejected_and_garbage_bases(engine, oracle, curve):
    """ECCX_VALIDATE_POINTS under ECCX_CT_SCAN: rejected records keep flag 2 and zero bytes, and a record that is
    not a curve point disturbs no unit normalised beside it."""
    fb, sb = _SIZES[curve]
    n = 300
    ks = W.random_scalars(curve, n, seed=941).tobytes()
    pts = bytearray(_bases(oracle, curve, n, seed=942))
    bad = [3, 64, 65, 130, 299]
    for i in bad:
        pts[i * 2 * fb + fb - 1] ^= 1  # off the curve
    pts = bytes(pts)
    got = engine.scalarmul_var(curve, ks, pts, ct_scan=True, validate=True)
    want = oracle.var(curve, ks, pts, threads=16)
    for i in range(n):
        if i in bad:
            assert got[1][i] == 2 and got[0][i * 2 * fb:(i + 1) * 2 * fb] == bytes(2 * fb)
        else:
            assert got[1][i] == want[1][i] and got[0][i * 2 * fb:(i + 1) * 2 * fb] == want[0][i * 2 * fb:(i + 1) * 2 * fb]
    got2 = engine.scalarmul_var(curve, ks, pts, ct_scan=True)  # no validation: the neighbours must still be right
    for i in range(n):
        if i not in bad:
            assert got2[0][i * 2 * fb:(i + 1) * 2 * fb] == want[0][i * 2 * fb:(i + 1) * 2 * fb]
    if curve == "bls12_381_g1":  # the same through the secret-scalar endomorphism ladder (the good bases are in G1)
        got3 = engine.scalarmul_var(curve, ks, pts, ct_scan=True, assume_subgroup=True, validate=True)
        got4 = engine.scalarmul_var(curve, ks, pts, ct_scan=True, assume_subgroup=True)
        for i in range(n):
            if i in bad:
                assert got3[1][i] == 2 and got3[0][i * 2 * fb:(i + 1) * 2 * fb] == bytes(2 * fb)
            else:
                assert got3[1][i] == want[1][i] and got3[0][i * 2 * fb:(i + 1) * 2 * fb] == want[0][i * 2 * fb:(i + 1) * 2 * fb]
                assert got4[0][i * 2 * fb:(i + 1) * 2 * fb] == want[0][i * 2 * fb:(i + 1) * 2 * fb]


@pytest.mark.parametrize("curve", ALL)
def test_ct_ragged_batches(engine, oracle, curve):
    fb, sb = _SIZES[curve]
    for n in (1, 63, 64, 65, 255, 257, 1025):
        ks = W.random_scalars(curve, n, seed=950 + n).tobytes()
        want = oracle.base(curve, ks, threads=16)
        for lookup in LOOKUPS:
            got = engine.scalarmul_base(curve, ks, **lookup)
            assert got[0] == want[0] and got[1] == want[1]
        pts = want[0] if curve != "ed25519" else _bases(oracle, curve, n, seed=3)
        ks2 = W.random_scalars(curve, n, seed=960 + n).tobytes()
        wantv = oracle.var(curve, ks2, pts, threads=16)
        gotv = engine.scalarmul_var(curve, ks2, pts, ct_scan=True)
        assert gotv[0] == wantv[0] and gotv[1] == wantv[1]


def test_ct_option_combinations(engine):
    import eccoxide_amd as E

    import ctypes

    out, fl = ctypes.create_string_buffer(64), ctypes.create_string_buffer(1)
    with pytest.raises(E.EccxError) as ei:
        engine._check(engine._lib.eccx_scalarmul_base(engine._ctx, 4, 1, bytes(32), out, fl, None, (1 << 8) | (1 << 2)))
    assert "TABLE_IN_LDS" in str(ei.value)
    # the reference-mirroring scan stays reachable: same bytes
    k = W.random_scalars("p256r1", 70, seed=5).tobytes()
    a = engine.scalarmul_base("p256r1", k, ct_scan=True)
    b = engine.scalarmul_base("p256r1", k, ct_scan=True, mirror=True)
    assert a == b


def test_engine_level_secrecy_selects_the_secret_scalar_kernels(oracle):
    """Engine(device, secret_scalars=True): calls that do not say ct_scan= take the secret-scalar kernels -- visible from
    outside through an option those kernels refuse (ECCX_ASSUME_SUBGROUP) -- and an explicit ct_scan=False still wins."""
    import eccoxide_amd as E

    with E.Engine(0, secret_scalars=True) as eng:
        k = W.random_scalars("bls12_381_g1", 70, seed=9).tobytes()
        pts = _bases(oracle, "bls12_381_g1", 70, seed=10)
        want = oracle.var("bls12_381_g1", k, pts, threads=8)
        assert eng.scalarmul_var("bls12_381_g1", k, pts) == (want[0], want[1])
        assert eng.scalarmul_var("bls12_381_g1", k, pts, assume_subgroup=True, ct_scan=False) == (want[0], want[1])
        kb = W.random_scalars("ed25519", 70, seed=11).tobytes()
        wb = oracle.base("ed25519", kb, threads=8)
        assert eng.scalarmul_base("ed25519", kb) == (wb[0], wb[1])
        import torch

        dev = torch.device("cuda", 0)
        kt = torch.frombuffer(bytearray(kb), dtype=torch.uint8).to(dev)
        with pytest.raises(E.EccxError) as ei:  # the LDS-resident comb indexes by the digit: refused for secret scalars
            eng.scalarmul_base_t("ed25519", kt, table_in_lds=True)
        assert "TABLE_IN_LDS" in str(ei.value)
        out, fl = eng.scalarmul_base_t("ed25519", kt, table_in_lds=True, ct_scan=False)
        torch.cuda.synchronize()
        assert out.cpu().numpy().tobytes() == wb[0]


@pytest.mark.parametrize("curve,log2n", [("p256r1", 20), ("ed25519", 20), ("p384r1", 19), ("p521r1", 19), ("bls12_381_g1", 20)])
def test_ct_full_size_batches_match_the_default_kernels(engine, oracle, curve, log2n):
    """BASELINE.json's per-GPU batch sizes: the secret-scalar kernels against the default kernels on every unit,
    and a sample against the oracle."""
    import torch

    fb, sb = _SIZES[curve]
    n = 1 << log2n
    dev = torch.device("cuda:0")
    ks = torch.from_numpy(W.random_scalars(curve, n, seed=971)).to(dev)
    engine.prepare(curve, base=True, ct=True, ct_gather=True)
    engine.reserve(curve, n, var=True, ct=True)
    ob, fl_b = engine.scalarmul_base_t(curve, ks)
    oc, fl_c = engine.scalarmul_base_t(curve, ks, ct_scan=True)
    og, fl_g = engine.scalarmul_base_t(curve, ks, ct_gather=True)
    torch.cuda.synchronize()
    assert torch.equal(ob, oc) and torch.equal(fl_b, fl_c)
    assert torch.equal(ob, og) and torch.equal(fl_b, fl_g)
    sample = np.arange(0, n, n // 257)
    kb = ks.cpu().numpy().reshape(n, sb)[sample].tobytes()
    want = oracle.base(curve, kb, threads=16)
    assert oc.cpu().numpy().reshape(n, 2 * fb)[sample].tobytes() == want[0]
    pts = ob  # k_i G: bases in the prime-order subgroup
    k2 = torch.from_numpy(W.random_scalars(curve, n, seed=972)).to(dev)
    ov, fv = engine.scalarmul_var_t(curve, k2, pts)
    oct_, fct = engine.scalarmul_var_t(curve, k2, pts, ct_scan=True)
    torch.cuda.synchronize()
    assert torch.equal(ov, oct_) and torch.equal(fv, fct)
    kv = k2.cpu().numpy().reshape(n, sb)[sample].tobytes()
    pv = pts.cpu().numpy().reshape(n, 2 * fb)[sample].tobytes()
    wantv = oracle.var(curve, kv, pv, threads=16)
    assert oct_.cpu().numpy().reshape(n, 2 * fb)[sample].tobytes() == wantv[0]


def test_prepare_waits_for_work_on_other_streams(engine, oracle):
    """eccx_prepare builds tables with the context's scratch slab and row buffer: a batch of another curve still
    running on a user stream must not be disturbed (the build only waited for its own stream)."""
    import torch

    import eccoxide_amd as E

    with E.Engine(0) as eng:
        curve = "p256r1"
        n = 1 << 16
        dev = torch.device("cuda:0")
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=981)).to(dev)
        eng.prepare(curve)
        pts, _ = eng.scalarmul_base_t(curve, ks)
        torch.cuda.synchronize()
        eng.reserve(curve, n)
        user = torch.cuda.Stream()
        k2 = torch.from_numpy(W.random_scalars(curve, n, seed=982)).to(dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(user):
            out, fl = eng.scalarmul_var_t(curve, k2, pts, stream=user.cuda_stream)
        eng.prepare("ed25519", ct=True)  # a table build for ANOTHER curve while the batch is in flight
        torch.cuda.synchronize()
        sample = np.arange(0, n, n // 64)
        want = oracle.var(curve, k2.cpu().numpy().reshape(n, 32)[sample].tobytes(),
                          pts.cpu().numpy().reshape(n, 64)[sample].tobytes(), threads=16)
        assert out.cpu().numpy().reshape(n, 64)[sample].tobytes() == want[0]
