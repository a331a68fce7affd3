"""Pins the CPU oracle (oracle/ecc_ref.py and oracle/eccx_oracle.c) against every golden
vector the reference's own tests hold for the scalar-multiplication path,
extracted as data by tools/extract_fixtures.py into tests/golden/.  CPU only.
"""
import hashlib
import random

import pytest

from oracle import ecc_ref as R
from tests import oracle_lib
from tests.oracle_lib import golden

WEI = ["p256r1", "p384r1", "p521r1", "bls12_381_g1"]
ALL = WEI + ["ed25519"]


def be(x: int, n: int) -> bytes:
    return x.to_bytes(n, "big")


def gen_bytes(c) -> bytes:
    return be(c.gx, c.fb) + be(c.gy, c.fb)


# ---- constants ---------------------------------------------------------------
def test_params_match_reference():
    """src/params/sec2.rs, src/params/bls12_381.rs, src/curve/curve25519.rs:391-417"""
    P = golden("params.json")
    for name in WEI:
        c, f = R.CURVES[name], P[name]
        assert int(f["p"], 16) == c.p and int(f["order"], 16) == c.n
        assert int(f["a"], 16) == c.a and int(f["b"], 16) == c.b and int(f["b3"], 16) == c.b3
        assert int(f["gx"], 16) == c.gx and int(f["gy"], 16) == c.gy
        assert R.on_curve(c, (c.gx, c.gy))
    f, c = P["ed25519"], R.ED25519
    assert int(f["d"], 16) == c.d and int(f["d2"], 16) == c.d2
    assert int(f["gx"], 16) == c.gx and int(f["gy"], 16) == c.gy
    assert int(f["gt"], 16) == c.gx * c.gy % c.p


def test_engine_side_constants_match_reference():
    """The product's own copies (workload orders, generated Montgomery constants)."""
    import re

    from eccoxide_amd import workload

    P = golden("params.json")
    for name in WEI:
        assert workload.ORDERS[name] == P[name]["order"]
    assert workload.order("ed25519") == R.ED25519.n
    txt = open(oracle_lib.ROOT + "/eccoxide_amd/csrc/curve_consts.inc").read()
    for sname, name in (("P256", "p256r1"), ("P384", "p384r1"), ("P521", "p521r1"), ("BLS12_381", "bls12_381_g1")):
        body = txt[txt.index("struct %s {" % sname):]
        body = body[: body.index("\n};")]

        def limbs(field):
            m = re.search(r"uint32_t %s\[\d+\] = \{([^}]*)\}" % field, body)
            vals = [int(v.strip().rstrip("u"), 16) for v in m.group(1).split(",")]
            return sum(v << (32 * i) for i, v in enumerate(vals)), len(vals)

        c = R.CURVES[name]
        p, L = limbs("P")
        assert p == c.p
        mersenne = int(re.search(r"MERSENNE = (\d+);", body).group(1))
        Rm = 1 if mersenne else 1 << (32 * L)   # Mersenne fields are kept in plain form
        assert (mersenne != 0) == (name == "p521r1") and (not mersenne or p == 2**mersenne - 1)
        assert limbs("ONE")[0] == Rm % p and limbs("R2")[0] == Rm * Rm % p
        assert limbs("B")[0] == c.b * Rm % p and limbs("B3")[0] == c.b3 * Rm % p
        assert limbs("GX")[0] == c.gx * Rm % p and limbs("GY")[0] == c.gy * Rm % p
        assert limbs("PM2")[0] == p - 2


def test_unsaturated_field_constants_match_reference():
    """The digit tables of the unsaturated field layer (ufe.hpp) against the curve parameters:
    modulus, p + 1 and its Solinas / sparse terms, the 4p subtraction bias (every limb at least
    the largest tight digit), Montgomery constants, generator."""
    import re

    txt = open(oracle_lib.ROOT + "/eccoxide_amd/csrc/curve_consts.inc").read()
    cases = (("P256U", R.CURVES["p256r1"]), ("P384U", R.CURVES["p384r1"]), ("P521U", R.CURVES["p521r1"]),
             ("BLS12_381U", R.CURVES["bls12_381_g1"]), ("ED25519U", R.ED25519))
    for sname, c in cases:
        body = txt[txt.index("struct %s {" % sname):]
        body = body[: body.index("\n};")]
        num = lambda f: int(re.search(r"%s = (-?\w+?)u?;" % f, body).group(1), 0)
        N, B, kind = num("N"), num("B"), num("KIND")

        def arr(field):
            m = re.search(r"(?:uint32_t|int) %s\[\d+\] = \{([^}]*)\}" % field, body)
            return [int(v.strip().rstrip("u"), 0) for v in m.group(1).split(",")]

        val = lambda field: sum(v << (B * i) for i, v in enumerate(arr(field)))
        p = c.p
        mont = kind in (0, 1)
        Rm = (1 << (B * N)) if mont else 1
        assert B * N >= p.bit_length() and num("PBITS") == p.bit_length()
        assert val("P") == p and val("PP1") == p + 1 and val("P2") == 2 * p
        assert all(d < (1 << B) for f in ("P", "PP1", "ONE", "R2", "GX", "GY") for d in arr(f)[:-1])
        assert val("ONE") == Rm % p and val("R2") == Rm * Rm % p
        assert val("GX") == c.gx * Rm % p and val("GY") == c.gy * Rm % p
        bias = arr("BIAS")
        assert val("BIAS") == 4 * p and all(d >= (1 << B) - 1 for d in bias[:-1]) and all(d < (1 << (B + 1)) for d in bias)
        assert num("TOPSHIFT") == p.bit_length() - B * (N - 1)
        assert num("N0B") == (-pow(p, -1, 1 << B)) % (1 << B)
        if kind == 0:
            assert p % (1 << B) == (1 << B) - 1          # the Montgomery factor is the low limb itself
        if kind in (2, 3):
            assert num("FOLD") == (1 << (B * N)) % p
        if num("SOL_N"):
            terms = sum(sg << (B * l + sh) for l, sh, sg in zip(arr("SOL_LIMB"), arr("SOL_SHIFT"), arr("SOL_SIGN")))
            assert (1 << p.bit_length()) + terms == p + 1
        if num("SPARSE_N"):
            terms = sum(sg << (B * o + sh) for o, sh, sg in zip(arr("SPARSE_OFF"), arr("SPARSE_SHIFT"), arr("SPARSE_SIGN")))
            assert terms == p + 1
        if num("QMUL"):
            ptop = p >> (B * (N - 1))
            assert num("QMUL") == (1 << 32) // (ptop + 1)
    d2 = txt[txt.index("struct ED25519U {"):]
    d2 = d2[: d2.index("\n};")]
    vals = [int(v.strip().rstrip("u"), 16) for v in re.search(r"uint32_t D2\[9\] = \{([^}]*)\}", d2).group(1).split(",")]
    assert sum(v << (29 * i) for i, v in enumerate(vals)) == 2 * R.ED25519.d % R.ED25519.p


# ---- NIST kG (src/tests/kats.rs:3-40 over kats_data.rs) ----------------------------
@pytest.mark.parametrize("name", ["p256r1", "p384r1", "p521r1"])
def test_nist_kg_python_oracle(name):
    c = R.CURVES[name]
    for kv in golden("nist_kg.json")[name]:
        k = bytes.fromhex(kv["k"]).rjust(c.sb, b"\0")
        want = (int(kv["x"], 16), int(kv["y"], 16))
        q = R.ref_scalar_mul_fixed_window(c, (c.gx, c.gy, 1), k)
        assert R.ref_to_affine(c, q) == want                       # the reference's algorithm
        assert R.affine_mul(c, int(kv["k"], 16), (c.gx, c.gy)) == want  # textbook arithmetic
        assert R.ref_to_affine(c, R.ref_mul_base_table(c, R.comb_table(name), k)) == want  # comb path


@pytest.mark.parametrize("name", ["p256r1", "p384r1", "p521r1"])
def test_nist_kg_c_oracle(oracle, name):
    c = R.CURVES[name]
    kats = golden("nist_kg.json")[name]
    ks = b"".join(bytes.fromhex(kv["k"]).rjust(c.sb, b"\0") for kv in kats)
    want = b"".join(bytes.fromhex(kv["x"]).rjust(c.fb, b"\0") + bytes.fromhex(kv["y"]).rjust(c.fb, b"\0") for kv in kats)
    out, inf, _ = oracle.var(name, ks, gen_bytes(c) * len(kats))
    assert out == want and inf == bytes(len(kats))
    out, inf, _ = oracle.base(name, ks)
    assert out == want and inf == bytes(len(kats))


# ---- RFC 6979 (src/protocol/ecdsa.rs:700-707, :808-915) ---------------------------
@pytest.mark.parametrize("name", ["p256r1", "p384r1", "p521r1"])
def test_rfc6979_public_keys_and_r(oracle, name):
    c = R.CURVES[name]
    v = golden("rfc6979.json")[name]
    secret = bytes.fromhex(v["secret"].rjust(2 * c.sb, "0"))
    out, inf, _ = oracle.base(name, secret)
    assert out == bytes.fromhex(v["ux"].rjust(2 * c.fb, "0")) + bytes.fromhex(v["uy"].rjust(2 * c.fb, "0"))
    assert inf == b"\0"
    # signing nonces: r = x(k*G) mod n   (ecdsa.rs:174)
    for kat in v["sign_kats"]:
        k = bytes.fromhex(kat["k"].rjust(2 * c.sb, "0"))
        o, _, _ = oracle.base(name, k)
        assert int.from_bytes(o[: c.fb], "big") % c.n == int(kat["r"], 16)
        o2, _, _ = oracle.var(name, k, gen_bytes(c))
        assert o2 == o
        q = R.ref_to_affine(c, R.ref_mul_base_table(c, R.comb_table(name), k))
        assert q[0] % c.n == int(kat["r"], 16)


# ---- RFC 8032 (src/protocol/ed25519.rs:271-315) -------------------------------------
def test_rfc8032_public_keys(oracle):
    c = R.ED25519
    for v in golden("rfc8032.json"):
        a = R.ed25519_secret_scalar(bytes.fromhex(v["seed"]))
        k = a.to_bytes(32, "big")
        q = R.ref_ed_mul_base(c, R.comb_table("ed25519"), k)
        assert R.ed_encode_point(c, R.ref_ed_to_affine(c, q)).hex() == v["public"]
        out, _, _ = oracle.base("ed25519", k)
        x, y = int.from_bytes(out[:32], "little"), int.from_bytes(out[32:], "little")
        assert R.ed_encode_point(c, (x, y)).hex() == v["public"]
        gen = c.gx.to_bytes(32, "little") + c.gy.to_bytes(32, "little")
        out2, _, _ = oracle.var("ed25519", k, gen)   # scale_bytes == mul_base (curve25519.rs:1373-1387)
        assert out2 == out


# ---- BLS12-381 G1 (src/curve/bls12_381/g1.rs:605-692) --------------------------------
def test_bls_g1_kats(oracle):
    c = R.BLS12_381_G1
    v = golden("bls_g1.json")
    for e in v["uncompressed"]:
        k = e["k"].to_bytes(32, "big")
        out, inf, _ = oracle.var("bls12_381_g1", k, gen_bytes(c))
        assert out.hex() == e["bytes"] and inf == b"\0"   # flag bits are clear for finite uncompressed points
        assert oracle.base("bls12_381_g1", k)[0] == out
    for e in v["compressed"]:
        k = e["k"].to_bytes(32, "big")
        out, _, _ = oracle.base("bls12_381_g1", k)
        x, y = int.from_bytes(out[:48], "big"), int.from_bytes(out[48:], "big")
        enc = bytearray(bytes.fromhex(e["bytes"]))
        sort_flag = bool(enc[0] & 0x20)
        assert enc[0] & 0x80 and not enc[0] & 0x40      # compressed, not infinity
        enc[0] &= 0x1F
        assert int.from_bytes(enc, "big") == x
        assert sort_flag == (y > (c.p - 1) // 2)
        assert R.affine_mul(c, e["k"], (c.gx, c.gy)) == (x, y)


# ---- comb tables (src/params/comb/*.rs) ---------------------------------------------
@pytest.mark.parametrize("name", ALL)
def test_comb_table_matches_reference(oracle, name):
    cs = golden("comb_samples.json")[name]
    c = R.CURVES[name]
    assert cs["windows"] == c.nw and cs["field_bytes"] == c.fb
    raw = oracle.comb_table(name)
    assert hashlib.sha256(raw).hexdigest() == cs["sha256_xy_concat"]
    tab = R.comb_table(name)
    h = hashlib.sha256()
    for row in tab:
        for x, y in row:
            h.update(x.to_bytes(c.fb, cs["byte_order"]))
            h.update(y.to_bytes(c.fb, cs["byte_order"]))
    assert h.hexdigest() == cs["sha256_xy_concat"]
    for w, entries in cs["samples"].items():
        for j, (xs, ys) in enumerate(entries):
            off = (int(w) * 15 + j) * 2 * c.fb
            assert raw[off: off + c.fb].hex() == xs and raw[off + c.fb: off + 2 * c.fb].hex() == ys


def test_ed25519_wnaf_base_table():
    """WNAF_BASE_TABLE[i] = (2i+1)*B as (y-x, y+x, 2dxy) (comb/curve25519.rs:1100-1107)"""
    c = R.ED25519
    v = golden("comb_samples.json")["ed25519_wnaf_base"]
    h = hashlib.sha256()
    for i in range(64):
        x, y = R.ed_affine_mul(c, 2 * i + 1, (c.gx, c.gy))
        row = [((y - x) % c.p), ((y + x) % c.p), (c.d2 * x * y % c.p)]
        enc = [r.to_bytes(32, "little") for r in row]
        for e in enc:
            h.update(e)
        if i < 8:
            assert [e.hex() for e in enc] == v["first8"][i]
    assert h.hexdigest() == v["sha256_concat"]


# ---- C restatement == Python restatement, incl. un-normalised coordinates ------------
@pytest.mark.parametrize("name", WEI)
def test_c_oracle_matches_python_oracle(oracle, name):
    c = R.CURVES[name]
    rng = random.Random(0xC0FFEE + len(name))
    special = [0, 1, 2, 15, 16, c.n - 1, c.n, c.n + 1, (1 << (8 * c.sb)) - 1 if name != "p521r1" else (1 << 521) - 1]
    ks = [rng.randrange(1, c.n) for _ in range(6)] + special
    pts = [R.affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in ks]
    kb = b"".join(be(k, c.sb) for k in ks)
    pb = b"".join(be(x, c.fb) + be(y, c.fb) for x, y in pts)
    out, inf, proj = oracle.var(name, kb, pb)
    eo, ei = R.scalarmul_var_bytes(name, kb, pb)
    assert out == eo and inf == ei
    for i, (k, P) in enumerate(zip(ks, pts)):
        q = R.ref_scalar_mul_fixed_window(c, (P[0], P[1], 1), be(k, c.sb))
        assert proj[i * 3 * c.fb:(i + 1) * 3 * c.fb] == b"".join(be(v, c.fb) for v in q)
        a = R.affine_mul(c, k, P)   # ground truth
        if a is None:
            assert inf[i] == 1 and out[i * 2 * c.fb:(i + 1) * 2 * c.fb] == bytes(2 * c.fb)
        else:
            assert inf[i] == 0 and out[i * 2 * c.fb:(i + 1) * 2 * c.fb] == be(a[0], c.fb) + be(a[1], c.fb)
    out, inf, proj = oracle.base(name, kb)
    eo, ei = R.scalarmul_base_bytes(name, kb)
    assert out == eo and inf == ei
    for i, k in enumerate(ks):
        q = R.ref_mul_base_table(c, R.comb_table(name), be(k, c.sb))
        assert proj[i * 3 * c.fb:(i + 1) * 3 * c.fb] == b"".join(be(v, c.fb) for v in q)


def test_c_oracle_matches_python_oracle_ed25519(oracle):
    c = R.ED25519
    rng = random.Random(25519)
    ks = [rng.randrange(1, c.n) for _ in range(6)] + [0, 1, c.n - 1, c.n, (1 << 256) - 1]
    kb = b"".join(be(k, 32) for k in ks)
    out, inf, proj = oracle.base("ed25519", kb)
    eo, ei = R.scalarmul_base_bytes("ed25519", kb)
    assert out == eo and inf == ei
    for i, k in enumerate(ks):
        want = R.ed_affine_mul(c, k, (c.gx, c.gy))
        assert out[i * 64: i * 64 + 32] == want[0].to_bytes(32, "little")
        assert out[i * 64 + 32:(i + 1) * 64] == want[1].to_bytes(32, "little")
        q = R.ref_ed_mul_base(c, R.comb_table("ed25519"), be(k, 32))
        assert proj[i * 128:(i + 1) * 128] == b"".join(v.to_bytes(32, "little") for v in q)
    pts = [R.ed_affine_mul(c, rng.randrange(1, c.n), (c.gx, c.gy)) for _ in ks]
    pb = b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for x, y in pts)
    out, inf, proj = oracle.var("ed25519", kb, pb)
    for i, (k, P) in enumerate(zip(ks, pts)):
        want = R.ed_affine_mul(c, k, P)
        assert out[i * 64:(i + 1) * 64] == want[0].to_bytes(32, "little") + want[1].to_bytes(32, "little")
        q = R.ref_ed_scale_bytes(c, R.ref_ed_from_affine(c, *P), be(k, 32))
        assert proj[i * 128:(i + 1) * 128] == b"".join(v.to_bytes(32, "little") for v in q)
        assert inf[i] == (1 if want == (0, 1) else 0)


# ---- reference properties re-run on the oracle (src/tests/completeness.rs:21-117) ------
@pytest.mark.parametrize("name", WEI)
def test_completeness_properties(name):
    c = R.CURVES[name]
    rng = random.Random(99)
    G = (c.gx, c.gy, 1)
    P = R.ref_scalar_mul_fixed_window(c, G, be(rng.randrange(1, c.n), c.sb))
    aff = lambda q: R.ref_to_affine(c, q)
    assert aff(R.ref_add(c, P, P)) == aff(R.ref_double(c, P))                       # P + P
    negP = (P[0], -P[1] % c.p, P[2])
    assert aff(R.ref_add(c, P, negP)) is None                                       # P + (-P)
    assert aff(R.ref_add(c, P, R.INFINITY)) == aff(P) == aff(R.ref_add(c, R.INFINITY, P))  # P + O
    a, b = rng.randrange(1, c.n), rng.randrange(1, c.n)                             # homomorphism
    lhs = R.ref_add(c, R.ref_scalar_mul_fixed_window(c, G, be(a, c.sb)), R.ref_scalar_mul_fixed_window(c, G, be(b, c.sb)))
    assert aff(lhs) == aff(R.ref_scalar_mul_fixed_window(c, G, be((a + b) % c.n, c.sb)))
    k = be(rng.randrange(1, c.n), c.sb)                                            # mul_base == generic
    assert aff(R.ref_mul_base_table(c, R.comb_table(name), k)) == aff(R.ref_scalar_mul_fixed_window(c, G, k))


def test_g1_endomorphism_subgroup_test_agrees_with_the_definition():
    """The oracle's restatement of PointAffine::is_in_subgroup (sigma(P) == [-x^2]P, g1.rs:90-109) with
    the reference's beta (tests/golden/params.json, from src/params/bls12_381.rs:100) against the
    definition [r]P == infinity, on G1 points, arbitrary curve points and cofactor-subgroup points."""
    c = R.BLS12_381_G1
    assert int(golden("params.json")["bls12_381_g1"]["beta"], 16) == R.BLS_BETA
    assert pow(R.BLS_BETA, 3, c.p) == 1 and R.BLS_BETA != 1
    G = (c.gx, c.gy)
    pts = [R.affine_mul(c, k, G) for k in (1, 2, 7, c.n - 1, 0x1234567890ABCDEF)]
    x = 1
    while len(pts) < 15:
        x += 1
        P = R.ref_w_decompress_xy(c, x, bool(x & 1))
        if P is not None:
            pts.append(P)
            pts.append(R.affine_mul(c, c.n, P))
    for P in pts:
        assert R.ref_g1_is_in_subgroup(P) == (R.affine_mul(c, c.n, P) is None)
    assert R.ref_g1_is_in_subgroup(None)
    # the G1 encodings the reference's tests hold decode under the check
    for e in golden("bls_g1.json")["compressed"]:
        P, st = R.ref_point_decompress("bls12_381_g1", bytes.fromhex(e["bytes"]), check_subgroup=True)
        assert st == R.CODEC_OK and P is not None
