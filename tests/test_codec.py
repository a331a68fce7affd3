"""Point wire formats: eccx_point_compress / eccx_point_decompress against the
oracle's restatement of the reference's codecs and against the reference's own encodings.

Pinned by reference fixtures: the zcash encodings of k*G (src/curve/bls12_381/g1.rs:605-692,
tests/golden/bls_g1.json) and the RFC 8032 public keys (src/protocol/ed25519.rs:271-292,
tests/golden/rfc8032.json).  The sec2 curves have no byte-level fixture in the reference
(PointAffine::compress returns (x, Sign), curve_macros.rs:211-213): the SEC1 prefix byte is
"parity unpinned" at the byte level; the sign rule itself (Sign::Negative = y odd,
field_macros.rs:557-565) is what the oracle and the kernels both implement, and the round
trips below pin that x and the sign recover the NIST kG points (tests/golden/nist_kg.json)."""
import random

import pytest

from eccoxide_amd import workload as W
from oracle import ecc_ref as R
from tests.oracle_lib import golden

CURVES = ["p256r1", "p384r1", "p521r1", "bls12_381_g1", "ed25519"]


def _points(oracle, curve, n, seed):
    """n multiples of the generator as x||y records, the first one replaced by a point with an
    infinity flag (Weierstrass)."""
    xy, flags, _ = oracle.base(curve, W.random_scalars(curve, n, seed=seed).tobytes())
    assert not any(flags)
    return xy


# ---- CPU: the oracle against the reference's encodings -------------------------------------
def test_oracle_codec_matches_reference_encodings():
    c = R.BLS12_381_G1
    for e in golden("bls_g1.json")["compressed"]:
        P = R.affine_mul(c, e["k"], (c.gx, c.gy))
        assert R.ref_point_compress("bls12_381_g1", P).hex() == e["bytes"]
        assert R.ref_point_decompress("bls12_381_g1", bytes.fromhex(e["bytes"]), True) == (P, 0)
    for v in golden("rfc8032.json"):
        k = R.ed25519_secret_scalar(bytes.fromhex(v["seed"]))
        P = R.ed_affine_mul(R.ED25519, k, (R.ED25519.gx, R.ED25519.gy))
        assert R.ref_point_compress("ed25519", P).hex() == v["public"]
        assert R.ref_point_decompress("ed25519", bytes.fromhex(v["public"])) == (P, 0)


def test_oracle_codec_round_trips_nist_points():
    for curve, vecs in golden("nist_kg.json").items():
        c = R.WEIERSTRASS[curve]
        for v in vecs[:12]:
            P = (int(v["x"], 16), int(v["y"], 16))
            enc = R.ref_point_compress(curve, P)
            assert len(enc) == c.fb + 1 and enc[0] == 2 + (P[1] & 1)
            assert R.ref_point_decompress(curve, enc) == (P, 0)
            other = bytes([enc[0] ^ 1]) + enc[1:]  # the other sign gives the negated point (curve_macros.rs:433-456)
            assert R.ref_point_decompress(curve, other) == ((P[0], c.p - P[1]), 0)


def test_oracle_codec_rejections():
    """The rejections the reference tests (serialize.rs:640-726, ed25519.rs decode_point)."""
    c = R.BLS12_381_G1
    g = bytes.fromhex(golden("bls_g1.json")["compressed"][0]["bytes"])
    assert R.ref_point_decompress("bls12_381_g1", bytes([g[0] & 0x7F]) + g[1:])[1] == 2      # compression bit clear
    assert R.ref_point_decompress("bls12_381_g1", bytes([0xC0]) + bytes(47)) == (None, 1)     # infinity
    assert R.ref_point_decompress("bls12_381_g1", bytes([0xE0]) + bytes(47))[1] == 2         # infinity with sort
    assert R.ref_point_decompress("bls12_381_g1", bytes([0xC0]) + bytes(46) + b"\x01")[1] == 2
    assert R.ref_point_decompress("bls12_381_g1", bytes([0x80 | (c.p >> 376)]) + (c.p & ((1 << 376) - 1)).to_bytes(47, "big"))[1] == 2
    p = R.ED25519.p
    assert R.ref_point_decompress("ed25519", p.to_bytes(32, "little"))[1] == 2                # y = p: not canonical
    one_signed = bytearray((1).to_bytes(32, "little")); one_signed[31] |= 0x80
    assert R.ref_point_decompress("ed25519", bytes(one_signed))[1] == 2                      # x = 0 with the sign bit
    assert R.ref_point_decompress("ed25519", (1).to_bytes(32, "little")) == ((0, 1), 0)


def test_oracle_uncompressed_flavour_matches_reference_encodings():
    c = R.BLS12_381_G1
    for e in golden("bls_g1.json")["uncompressed"]:
        P = R.affine_mul(c, e["k"], (c.gx, c.gy))
        assert R.ref_g1_to_uncompressed(P).hex() == e["bytes"]
        assert R.ref_g1_from_uncompressed(bytes.fromhex(e["bytes"]), True) == (P, 0)
    assert R.ref_g1_from_uncompressed(bytes([0x40]) + bytes(95)) == (None, 1)
    assert R.ref_g1_from_uncompressed(bytes([0x40]) + bytes(94) + b"\x01")[1] == 2


# ---- GPU ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("curve", CURVES)
def test_compress_matches_oracle_and_round_trips(engine, oracle, curve):
    n = 777
    fb = oracle.fb(curve)
    xy = bytearray(_points(oracle, curve, n, seed=901))
    inf = bytearray(n)
    if curve != "ed25519":
        for i in (0, 300, n - 1):  # infinity records: coordinates are ignored
            inf[i] = 1
    enc = engine.point_compress(curve, bytes(xy), bytes(inf) if curve != "ed25519" else None)
    assert len(enc) == n * engine.compressed_bytes(curve) == n * R.compressed_bytes(curve)
    assert enc == R.point_compress_bytes(curve, bytes(xy), bytes(inf))
    out, flags = engine.point_decompress(curve, enc)
    for i in range(n):
        if inf[i]:
            assert flags[i] == 1 and out[2 * fb * i:2 * fb * (i + 1)] == bytes(2 * fb)
        else:
            assert flags[i] == 0 and out[2 * fb * i:2 * fb * (i + 1)] == bytes(xy[2 * fb * i:2 * fb * (i + 1)]), i


@pytest.mark.gpu
@pytest.mark.parametrize("curve", CURVES)
def test_decompress_random_and_malformed_encodings(engine, oracle, curve):
    """Differential test on encodings of which about half are rejected: random coordinates (no
    point / a point with either sign), every prefix and flag pattern, non-canonical coordinates,
    edge values."""
    rnd = random.Random(4242 + len(curve))
    eb = R.compressed_bytes(curve)
    p = R.CURVES[curve].p
    fb = R.CURVES[curve].fb
    recs = []
    if curve == "ed25519":
        for _ in range(1500):
            recs.append(bytes(rnd.getrandbits(8) for _ in range(32)))
        for y in (0, 1, 2, p - 1, p - 2, p, p + 1, p + 18, (1 << 255) - 1, 4 * pow(5, -1, p) % p):
            for s in (0, 1):
                recs.append((y | (s << 255)).to_bytes(32, "little"))
    else:
        zc = curve == "bls12_381_g1"
        prefixes = [0x80, 0xA0] if zc else [2, 3]
        for _ in range(700):
            x = rnd.randrange(p)
            pre = rnd.choice(prefixes)
            recs.append(bytes([x >> (8 * (fb - 1)) | pre]) + (x & ((1 << (8 * (fb - 1))) - 1)).to_bytes(fb - 1, "big") if zc
                        else bytes([pre]) + x.to_bytes(fb, "big"))
        xs = [0, 1, 2, 3, 5, p - 1, p - 2, p, p + 1, (1 << (8 * fb - (3 if zc else 0))) - 1, R.CURVES[curve].gx]
        pres = [0x00, 0x20, 0x40, 0x60, 0x80, 0xA0, 0xC0, 0xE0] if zc else [0, 1, 2, 3, 4, 5, 6, 7, 0x80, 0xFF]
        for x in xs:
            for pre in pres:
                if zc:
                    xm = x & ((1 << (8 * fb - 3)) - 1)
                    recs.append(bytes([xm >> (8 * (fb - 1)) | pre]) + (xm & ((1 << (8 * (fb - 1))) - 1)).to_bytes(fb - 1, "big"))
                else:
                    recs.append(bytes([pre]) + (x & ((1 << (8 * fb)) - 1)).to_bytes(fb, "big"))
    enc = b"".join(recs)
    assert len(enc) == len(recs) * eb
    out, flags = engine.point_decompress(curve, enc)
    want_out, want_flags = R.point_decompress_bytes(curve, enc)
    assert flags == want_flags
    assert out == want_out
    counts = [want_flags.count(bytes([k])) for k in range(3)]
    assert counts[0] > len(recs) // 4 and counts[2] > len(recs) // 4, counts  # both branches are exercised


@pytest.mark.gpu
def test_reference_encodings_through_the_engine(engine):
    kats = golden("bls_g1.json")
    enc = b"".join(bytes.fromhex(e["bytes"]) for e in kats["compressed"])
    out, flags = engine.point_decompress("bls12_381_g1", enc, check_subgroup=True)
    assert flags == bytes(len(kats["compressed"]))
    assert engine.point_compress("bls12_381_g1", out) == enc
    # the uncompressed encodings the reference holds for two of them are x||y as they stand
    by_k = {e["k"]: i for i, e in enumerate(kats["compressed"])}
    for e in kats["uncompressed"]:
        i = by_k[e["k"]]
        assert out[96 * i:96 * i + 96] == bytes.fromhex(e["bytes"])
    pubs = b"".join(bytes.fromhex(v["public"]) for v in golden("rfc8032.json"))
    xy, flags = engine.point_decompress("ed25519", pubs)
    assert flags == bytes(len(pubs) // 32)
    assert engine.point_compress("ed25519", xy) == pubs
    # keygen shape on the device: mul_base of the clamped secret scalar, then the encoding
    ks = b"".join(R.ed25519_secret_scalar(bytes.fromhex(v["seed"])).to_bytes(32, "big") for v in golden("rfc8032.json"))
    pts, fl = engine.scalarmul_base("ed25519", ks)
    assert engine.point_compress("ed25519", pts) == pubs


@pytest.mark.gpu
def test_bls_subgroup_check(engine, oracle):
    """from_compressed against from_compressed_oncurve_only (serialize.rs:299-335,728-747): curve
    points outside G1 decode without the check and are rejected with it."""
    curve = "bls12_381_g1"
    c = R.BLS12_381_G1
    inside = R.point_compress_bytes(curve, _points(oracle, curve, 40, seed=77))
    outside = []
    x = 1
    while len(outside) < 40:  # small x on the curve: such points are not in G1 (cofactor ~2^126)
        x += 1
        P = R.ref_w_decompress_xy(c, x, bool(x & 1))
        if P is not None and R.affine_mul(c, c.n, P) is not None:
            outside.append(R.ref_point_compress(curve, P))
    enc = inside + b"".join(outside) + bytes([0xC0]) + bytes(47)
    n = len(enc) // 48
    out0, fl0 = engine.point_decompress(curve, enc)
    assert fl0 == bytes(80) + b"\x01"
    assert (out0, fl0) == R.point_decompress_bytes(curve, enc)
    out1, fl1 = engine.point_decompress(curve, enc, check_subgroup=True)
    assert fl1 == bytes(40) + b"\x02" * 40 + b"\x01"
    assert (out1, fl1) == R.point_decompress_bytes(curve, enc, check_subgroup=True)
    assert out1[:40 * 96] == out0[:40 * 96] and out1[40 * 96:] == bytes((n - 40) * 96)
    with pytest.raises(Exception):
        engine.point_decompress("ed25519", bytes(32), check_subgroup=True)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["p256r1", "ed25519"])
def test_codec_chain_on_device_tensors(engine, oracle, curve):
    """decompress -> scalarmul -> compress without leaving the GPU, on the caller's stream."""
    import torch

    n = 900
    xy = _points(oracle, curve, n, seed=31)
    k = W.random_scalars(curve, n, seed=32).tobytes()
    enc = R.point_compress_bytes(curve, xy)
    want_xy, want_fl, _ = oracle.var(curve, k, xy)
    want = R.point_compress_bytes(curve, want_xy, want_fl)
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        pts, fl = engine.point_decompress_t(curve, t(enc).reshape(n, -1), stream=stream.cuda_stream)
        res, rfl = engine.scalarmul_var_t(curve, t(k).reshape(n, -1), pts, stream=stream.cuda_stream)
        got = engine.point_compress_t(curve, res, rfl if curve != "ed25519" else None, stream=stream.cuda_stream)
    stream.synchronize()
    assert fl.cpu().numpy().tobytes() == bytes(n)
    assert got.cpu().numpy().tobytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["p256r1", "bls12_381_g1", "ed25519"])
def test_codec_round_trip_at_full_batch(engine, curve):
    """2^20 units: mul_base -> compress -> decompress gives the points back (size-independent
    property; the oracle covers the small batches above)."""
    import torch

    n = 1 << 20
    dev = torch.device("cuda", 0)
    k = torch.from_numpy(W.random_scalars(curve, n, seed=5)).to(dev)
    pts, fl = engine.scalarmul_base_t(curve, k)
    enc = engine.point_compress_t(curve, pts, fl if curve != "ed25519" else None)
    back, bfl = engine.point_decompress_t(curve, enc)
    torch.cuda.synchronize()
    assert int(fl.sum()) == 0 and int(bfl.sum()) == 0
    assert torch.equal(back, pts)


@pytest.mark.gpu
def test_zcash_uncompressed_flavour(engine, oracle):
    """to_uncompressed / from_uncompressed[_oncurve_only] (serialize.rs:269-279,353-383) against
    the oracle: reference encodings, round trip with an infinity record, every flag pattern,
    non-canonical and off-curve coordinates, a curve point outside G1."""
    curve = "bls12_381_g1"
    c = R.BLS12_381_G1
    kats = golden("bls_g1.json")["uncompressed"]
    enc = b"".join(bytes.fromhex(e["bytes"]) for e in kats)
    out, flags = engine.point_decompress(curve, enc, uncompressed=True, check_subgroup=True)
    assert flags == bytes(len(kats)) and out == enc
    n = 600
    xy = _points(oracle, curve, n, seed=17)
    inf = bytearray(n); inf[4] = 1
    raw = engine.point_compress(curve, xy, bytes(inf), uncompressed=True)
    want = b"".join(R.ref_g1_to_uncompressed(None if inf[i] else (int.from_bytes(xy[96 * i:96 * i + 48], "big"),
                                                                    int.from_bytes(xy[96 * i + 48:96 * i + 96], "big"))) for i in range(n))
    assert raw == want
    rnd = random.Random(99)
    recs = [raw[96 * i:96 * i + 96] for i in range(40)]
    g = raw[:96]
    for pre in (0x20, 0x40, 0x60, 0x80, 0xA0, 0xC0, 0xE0):
        recs.append(bytes([g[0] | pre]) + g[1:])
        recs.append(bytes([pre]) + bytes(95))
    recs.append(c.p.to_bytes(48, "big") + g[48:])                       # x = p
    recs.append(g[:48] + c.p.to_bytes(48, "big"))                       # y = p
    recs.append(g[:48] + (c.p - int.from_bytes(g[48:], "big")).to_bytes(48, "big"))   # -G
    recs.append(g[:95] + bytes([g[95] ^ 1]))                            # off the curve
    x = 1
    while True:                                                          # a curve point outside G1
        x += 1
        P = R.ref_w_decompress_xy(c, x, False)
        if P is not None:
            recs.append(R.ref_g1_to_uncompressed(P))
            break
    for _ in range(100):
        recs.append(bytes(rnd.getrandbits(8) for _ in range(96)))
    blob = b"".join(recs)
    for chk in (False, True):
        out, flags = engine.point_decompress(curve, blob, uncompressed=True, check_subgroup=chk)
        for i, r in enumerate(recs):
            P, st = R.ref_g1_from_uncompressed(r, chk)
            assert flags[i] == st, (i, chk)
            assert out[96 * i:96 * i + 96] == (bytes(96) if P is None else P[0].to_bytes(48, "big") + P[1].to_bytes(48, "big")), (i, chk)
    with pytest.raises(Exception):
        engine.point_compress("p256r1", bytes(64), uncompressed=True)
    # device-tensor forms
    import torch

    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    raw_t = engine.point_compress_t(curve, t(xy).reshape(n, -1), t(bytes(inf)), uncompressed=True)
    back_t, fl_t = engine.point_decompress_t(curve, raw_t, uncompressed=True)
    torch.cuda.synchronize()
    assert raw_t.cpu().numpy().tobytes() == raw
    assert (back_t.cpu().numpy().tobytes(), fl_t.cpu().numpy().tobytes()) == engine.point_decompress(curve, raw, uncompressed=True)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["p256r1", "p521r1", "ed25519"])
def test_codec_on_misaligned_device_buffers(engine, oracle, curve):
    """Device buffers at odd byte offsets (tensor views): the 16-byte staging of the compress kernel
    falls back to byte accesses; a ragged last workgroup on top."""
    import torch

    n = 256 * 3 + 41
    fb, eb = oracle.fb(curve), R.compressed_bytes(curve)
    xy = _points(oracle, curve, n, seed=23)
    want = R.point_compress_bytes(curve, xy)
    dev = torch.device("cuda", 0)
    for off_in, off_out in ((1, 0), (0, 3), (5, 7), (16, 32)):
        src = torch.zeros(n * 2 * fb + 64, dtype=torch.uint8, device=dev)
        src[off_in:off_in + n * 2 * fb] = torch.frombuffer(bytearray(xy), dtype=torch.uint8).to(dev)
        dst = torch.full((n * eb + 64,), 0xAA, dtype=torch.uint8, device=dev)
        engine.point_compress_t(curve, src[off_in:off_in + n * 2 * fb], None, dst[off_out:off_out + n * eb])
        torch.cuda.synchronize()
        assert dst[off_out:off_out + n * eb].cpu().numpy().tobytes() == want
        assert bool((dst[:off_out] == 0xAA).all()) and bool((dst[off_out + n * eb:] == 0xAA).all())   # nothing written outside
        back, fl = engine.point_decompress_t(curve, dst[off_out:off_out + n * eb])
        torch.cuda.synchronize()
        assert back.cpu().numpy().tobytes() == xy and int(fl.sum()) == 0
