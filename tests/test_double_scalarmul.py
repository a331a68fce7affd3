"""Double-scalar "verify shape" u1*G + u2*Q: ECDSA verification of the
RFC 6979 signatures the reference's tests hold (src/protocol/ecdsa.rs:808-915, verify :197-233),
Ed25519's [s]B - [k]A, and random batches against the big-int oracle."""
import hashlib
import random

import pytest

from eccoxide_amd import workload as W
from oracle import ecc_ref as R
from tests.oracle_lib import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1"])
def test_ecdsa_verify_rfc6979_signatures(engine, curve):
    """verify: w = s^-1, u1 = e*w, u2 = r*w, accept iff x(u1*G + u2*Q) mod n == r."""
    c = R.CURVES[curve]
    v = golden("rfc6979.json")[curve]
    q = bytes.fromhex(v["ux"].rjust(2 * c.fb, "0")) + bytes.fromhex(v["uy"].rjust(2 * c.fb, "0"))
    u1s, u2s, rs = b"", b"", []
    for kat in v["sign_kats"]:
        h = getattr(hashlib, kat["alg"])(kat["message"].encode()).digest()
        e = int.from_bytes(h, "big")
        if 8 * len(h) > c.n.bit_length():                  # bits2int: keep the leftmost qlen bits
            e >>= 8 * len(h) - c.n.bit_length()
        r, s = int(kat["r"], 16), int(kat["s"], 16)
        w = pow(s, -1, c.n)
        u1s += (e * w % c.n).to_bytes(c.sb, "big")
        u2s += (r * w % c.n).to_bytes(c.sb, "big")
        rs.append(r)
    n = len(rs)
    out, flags = engine.double_scalarmul(curve, u1s, u2s, q * n)
    for i, r in enumerate(rs):
        assert flags[i] == 0
        assert int.from_bytes(out[i * 2 * c.fb: i * 2 * c.fb + c.fb], "big") % c.n == r
    # a tampered signature does not verify
    bad = bytearray(u2s)
    for i in range(n):
        bad[(i + 1) * c.sb - 1] ^= 1
    out, flags = engine.double_scalarmul(curve, u1s, bytes(bad), q * n)
    assert all(int.from_bytes(out[i * 2 * c.fb: i * 2 * c.fb + c.fb], "big") % c.n != r for i, r in enumerate(rs))


@pytest.mark.parametrize("curve", ["p256r1", "bls12_381_g1"])
def test_double_scalarmul_matches_oracle(engine, oracle, curve):
    c = R.CURVES[curve]
    n = 48
    u1 = W.random_scalars(curve, n, seed=71).tobytes()
    u2 = bytearray(W.random_scalars(curve, n, seed=72).tobytes())
    q = oracle.base(curve, W.random_scalars(curve, n, seed=73).tobytes())[0]
    u2[0:c.sb] = bytes(c.sb)                                # u2 = 0: result u1*G
    u1b = bytearray(u1)
    u1b[c.sb:2 * c.sb] = bytes(c.sb)                        # u1 = 0: result u2*Q
    u1, u2 = bytes(u1b), bytes(u2)
    A = oracle.base(curve, u1)
    B = oracle.var(curve, u2, q)
    pb = 2 * c.fb

    def pt(buf, fl, i):
        return None if fl[i] else (int.from_bytes(buf[i * pb:i * pb + c.fb], "big"), int.from_bytes(buf[i * pb + c.fb:(i + 1) * pb], "big"))

    for subtract in (False, True):
        out, flags = engine.double_scalarmul(curve, u1, u2, q, subtract=subtract)
        for i in range(n):
            P, Q = pt(A[0], A[1], i), pt(B[0], B[1], i)
            if subtract and Q is not None:
                Q = (Q[0], (-Q[1]) % c.p)
            want = R.affine_add(c, P, Q)
            enc = bytes(pb) if want is None else want[0].to_bytes(c.fb, "big") + want[1].to_bytes(c.fb, "big")
            assert out[i * pb:(i + 1) * pb] == enc and flags[i] == (1 if want is None else 0)


def test_ed25519_verify_shape(engine, oracle):
    """[s]B - [k]A == R for honest signatures: with A = a*B, R = r*B and s = r + k*a (mod l)
    (src/protocol/ed25519.rs:91-112 sign, :127-150 verify)."""
    c = R.ED25519
    rng = random.Random(8032)
    n = 64
    a = [rng.randrange(1, c.n) for _ in range(n)]
    r = [rng.randrange(1, c.n) for _ in range(n)]
    k = [rng.randrange(1, c.n) for _ in range(n)]
    s = [(ri + ki * ai) % c.n for ri, ki, ai in zip(r, k, a)]
    be = lambda xs: b"".join(x.to_bytes(32, "big") for x in xs)
    A = oracle.base("ed25519", be(a))[0]
    Rpt = oracle.base("ed25519", be(r))[0]
    out, flags = engine.double_scalarmul("ed25519", be(s), be(k), A, subtract=True)
    assert out == Rpt and flags == bytes(n)


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1", "bls12_381_g1"])
def test_fused_accumulation_special_cases(engine, oracle, curve):
    """The fused kernel adds the comb of u1*G onto the Jacobian u2*Q with incomplete (mixed)
    additions.  With Q = G the two halves can be made to collide:
      u2 = d*256^w, u1 = d*256^w      -> accumulator == table entry      (doubling case)
      u1 = n - u2                     -> accumulator == -(last entry)    (infinity)
      u1 = 0, u2 = 0, u2 = n          -> accumulator at infinity before / after the comb
    and the same with subtraction (Q negated)."""
    c = R.CURVES[curve]
    g = oracle.base(curve, (1).to_bytes(c.sb, "big"))[0]
    rng = random.Random(99)
    pairs = []
    for w, d in ((0, 1), (0, 200), (1, 7), (5, 255), (c.sb - 2, 3)):
        k = d << (8 * w)
        pairs += [(k, k), (c.n - k, k), (k, c.n - k)]
    pairs += [(0, 0), (0, c.n), (c.n, 0), (5, 0), (0, 5), (c.n - 1, 1), (1, c.n - 1)]
    for _ in range(8):
        k = rng.randrange(1, c.n)
        pairs += [(c.n - k, k), (k, k)]
    n = len(pairs)
    u1 = b"".join(a.to_bytes(c.sb, "big") for a, _ in pairs)
    u2 = b"".join(b.to_bytes(c.sb, "big") for _, b in pairs)
    pb = 2 * c.fb
    for subtract in (False, True):
        out, flags = engine.double_scalarmul(curve, u1, u2, g * n, subtract=subtract)
        tot = [((a - b) if subtract else (a + b)) % c.n for a, b in pairs]
        want = oracle.base(curve, b"".join(t.to_bytes(c.sb, "big") for t in tot))
        assert out == want[0] and flags == want[1], (curve, subtract)
        assert any(f == 1 for f in flags)   # some of them are the point at infinity


def test_ed25519_double_scalarmul_matches_oracle(engine, oracle):
    """[u1]B +- [u2]A against the big-int oracle, on subgroup points, on points with an 8-torsion
    part, and with zero / edge scalars on either side (the fused Edwards kernel has no special
    cases: the unified additions are complete)."""
    from tests.test_gpu_parity import _ed25519_points_outside_the_subgroup

    c = R.ED25519
    rng = random.Random(25519)
    le = lambda v: v.to_bytes(32, "little")
    sub = oracle.base("ed25519", W.random_scalars("ed25519", 24, seed=5).tobytes())[0]
    mixed = b"".join(le(x) + le(y) for x, y in _ed25519_points_outside_the_subgroup(16))
    special = le(0) + le(1) + le(0) + le(c.p - 1)
    q = sub + mixed + special
    n = len(q) // 64
    edge = [0, 1, 2, 8, c.n - 1, c.n, c.n + 1, (1 << 256) - 1]
    u1 = [rng.randrange(1 << 256) if i % 3 else edge[i % len(edge)] for i in range(n)]
    u2 = [rng.randrange(1 << 256) if i % 4 else edge[(i // 4) % len(edge)] for i in range(n)]
    be = lambda xs: b"".join(x.to_bytes(32, "big") for x in xs)
    A = oracle.base("ed25519", be(u1))[0]
    B = oracle.var("ed25519", be(u2), q)[0]
    pt = lambda buf, i: (int.from_bytes(buf[64 * i:64 * i + 32], "little"), int.from_bytes(buf[64 * i + 32:64 * i + 64], "little"))
    for subtract in (False, True):
        out, flags = engine.double_scalarmul("ed25519", be(u1), be(u2), q, subtract=subtract)
        for i in range(n):
            P, Q = pt(A, i), pt(B, i)
            if subtract:
                Q = ((-Q[0]) % c.p, Q[1])
            x, y = R.ed_affine_add(c, P, Q)
            assert out[64 * i:64 * i + 64] == le(x) + le(y), (i, subtract)
            assert flags[i] == (1 if (x, y) == (0, 1) else 0)


def test_double_scalarmul_device_tensors(engine, oracle):
    """eccx_double_scalarmul_dev on torch tensors, on the caller's stream, against the host-buffer form."""
    import torch

    curve, n = "p256r1", 700
    u1 = W.random_scalars(curve, n, seed=81)
    u2 = W.random_scalars(curve, n, seed=82)
    q = oracle.base(curve, W.random_scalars(curve, n, seed=83).tobytes())[0]
    want = engine.double_scalarmul(curve, u1.tobytes(), u2.tobytes(), q)
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        out, flags = engine.double_scalarmul_t(curve, t(u1.tobytes()).reshape(n, -1), t(u2.tobytes()).reshape(n, -1),
                                               t(q).reshape(n, -1), stream=stream.cuda_stream)
    stream.synchronize()
    assert out.cpu().numpy().tobytes() == want[0] and flags.cpu().numpy().tobytes() == want[1]


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1", "bls12_381_g1"])
def test_x_only_output(engine, oracle, curve):
    """ECCX_OUT_X_ONLY: FB bytes per unit, the x of the full result (Point::to_affine_x_ct, src/curve/projective.rs:690,
    consumed by verify at src/protocol/ecdsa.rs:383): the RFC 6979 signatures verify through it, infinity and rejected
    inputs keep their flags, and every x equals the first half of the x||y output."""
    c = R.CURVES[curve]
    n = 300
    u1 = bytearray(W.random_scalars(curve, n, seed=81).tobytes())
    u2 = bytearray(W.random_scalars(curve, n, seed=82).tobytes())
    q = bytearray(oracle.base(curve, W.random_scalars(curve, n, seed=83).tobytes(), threads=16)[0])
    u1[0:c.sb] = bytes(c.sb)
    u2[0:c.sb] = bytes(c.sb)                                 # unit 0: 0*G + 0*Q = infinity
    q[5 * 2 * c.fb + c.fb - 1] ^= 1                          # unit 5: Q off the curve
    u1, u2, q = bytes(u1), bytes(u2), bytes(q)
    full, fl = engine.double_scalarmul(curve, u1, u2, q, validate=True)
    xs, flx = engine.double_scalarmul(curve, u1, u2, q, validate=True, x_only=True)
    assert len(xs) == n * c.fb and flx == fl and fl[0] == 1 and fl[5] == 2
    for i in range(n):
        assert xs[i * c.fb:(i + 1) * c.fb] == full[i * 2 * c.fb:i * 2 * c.fb + c.fb]
    if curve in golden("rfc6979.json"):
        v = golden("rfc6979.json")[curve]
        qk = bytes.fromhex(v["ux"].rjust(2 * c.fb, "0")) + bytes.fromhex(v["uy"].rjust(2 * c.fb, "0"))
        u1s, u2s, rs = b"", b"", []
        for kat in v["sign_kats"]:
            h = getattr(hashlib, kat["alg"])(kat["message"].encode()).digest()
            e = int.from_bytes(h, "big")
            if 8 * len(h) > c.n.bit_length():
                e >>= 8 * len(h) - c.n.bit_length()
            r, s = int(kat["r"], 16), int(kat["s"], 16)
            w = pow(s, -1, c.n)
            u1s += (e * w % c.n).to_bytes(c.sb, "big")
            u2s += (r * w % c.n).to_bytes(c.sb, "big")
            rs.append(r)
        xs, flx = engine.double_scalarmul(curve, u1s, u2s, qk * len(rs), x_only=True)
        assert all(f == 0 for f in flx)
        assert [int.from_bytes(xs[i * c.fb:(i + 1) * c.fb], "big") % c.n for i in range(len(rs))] == rs


def test_x_only_is_refused_for_edwards(engine):
    import eccoxide_amd as E

    with pytest.raises(E.EccxError) as ei:
        engine.double_scalarmul("ed25519", bytes(32), bytes(32), bytes(64), x_only=True)
    assert "X_ONLY" in str(ei.value)
