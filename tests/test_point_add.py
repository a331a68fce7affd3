"""Batched group law (eccx_point_add): impl Add / Sub / Neg for Point and CurveGroup::double
(src/curve/fiat/curve_macros.rs:297-411, src/curve/group.rs:28-70) against the Python big-int
oracle, including the cases the reference's completeness tests cover
(src/tests/completeness.rs:21-57: P + P, P + (-P), P + O, homomorphism)."""
import pytest

from eccoxide_amd import workload as W
from oracle import ecc_ref as R

pytestmark = pytest.mark.gpu

WEI = ["p256r1", "p384r1", "p521r1", "bls12_381_g1"]


def _enc(c, P):
    return bytes(2 * c.fb) if P is None else P[0].to_bytes(c.fb, "big") + P[1].to_bytes(c.fb, "big")


@pytest.mark.parametrize("curve", WEI)
def test_point_add_weierstrass(engine, oracle, curve):
    c = R.CURVES[curve]
    n = 40
    ka = W.random_scalars(curve, n, seed=51).tobytes()
    kb = W.random_scalars(curve, n, seed=52).tobytes()
    A = oracle.base(curve, ka)[0]
    B = bytearray(oracle.base(curve, kb)[0])
    pb = 2 * c.fb
    B[0:pb] = A[0:pb]                                            # unit 0: B == A  (doubling)
    ax, ay = A[pb:pb + c.fb], int.from_bytes(A[pb + c.fb:2 * pb], "big")
    B[pb:2 * pb] = ax + ((-ay) % c.p).to_bytes(c.fb, "big")      # unit 1: B == -A (infinity)
    a_inf = bytearray(n)
    b_inf = bytearray(n)
    a_inf[2] = 1                                                 # unit 2: O + B
    b_inf[3] = 1                                                 # unit 3: A + O
    a_inf[4] = b_inf[4] = 1                                      # unit 4: O + O
    B = bytes(B)

    def pt(buf, inf, i):
        if inf[i]:
            return None
        return (int.from_bytes(buf[i * pb:i * pb + c.fb], "big"), int.from_bytes(buf[i * pb + c.fb:(i + 1) * pb], "big"))

    for subtract, mirror in ((False, False), (True, False), (False, True), (True, True)):
        out, flags = engine.point_add(curve, A, B, a_inf=bytes(a_inf), b_inf=bytes(b_inf), subtract=subtract, mirror=mirror)
        for i in range(n):
            P, Q = pt(A, a_inf, i), pt(B, b_inf, i)
            if subtract and Q is not None:
                Q = (Q[0], (-Q[1]) % c.p)
            want = R.affine_add(c, P, Q)
            assert out[i * pb:(i + 1) * pb] == _enc(c, want), (curve, i, subtract)
            assert flags[i] == (1 if want is None else 0)
    # homomorphism through the scalar-multiplication kernels: a*G + b*G == (a + b)*G
    out, flags = engine.point_add(curve, A, oracle.base(curve, kb)[0])
    ks = b"".join(((int.from_bytes(ka[i * c.sb:(i + 1) * c.sb], "big") + int.from_bytes(kb[i * c.sb:(i + 1) * c.sb], "big")) % c.n)
                  .to_bytes(c.sb, "big") for i in range(n))
    assert (out, flags) == engine.scalarmul_base(curve, ks)
    # double(P) == 2 * P
    out, flags = engine.point_add(curve, A, A)
    two = (2).to_bytes(c.sb, "big") * n
    assert (out, flags) == engine.scalarmul_var(curve, two, A)


def test_point_add_ed25519(engine, oracle):
    c = R.ED25519
    n = 40
    ka = W.random_scalars("ed25519", n, seed=61).tobytes()
    kb = W.random_scalars("ed25519", n, seed=62).tobytes()
    A = oracle.base("ed25519", ka)[0]
    B = bytearray(oracle.base("ed25519", kb)[0])
    B[0:64] = A[0:64]                                                           # doubling
    x = int.from_bytes(A[64:96], "little")
    B[64:128] = ((-x) % c.p).to_bytes(32, "little") + A[96:128]                 # A + (-A) = neutral
    B[128:192] = (0).to_bytes(32, "little") + (1).to_bytes(32, "little")        # A + neutral
    B = bytes(B)
    for subtract, mirror in ((False, False), (True, False), (False, True), (True, True)):
        out, flags = engine.point_add("ed25519", A, B, subtract=subtract, mirror=mirror)
        for i in range(n):
            P = (int.from_bytes(A[i * 64:i * 64 + 32], "little"), int.from_bytes(A[i * 64 + 32:(i + 1) * 64], "little"))
            Q = (int.from_bytes(B[i * 64:i * 64 + 32], "little"), int.from_bytes(B[i * 64 + 32:(i + 1) * 64], "little"))
            if subtract:
                Q = ((-Q[0]) % c.p, Q[1])
            want = R.ed_affine_add(c, P, Q)
            assert out[i * 64:(i + 1) * 64] == want[0].to_bytes(32, "little") + want[1].to_bytes(32, "little")
            assert flags[i] == (1 if want == (0, 1) else 0)


@pytest.mark.parametrize("curve", ["p256r1", "bls12_381_g1", "ed25519"])
def test_point_add_device_tensors(engine, oracle, curve):
    """eccx_point_add_dev on torch tensors (with and without infinity flags, both signs), on the
    caller's stream, against the host-buffer form."""
    import torch

    n = 1500
    fb = oracle.fb(curve)
    A = oracle.base(curve, W.random_scalars(curve, n, seed=61).tobytes())[0]
    B = bytearray(oracle.base(curve, W.random_scalars(curve, n, seed=62).tobytes())[0])
    B[:2 * fb * 10] = A[:2 * fb * 10]          # doublings (cancellations under subtract)
    B = bytes(B)
    a_inf = bytes(1 if i % 97 == 5 else 0 for i in range(n))
    b_inf = bytes(1 if i % 89 == 7 else 0 for i in range(n))
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    stream = torch.cuda.Stream(dev)
    for subtract in (False, True):
        for with_inf in ((False, True) if curve != "ed25519" else (False,)):
            want = engine.point_add(curve, A, B, a_inf=a_inf if with_inf else None, b_inf=b_inf if with_inf else None,
                                    subtract=subtract)
            with torch.cuda.stream(stream):
                out, flags = engine.point_add_t(curve, t(A).reshape(n, -1), t(B).reshape(n, -1),
                                                a_inf=t(a_inf) if with_inf else None, b_inf=t(b_inf) if with_inf else None,
                                                subtract=subtract, stream=stream.cuda_stream)
            stream.synchronize()
            assert out.cpu().numpy().tobytes() == want[0] and flags.cpu().numpy().tobytes() == want[1], (subtract, with_inf)
            # the saturated-limb kernels give the same bytes
            assert want == engine.point_add(curve, A, B, a_inf=a_inf if with_inf else None, b_inf=b_inf if with_inf else None,
                                            subtract=subtract, mirror=True)
