"""X25519 / curve25519 Montgomery ladder: oracle pinned by the reference's
RFC 7748 vectors and ladder KATs (CPU); HIP path bit-exact against the oracle (GPU)."""
import random

import numpy as np
import pytest

from oracle import ecc_ref as R
from tests.oracle_lib import golden

P = 2**255 - 19


def test_oracles_match_rfc7748_and_ladder_kats(oracle):
    g = golden("x25519.json")
    for v in g["rfc7748"]:  # src/protocol/x25519.rs:120-147
        k, u, r = (bytes.fromhex(v[x]) for x in ("k", "u", "r"))
        assert R.ref_x25519(k, u) == r
        assert oracle.x25519(k, u)[0] == r
    dh = g["diffie_hellman"]  # x25519.rs:149-166 (RFC 7748 section 6.1)
    a, b = bytes.fromhex(dh["a"]), bytes.fromhex(dh["b"])
    base = (9).to_bytes(32, "little")
    pa, pb = R.ref_x25519(a, base), R.ref_x25519(b, base)
    assert R.ref_x25519(a, pb).hex() == dh["shared"] == R.ref_x25519(b, pa).hex()
    assert oracle.x25519(a, None)[0] == pa and oracle.x25519(a, pb)[0].hex() == dh["shared"]
    for k, want in g["ladder_u9"].items():  # src/curve/curve25519.rs:1629-1652
        kb = int(k).to_bytes(1, "big")
        assert R.ref_ladder(9, kb) == int(want, 16)
        out, fl = oracle.x25519(int(k).to_bytes(32, "big"), None, rfc=False)
        assert int.from_bytes(out, "little") == int(want, 16) and fl == b"\0"
    assert R.ref_ladder(9, b"\x01") == 9


def test_ladder_matches_edwards(oracle):
    """curve25519.rs:1654-1665: [k](u=9) via the ladder == montgomery-u of [k]*B (Edwards)."""
    c = R.ED25519
    rng = random.Random(3)
    for k in [1, 2, 3, 4, 5, 9, 1000, 1234567] + [rng.randrange(1, c.n) for _ in range(4)]:
        x, y = R.ed_affine_mul(c, k, (c.gx, c.gy))
        want = R.montgomery_u_of_edwards(y)
        kb = k.to_bytes(32, "big")
        assert R.ref_ladder(9, kb) == want
        assert int.from_bytes(oracle.x25519(kb, None, rfc=False)[0], "little") == want


def test_c_oracle_matches_python_oracle_random(oracle):
    rng = random.Random(7748)
    n = 24
    ks = bytes(rng.getrandbits(8) for _ in range(32 * n))
    us = bytearray(rng.getrandbits(8) for _ in range(32 * n))
    us[0:32] = (0).to_bytes(32, "little")                 # u = 0: low order -> zero result
    us[32:64] = (1).to_bytes(32, "little")                # u = 1: order 4
    us[64:96] = (P - 1).to_bytes(32, "little")            # u = -1
    us[96:128] = (P + 5).to_bytes(32, "little")           # non-canonical encoding of 5
    us[128:160] = b"\xff" * 32                            # top bit set
    for rfc in (True, False):
        out, fl = oracle.x25519(ks, bytes(us), rfc=rfc)
        for i in range(n):
            k, u = ks[32 * i:32 * i + 32], bytes(us[32 * i:32 * i + 32])
            if rfc:
                want = R.ref_x25519(k, u)
            else:
                want = R.ref_ladder(int.from_bytes(u, "little"), k).to_bytes(32, "little")
            assert out[32 * i:32 * i + 32] == want
            assert fl[i] == (1 if want == bytes(32) else 0)
    assert oracle.x25519(ks, bytes(us), rfc=True)[1][0] == 1   # u = 0 is flagged


@pytest.mark.gpu
def test_gpu_x25519_matches_oracle(engine, oracle):
    g = golden("x25519.json")
    for v in g["rfc7748"]:
        k, u, r = (bytes.fromhex(v[x]) for x in ("k", "u", "r"))
        assert engine.x25519(k, u) == (r, b"\0")
    rng = np.random.Generator(np.random.PCG64(25519))
    for n in (1, 65, 700):
        ks = rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes()
        us = bytearray(rng.integers(0, 256, size=32 * n, dtype=np.uint8).tobytes())
        us[0:32] = (0).to_bytes(32, "little")
        if n > 4:
            us[32:64] = (1).to_bytes(32, "little")
            us[64:96] = (P - 1).to_bytes(32, "little")
            us[96:128] = (P + 5).to_bytes(32, "little")
            us[128:160] = b"\xff" * 32
        for raw in (False, True):
            want = oracle.x25519(ks, bytes(us), rfc=not raw, threads=16)
            got = engine.x25519(ks, bytes(us), raw_ladder=raw)
            assert got == want
        assert engine.x25519(ks, None) == oracle.x25519(ks, None, threads=16)
    assert engine.x25519(b"") == (b"", b"")


@pytest.mark.gpu
def test_gpu_x25519_diffie_hellman_full_batch(engine):
    """Size-independent property at 2^18 units: X25519(a, X25519(b, 9)) == X25519(b, X25519(a, 9))."""
    import torch

    n = 1 << 18
    g = torch.Generator().manual_seed(9)
    a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g).cuda()
    b = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g).cuda()
    pa, fa = engine.x25519_t(a)
    pb, fb = engine.x25519_t(b)
    sab, f1 = engine.x25519_t(a, pb.clone())
    sba, f2 = engine.x25519_t(b, pa.clone())
    torch.cuda.synchronize()
    assert torch.equal(sab, sba)
    assert int(f1.max()) == 0 and int(fa.max()) == 0
