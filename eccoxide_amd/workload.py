"""Synthetic workloads for the scalar-multiplication engine:
uniform random canonical scalars (big-endian, < group order, non-zero) from a seeded
generator; base points r_i*G are produced by the engine's own fixed-base path.
"""
from __future__ import annotations

import numpy as np

# group orders (SEC 2 / BLS12-381 / RFC 8032), big-endian hex
ORDERS = {
    "p256r1": "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551",
    "p384r1": "ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973",
    "p521r1": "01fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffa51868783bf2f966b7fcc0148f709a5d03bb5c9b8899c47aebb6fb71e91386409",
    "bls12_381_g1": "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001",
    "ed25519": "1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3ed",
}
SEED_BASE = 0xECC051DE00000000


def order(curve: str) -> int:
    return int(ORDERS[curve], 16)


def _lex_less(a: np.ndarray, bound: np.ndarray) -> np.ndarray:
    """row-wise big-endian a < bound (a: n x k uint8, bound: k uint8)"""
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for j in range(a.shape[1]):
        lt |= eq & (a[:, j] < bound[j])
        eq &= a[:, j] == bound[j]
    return lt


def random_scalars(curve: str, n: int, seed: int = 0) -> np.ndarray:
    """n x SB uint8, big-endian, uniform in [1, order)."""
    nb = bytes.fromhex(ORDERS[curve])
    sb = len(nb)
    bound = np.frombuffer(nb, dtype=np.uint8)
    top_mask = (1 << int(nb[0]).bit_length()) - 1
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    out = np.empty((n, sb), dtype=np.uint8)
    todo = np.arange(n)
    while todo.size:
        cand = rng.integers(0, 256, size=(todo.size, sb), dtype=np.uint8)
        cand[:, 0] &= top_mask
        ok = _lex_less(cand, bound) & cand.any(axis=1)
        out[todo[ok]] = cand[ok]
        todo = todo[~ok]
    return out


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous block split of n units over `world` ranks."""
    return n * rank // world, n * (rank + 1) // world
