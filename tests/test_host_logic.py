"""Host-side logic: synthetic workload generator, shard partition.  CPU only."""
import numpy as np
import pytest

from eccoxide_amd import workload as W


@pytest.mark.parametrize("curve", list(W.ORDERS))
def test_random_scalars_are_canonical_and_seeded(curve):
    n = 2000
    a = W.random_scalars(curve, n, seed=3)
    b = W.random_scalars(curve, n, seed=3)
    c = W.random_scalars(curve, n, seed=4)
    assert a.dtype == np.uint8 and a.shape == (n, len(W.ORDERS[curve]) // 2)
    assert (a == b).all() and not (a == c).all()
    order = W.order(curve)
    vals = [int.from_bytes(r.tobytes(), "big") for r in a]
    assert all(0 < v < order for v in vals)
    assert len(set(vals)) == n
    # top bits are exercised (not just small scalars)
    assert max(vals).bit_length() >= order.bit_length() - 1


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 64, 1000, 2**20 + 3):
        for world in (1, 2, 3, 8):
            spans = [W.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
