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


class _FakeCudaTensor:
    """Stands in for a CUDA tensor where there is no GPU: what Engine._tensors looks at."""

    def __init__(self, nbytes, index=0, contiguous=True, dtype=None):
        import torch

        self._n, self._c = nbytes, contiguous
        self.is_cuda = True
        self.dtype = torch.uint8 if dtype is None else dtype
        self.device = type("D", (), {"index": index})()

    def numel(self):
        return self._n

    def is_contiguous(self):
        return self._c


def test_device_tensor_arguments_are_checked_before_a_kernel_sees_them():
    """A short buffer or a tensor of another GPU must be a ValueError, not an out-of-bounds or
    cross-device access inside a kernel (the checks run before any call into the library)."""
    import torch

    import eccoxide_amd as E

    eng = E.Engine.__new__(E.Engine)   # no context: the checks must fire first
    eng.device = 0
    eng._lib = E._lib.load()
    T = _FakeCudaTensor
    n, sb, fb = 10, 32, 32
    ok = dict(scalars=T(n * sb), points=T(n * 2 * fb), out=T(n * 2 * fb), flags=T(n))
    bad = [
        dict(ok, scalars=T(n * sb + 5)),               # not a whole number of scalars
        dict(ok, points=T((n - 1) * 2 * fb)),          # short points
        dict(ok, out=T(n * 2 * fb - 1)),               # short out
        dict(ok, flags=T(n + 1)),                      # long flags
        dict(ok, points=T(n * 2 * fb, index=1)),       # other GPU
        dict(ok, out=T(n * 2 * fb, contiguous=False)),
        dict(ok, flags=T(n, dtype=torch.int32)),
        dict(ok, scalars=torch.zeros(n * sb, dtype=torch.uint8)),   # host tensor
    ]
    for kw in bad:
        with pytest.raises(ValueError):
            eng.scalarmul_var_t("p256r1", kw["scalars"], kw["points"], kw["out"], kw["flags"], stream=0)
    with pytest.raises(ValueError):
        eng.scalarmul_var_t("p256r1", ok["scalars"], ok["points"], ok["out"], ok["flags"], proj=T(n * 3 * fb - 1), stream=0)
    with pytest.raises(ValueError):
        eng.scalarmul_base_t("p256r1", ok["scalars"], T(n * 2 * fb), T(n - 1), stream=0)
    with pytest.raises(ValueError):
        eng.x25519_t(T(n * 32), T(n * 32, index=3), T(n * 32), T(n), stream=0)
    with pytest.raises(ValueError):
        eng.point_add_t("p384r1", T(n * 96), T(n * 96), T(n * 96), T(n), a_inf=T(n - 2), stream=0)
    with pytest.raises(ValueError):
        eng.double_scalarmul_t("p256r1", T(n * 32), T(n * 32 + 32), T(n * 64), T(n * 64), T(n), stream=0)
    with pytest.raises(ValueError):
        eng.point_decompress_t("p256r1", T(n * 33), T(n * 64), T(n + 4), stream=0)
    with pytest.raises(ValueError):
        eng.point_compress_t("p256r1", T(n * 64), T(n), T(n * 33 - 1), stream=0)
