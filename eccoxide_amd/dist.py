"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" on CPU for tests).

Units (scalar, base) are independent, so the batch is split into contiguous shards
 and every rank runs the single-GPU path on its shard with NO data-path
collective.  The only exchange is the final gather of the result bytes to rank 0 — one
message of (n/world) * 2FB bytes per peer, each over its own point-to-point xGMI link into
the root.  The reference has no distributed code at all; this is the batch API the build
adds around `&Point * &Scalar` (src/curve/fiat/curve_macros.rs:321-327).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .workload import shard_bounds


def gather_to_root(local: torch.Tensor, sizes, root: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather variable-length first-dimension shards to `root`; returns the concatenation on
    root, None elsewhere.  Shards may be ragged (n not divisible by world)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    rest = tuple(local.shape[1:])
    if len(set(sizes)) == 1:
        # equal shards: a single gather
        bufs = [torch.empty((sizes[0],) + rest, dtype=local.dtype, device=local.device) for _ in range(world)] \
            if rank == root else None
        dist.gather(local.contiguous(), bufs, dst=root, group=group)
        return torch.cat(bufs, dim=0) if rank == root else None
    # ragged: point-to-point into the root
    if rank == root:
        parts = []
        reqs = []
        for r in range(world):
            if r == root:
                parts.append(local)
            else:
                buf = torch.empty((sizes[r],) + rest, dtype=local.dtype, device=local.device)
                if sizes[r]:
                    reqs.append(dist.irecv(buf, src=r, group=group))
                parts.append(buf)
        for q in reqs:
            q.wait()
        return torch.cat(parts, dim=0)
    if sizes[rank]:
        dist.send(local.contiguous(), dst=root, group=group)
    return None


class GatherPipeline:
    """Gather of equal result shards to the root, overlapped with the next batch.

    The root pre-allocates `slots` receive areas (world x shard each) once; `start(slot, out,
    flags)` enqueues the two gathers of one batch asynchronously -- on RCCL they run on the
    communicator's own stream, each peer sending over its own xGMI link -- and `finish(slot)`
    makes the caller's stream wait for them.  With two slots batch i+1 computes (into the other
    pair of output buffers) while batch i is on the wire; a slot must be finished before its
    source buffers are written again.  Results land contiguously (no concatenation copy):
    `result(slot)` returns (out, flags) views on the root, (None, None) elsewhere.
    """

    def __init__(self, shard_rows: int, out_cols: int, device, slots: int = 2, root: int = 0, group=None,
                 force: bool = False):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.root, self.group, self.rows = root, group, shard_rows
        self.pending = [[] for _ in range(slots)]
        self.local = [None] * slots
        self.out = self.flags = None
        # force: run the collective even in a group of one (bench.py ECCX_FORCE_DIST: one GPU
        # exercising the RCCL path -- communicator, gather on its own stream, stream ordering)
        self.active = self.world > 1 or (force and dist.is_initialized())
        if self.active and self.rank == root:
            self.out = [torch.empty((self.world * shard_rows, out_cols), dtype=torch.uint8, device=device)
                        for _ in range(slots)]
            self.flags = [torch.empty((self.world * shard_rows,), dtype=torch.uint8, device=device)
                          for _ in range(slots)]

    def start(self, slot: int, out: torch.Tensor, flags: torch.Tensor) -> None:
        assert not self.pending[slot], "slot still in flight: call finish(slot) first"
        self.local[slot] = (out, flags)
        if not self.active:
            return
        for src, dst in ((out, self.out), (flags, self.flags)):
            bufs = list(dst[slot].split(self.rows, dim=0)) if self.rank == self.root else None
            self.pending[slot].append(dist.gather(src, bufs, dst=self.root, group=self.group, async_op=True))

    def finish(self, slot: int) -> None:
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []

    def result(self, slot: int):
        if not self.active:
            return self.local[slot]
        if self.rank != self.root:
            return None, None
        return self.out[slot], self.flags[slot]


def engine_compute(engine, curve, **opts):
    """The per-rank compute of sharded_scalarmul for an Engine (or anything with its device-tensor
    API): variable base when a points shard is given, fixed base otherwise.  Shards are slices of
    the global tensors, so they are made contiguous before their addresses reach the C ABI."""
    def compute(scalars, points):
        if points is None:
            return engine.scalarmul_base_t(curve, scalars.contiguous(), **opts)
        return engine.scalarmul_var_t(curve, scalars.contiguous(), points.contiguous(), **opts)
    return compute


def sharded_scalarmul(compute: Callable[[torch.Tensor, Optional[torch.Tensor]], Tuple[torch.Tensor, torch.Tensor]],
                      scalars: torch.Tensor, points: Optional[torch.Tensor], *, gather: bool = True, root: int = 0,
                      group=None):
    """Run `compute(scalars_shard, points_shard) -> (out, flags)` on this rank's contiguous
    shard of the global batch and gather the results on `root`.

    `scalars` / `points` are the GLOBAL (n x SB / n x 2FB) uint8 tensors, identical on every
    rank (or at least valid on the rank's own shard).  Returns (out, flags) for the whole
    batch on root (None, None elsewhere) when gather=True, else this rank's shard results.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = scalars.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    out, flags = compute(scalars[lo:hi], points[lo:hi] if points is not None else None)
    if not gather or world == 1:
        return out, flags
    sizes = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
    g_out = gather_to_root(out, sizes, root, group)
    g_flags = gather_to_root(flags, sizes, root, group)
    return g_out, g_flags
