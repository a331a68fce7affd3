"""The N>1 path (contiguous shards, no data-path collective, one gather to rank 0) with
world_size 2 on the gloo backend.  The per-rank compute is the CPU oracle here (test
infrastructure standing in for the GPU engine); what is under test is the sharding and
gather logic of eccoxide_amd/dist.py.  CPU only.  (The same pipeline on RCCL with the real
engine: tests/test_gpu_api.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eccoxide_amd import workload as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, curve, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eccoxide_amd.dist import sharded_scalarmul
        from tests import oracle_lib

        ora = oracle_lib.load()
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=11))
        rs = W.random_scalars(curve, n, seed=12)
        pts_b, _, _ = ora.base(curve, rs.tobytes(), threads=2)
        pts = torch.frombuffer(bytearray(pts_b), dtype=torch.uint8).reshape(n, -1)

        def compute(k, p):
            o, f, _ = ora.var(curve, k.contiguous().numpy().tobytes(), p.contiguous().numpy().tobytes(), threads=2)
            fb2 = p.shape[1]
            return (torch.frombuffer(bytearray(o), dtype=torch.uint8).reshape(-1, fb2) if len(o) else torch.empty((0, fb2), dtype=torch.uint8),
                    torch.frombuffer(bytearray(f), dtype=torch.uint8) if len(f) else torch.empty((0,), dtype=torch.uint8))

        out, flags = sharded_scalarmul(compute, ks, pts)
        if rank == 0:
            full_o, full_f, _ = ora.var(curve, ks.numpy().tobytes(), pts.numpy().tobytes(), threads=2)
            q.put((out.numpy().tobytes() == full_o, flags.numpy().tobytes() == full_f, tuple(out.shape)))
        else:
            assert out is None and flags is None
            q.put(("nonroot", True, None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 37])  # equal shards, and ragged shards (19 / 18)
def test_world2_sharded_matches_single(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, "p256r1", q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in res if r[0] != "nonroot"][0]
    assert root[0] is True and root[1] is True and root[2] == (n, 64)


def _pipeline_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eccoxide_amd.dist import GatherPipeline

        rows, cols, steps = 33, 8, 5
        pipe = GatherPipeline(rows, cols, torch.device("cpu"), slots=2)
        outs = [torch.empty((rows, cols), dtype=torch.uint8) for _ in range(2)]
        flags = [torch.empty((rows,), dtype=torch.uint8) for _ in range(2)]

        def batch(step, r):  # what rank r produces at this step
            g = torch.Generator().manual_seed(1000 * step + r)
            return (torch.randint(0, 256, (rows, cols), dtype=torch.uint8, generator=g),
                    torch.randint(0, 3, (rows,), dtype=torch.uint8, generator=g))

        ok = True
        for i in range(steps):
            slot = i & 1
            pipe.finish(slot)
            if rank == 0 and i >= 2:  # the batch that used this slot two steps ago is complete and intact
                o, f = pipe.result(slot)
                want = [batch(i - 2, r) for r in range(world)]
                ok &= torch.equal(o, torch.cat([w[0] for w in want])) and torch.equal(f, torch.cat([w[1] for w in want]))
            o, f = batch(i, rank)
            outs[slot].copy_(o)
            flags[slot].copy_(f)
            pipe.start(slot, outs[slot], flags[slot])
        pipe.finish(0)
        pipe.finish(1)
        if rank == 0:
            for i in (steps - 2, steps - 1):
                o, f = pipe.result(i & 1)
                want = [batch(i, r) for r in range(world)]
                ok &= torch.equal(o, torch.cat([w[0] for w in want])) and torch.equal(f, torch.cat([w[1] for w in want]))
        else:
            ok &= pipe.result(0) == (None, None)
        q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world2_gather_pipeline_overlaps_batches():
    """bench.py's N > 1 loop: two slots, gather of batch i in flight while batch i+1 is produced."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [True, True]


class OracleEngine:
    """Stands in for eccoxide_amd.Engine on a box without a GPU: the same device-tensor API shape
    (uint8 tensors in, (out, flags) tensors out), computed by the oracle.  Test infrastructure."""

    def __init__(self, threads=2):
        from tests import oracle_lib

        self.ora, self.threads = oracle_lib.load(), threads

    @staticmethod
    def _t(buf, cols):
        t = torch.frombuffer(bytearray(buf), dtype=torch.uint8) if len(buf) else torch.empty((0,), dtype=torch.uint8)
        return t.reshape(-1, cols) if cols else t

    def scalarmul_var_t(self, curve, scalars, points, out=None, flags=None, **_):
        assert scalars.is_contiguous() and points.is_contiguous()
        o, f, _p = self.ora.var(curve, scalars.numpy().tobytes(), points.numpy().tobytes(), threads=self.threads)
        return self._t(o, points.shape[1]), self._t(f, 0)

    def scalarmul_base_t(self, curve, scalars, out=None, flags=None, **_):
        assert scalars.is_contiguous()
        o, f, _p = self.ora.base(curve, scalars.numpy().tobytes(), threads=self.threads)
        return self._t(o, 2 * self.ora.fb(curve)), self._t(f, 0)


def _engine_api_worker(rank, world, port, n, curve, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eccoxide_amd.dist import engine_compute, sharded_scalarmul

        eng = OracleEngine()
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=21))
        rs = torch.from_numpy(W.random_scalars(curve, n, seed=22))
        # fixed base, sharded: every rank computes its slice of r_i * G, the root gets all of them
        pts, pf = sharded_scalarmul(engine_compute(eng, curve), rs, None)
        full_pts, full_pf = eng.scalarmul_base_t(curve, rs)
        ok = True
        if rank == 0:
            ok &= torch.equal(pts, full_pts) and torch.equal(pf, full_pf)
        # variable base, sharded, on the (replicated) bases
        out, flags = sharded_scalarmul(engine_compute(eng, curve), ks, full_pts)
        if rank == 0:
            want_o, want_f = eng.scalarmul_var_t(curve, ks, full_pts)
            ok &= torch.equal(out, want_o) and torch.equal(flags, want_f) and tuple(out.shape) == (n, full_pts.shape[1])
        else:
            ok &= out is None and flags is None
        # gather=False: the rank keeps its own shard
        mine, _ = sharded_scalarmul(engine_compute(eng, curve), ks, full_pts, gather=False)
        lo, hi = W.shard_bounds(n, world, rank)
        ok &= mine.shape[0] == hi - lo
        q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("curve,n", [("p256r1", 41), ("ed25519", 30)])
def test_world2_sharded_through_the_engine_tensor_api(curve, n):
    """sharded_scalarmul(engine_compute(engine, curve), ...) -- the call a multi-GPU host makes --
    with an object of the Engine's tensor-API shape on every rank: fixed and variable base, ragged
    and equal shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_api_worker, args=(r, 2, port, n, curve, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [True, True]
