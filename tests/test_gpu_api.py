"""GPU tests of the C ABI's behaviour around the kernels (round 2): one-time costs and their
ordering against a busy stream, eccx_prepare / eccx_reserve, ECCX_CT_SCAN parity, the sharded entry
points over two contexts, and the RCCL gather path on one GPU (ECCX_FORCE_DIST).  Results are still
compared with the oracle bit for bit.
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from eccoxide_amd import workload as W
from tests.oracle_lib import ROOT

pytestmark = pytest.mark.gpu

WEI = ["p256r1", "p384r1", "p521r1", "bls12_381_g1"]
ALL = WEI + ["ed25519"]
_SIZES = {"p256r1": (32, 32), "p384r1": (48, 48), "p521r1": (66, 66), "bls12_381_g1": (48, 32), "ed25519": (32, 32)}


def _bases(oracle, curve, n, seed):
    return oracle.base(curve, W.random_scalars(curve, n, seed=seed).tobytes(), threads=16)[0]


# ---- ECCX_CT_SCAN: same bytes as the default kernels and the oracle -------------------------------
@pytest.mark.parametrize("curve", ALL)
def test_ct_scan_matches_oracle(engine, oracle, curve):
    """The scanning lookup (select_from_table, projective.rs:427-434 / curve25519.rs:862-869) on the
    reference-mirroring kernels: every digit value occurs in 600 random scalars, plus the edge
    scalars 0, 1, order - 1 and one with every nibble zero but the top one."""
    fb, sb = _SIZES[curve]
    n = 600
    ks = bytearray(W.random_scalars(curve, n, seed=901).tobytes())
    order = W.order(curve)
    edge = [0, 1, order - 1, 0xF << (8 * sb - 8), 0x10]
    for i, v in enumerate(edge):
        ks[i * sb:(i + 1) * sb] = (v % (1 << (8 * sb))).to_bytes(sb, "big")
    ks = bytes(ks)
    pts = _bases(oracle, curve, n, seed=902)
    want = oracle.var(curve, ks, pts, threads=16)
    got = engine.scalarmul_var(curve, ks, pts, ct_scan=True)
    assert got[0] == want[0] and got[1] == want[1]
    assert engine.scalarmul_var(curve, ks, pts) == got
    want_b = oracle.base(curve, ks, threads=16)
    got_b = engine.scalarmul_base(curve, ks, ct_scan=True)
    assert got_b[0] == want_b[0] and got_b[1] == want_b[1]
    assert engine.scalarmul_base(curve, ks) == got_b


def test_ct_scan_is_refused_for_the_verify_shape(engine):
    import eccoxide_amd as E

    lib = engine._lib
    z = ctypes.create_string_buffer(64)
    rc = lib.eccx_double_scalarmul(engine._ctx, 0, 1, bytes(32), bytes(32), bytes(64), z, z, 1 << 8)
    assert rc == -2
    assert E.engine.CT_SCAN == 1 << 8


# ---- the first table build must not disturb work still running on the caller's stream -------------
def test_first_fixed_base_call_waits_for_the_callers_stream(oracle):
    """a fresh context, an asynchronous 2^20 variable-base batch, then immediately the
    first fixed-base call (which builds the comb tables with the context's scratch slab and row
    buffer).  The ladder that was still running must come back intact."""
    import torch

    import eccoxide_amd as E

    n = 1 << 20
    ks = W.random_scalars("p256r1", n, seed=911)
    with E.Engine(0) as boot:           # bases from another context: `eng` stays without tables
        d_r = torch.from_numpy(W.random_scalars("p256r1", n, seed=912)).cuda()
        pts, _ = boot.scalarmul_base_t("p256r1", d_r)
        torch.cuda.synchronize()
    d_k = torch.from_numpy(ks).cuda()
    small = torch.from_numpy(W.random_scalars("p256r1", 512, seed=913)).cuda()
    with E.Engine(0) as eng:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            out, flags = eng.scalarmul_var_t("p256r1", d_k, pts, stream=s.cuda_stream)     # ~16 ms of GPU work
            b_out, b_flags = eng.scalarmul_base_t("p256r1", small, stream=s.cuda_stream)   # first use: builds
        torch.cuda.synchronize()
        sample = np.random.Generator(np.random.PCG64(3)).choice(n, size=2048, replace=False)
        sample.sort()
        t = torch.from_numpy(sample).cuda()
        want = oracle.var("p256r1", ks[sample].tobytes(), pts[t].cpu().numpy().tobytes(), threads=16)
        assert out[t].cpu().numpy().tobytes() == want[0]
        assert flags[t].cpu().numpy().tobytes() == want[1]
        want_b = oracle.base("p256r1", small.cpu().numpy().tobytes(), threads=16)
        assert b_out.cpu().numpy().tobytes() == want_b[0]
        # and the whole batch again, now that nothing overlaps
        out2, _ = eng.scalarmul_var_t("p256r1", d_k, pts)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)


# ---- eccx_prepare + eccx_reserve: afterwards a _dev call is only an enqueue -----------------------
def test_prepared_dev_calls_do_not_wait_for_the_stream(oracle):
    """After prepare + reserve, scalarmul_var_dev / scalarmul_base_dev / double_scalarmul_dev on a
    stream that is busy for ~50 ms return in well under that (no synchronisation, allocation or
    table build inside the call), the context's memory does not change, and the results are right."""
    import torch

    import eccoxide_amd as E

    n = 1 << 18
    with E.Engine(0) as eng:
        assert eng.device_bytes() == 0
        eng.prepare("p256r1", base=True)
        eng.reserve("p256r1", n, var=True)
        owned = eng.device_bytes()
        assert owned > 134_000_000            # the 16-bit-window table alone
        d_k = torch.from_numpy(W.random_scalars("p256r1", n, seed=921)).cuda()
        d_r = torch.from_numpy(W.random_scalars("p256r1", n, seed=922)).cuda()
        d_u = torch.from_numpy(W.random_scalars("p256r1", n, seed=923)).cuda()
        out_b = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
        out_v = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
        out_d = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
        fl = [torch.empty((n,), dtype=torch.uint8, device="cuda") for _ in range(3)]
        s = torch.cuda.Stream()
        a = torch.randn(8192, 8192, device="cuda")
        (a @ a).sum().item()                              # rocBLAS initialised outside the measurement
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for _ in range(12):                           # >= 50 ms of work queued ahead on the stream
                a = (a @ a) * 1e-4
            t0 = time.perf_counter()
            eng.scalarmul_base_t("p256r1", d_r, out_b, fl[0], stream=s.cuda_stream)
            eng.scalarmul_var_t("p256r1", d_k, out_b, out_v, fl[1], stream=s.cuda_stream)
            eng.double_scalarmul_t("p256r1", d_u, d_k, out_b, out_d, fl[2], stream=s.cuda_stream)
            dt = time.perf_counter() - t0
            busy = not s.query()
        torch.cuda.synchronize()
        assert busy, "the stream drained before the calls returned: the test did not test anything"
        assert dt < 0.02, f"three prepared _dev calls took {dt * 1e3:.1f} ms on a busy stream"
        assert eng.device_bytes() == owned
        idx = torch.arange(0, n, 997, device="cuda")
        s_r, s_k = d_r[idx].cpu().numpy().tobytes(), d_k[idx].cpu().numpy().tobytes()
        want_b = oracle.base("p256r1", s_r, threads=16)
        assert out_b[idx].cpu().numpy().tobytes() == want_b[0]
        want_v = oracle.var("p256r1", s_k, want_b[0], threads=16)
        assert out_v[idx].cpu().numpy().tobytes() == want_v[0]
        # u*G + k*(r*G) = (u + k*r)*G
        order = W.order("p256r1")
        s_u = d_u[idx].cpu().numpy().tobytes()
        m = len(idx)
        comb = b"".join(((int.from_bytes(s_u[i * 32:(i + 1) * 32], "big")
                          + int.from_bytes(s_k[i * 32:(i + 1) * 32], "big") * int.from_bytes(s_r[i * 32:(i + 1) * 32], "big")) % order
                         ).to_bytes(32, "big") for i in range(m))
        assert out_d[idx].cpu().numpy().tobytes() == oracle.base("p256r1", comb, threads=16)[0]


def test_last_error_is_per_thread_copy(engine):
    """eccx_last_error hands out a copy: the pointer stays valid whatever other threads do."""
    import threading

    lib = engine._lib
    errs = []

    def worker():
        for _ in range(200):
            errs.append(lib.eccx_last_error(engine._ctx))

    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(isinstance(e, bytes) for e in errs)


# ---- sharded entry points over two contexts (one GPU here; one per GPU on a node) -------------------
@pytest.mark.parametrize("curve,n", [("p256r1", 1501), ("p384r1", 777), ("ed25519", 1023)])
def test_sharded_entry_points_two_contexts_ragged(oracle, curve, n):
    """eccx_scalarmul_{var,base}_sharded with two contexts on device 0 and a batch size that does
    not divide: contiguous shards [n*g/2, n*(g+1)/2), one host thread per context, results land in
    the caller's buffers in order."""
    import eccoxide_amd as E

    fb, sb = _SIZES[curve]
    cid = E.curve_id(curve)
    ks = W.random_scalars(curve, n, seed=931).tobytes()
    pts = _bases(oracle, curve, n, seed=932)
    want_v = oracle.var(curve, ks, pts, threads=16)
    want_b = oracle.base(curve, ks, threads=16)
    with E.Engine(0) as e1, E.Engine(0) as e2:
        lib = e1._lib
        arr = (ctypes.c_void_p * 2)(e1._ctx.value, e2._ctx.value)
        out = ctypes.create_string_buffer(n * 2 * fb)
        fl = ctypes.create_string_buffer(n)
        assert lib.eccx_scalarmul_var_sharded(arr, 2, cid, n, ks, pts, out, fl, 0) == 0
        assert out.raw == want_v[0] and fl.raw == want_v[1]
        out = ctypes.create_string_buffer(n * 2 * fb)
        fl = ctypes.create_string_buffer(n)
        assert lib.eccx_scalarmul_base_sharded(arr, 2, cid, n, ks, out, fl, 0) == 0
        assert out.raw == want_b[0] and fl.raw == want_b[1]
        # n smaller than the number of contexts: one shard is empty
        out = ctypes.create_string_buffer(2 * fb)
        fl = ctypes.create_string_buffer(1)
        assert lib.eccx_scalarmul_var_sharded(arr, 2, cid, 1, ks[:sb], pts[:2 * fb], out, fl, 0) == 0
        assert out.raw == want_v[0][:2 * fb]


# ---- the RCCL path on one GPU -----------------------------------------------------------------------
def test_bench_with_forced_rccl_gather():
    """ECCX_FORCE_DIST=1: bench.py initialises the nccl (= RCCL) process group at world size 1 and
    runs GatherPipeline's asynchronous gathers on RCCL's stream, two batches in flight; the parity
    sample is taken from the GATHERED buffers.  Proves communicator init, the gather and its
    ordering against the engine's kernels on real hardware (an 8-GPU run adds the peers, not code)."""
    env = dict(os.environ, ECCX_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["parity_sample_ok"] is True
    assert "forced" in line["config"]["gather"]
    assert line["value"] > 1e7


def test_gather_pipeline_on_rccl_world_of_one(oracle):
    """GatherPipeline(force=True) inside this process: results of the engine's device-tensor API
    gathered by RCCL into the root's contiguous area while the next batch computes."""
    import torch
    import torch.distributed as dist

    import eccoxide_amd as E
    from eccoxide_amd.dist import GatherPipeline

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 1 << 14
        dev = torch.device("cuda", 0)
        with E.Engine(0) as eng:
            ks = [torch.from_numpy(W.random_scalars("p256r1", n, seed=940 + i)).to(dev) for i in range(3)]
            pts, _ = eng.scalarmul_base_t("p256r1", ks[0])
            outs = [torch.empty((n, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
            fls = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(2)]
            pipe = GatherPipeline(n, 64, dev, slots=2, force=True)
            assert pipe.active
            got = []
            for i in range(3):
                slot = i & 1
                pipe.finish(slot)
                if i >= 2:
                    got.append(pipe.result(slot)[0].clone())
                eng.scalarmul_var_t("p256r1", ks[i], pts, outs[slot], fls[slot])
                pipe.start(slot, outs[slot], fls[slot])
            pipe.finish(0)
            pipe.finish(1)
            torch.cuda.synchronize()
            res = {0: got[0], 1: pipe.result(1)[0], 2: pipe.result(0)[0]}
            for i in range(3):
                want = oracle.var("p256r1", ks[i][:128].cpu().numpy().tobytes(), pts[:128].cpu().numpy().tobytes())[0]
                assert res[i][:128].cpu().numpy().tobytes() == want, i
                assert res[i].data_ptr() not in (outs[0].data_ptr(), outs[1].data_ptr())   # really the gathered copy
            # the one-shot form a multi-GPU host calls, on the real engine
            from eccoxide_amd.dist import engine_compute, sharded_scalarmul

            o, f = sharded_scalarmul(engine_compute(eng, "p256r1"), ks[1], pts)
            torch.cuda.synchronize()
            assert torch.equal(o, res[1]) and int(f.max()) == 0
    finally:
        dist.destroy_process_group()


# ---- BASELINE.json configs[2], literally: the table in LDS at the full batch ------------------------
def test_full_size_ed25519_mul_base_table_in_lds(engine, oracle):
    """Ed25519 mul_base with the comb table in LDS (ECCX_TABLE_IN_LDS) at batch 2^20: identical to
    the default path on every unit, and to the oracle on a sample."""
    import torch

    n = 1 << 20
    ks = W.random_scalars("ed25519", n, seed=2000)
    d_k = torch.from_numpy(ks).cuda()
    out, flags = engine.scalarmul_base_t("ed25519", d_k)
    out_l, flags_l = engine.scalarmul_base_t("ed25519", d_k, table_in_lds=True)
    torch.cuda.synchronize()
    assert torch.equal(out, out_l) and torch.equal(flags, flags_l)
    sample = np.random.Generator(np.random.PCG64(7)).choice(n, size=1024, replace=False)
    sample.sort()
    want_out, want_inf, _ = oracle.base("ed25519", ks[sample].tobytes(), threads=16)
    t = torch.from_numpy(sample).cuda()
    assert out_l[t].cpu().numpy().tobytes() == want_out and flags_l[t].cpu().numpy().tobytes() == want_inf


# ---- BLS12-381 G1 through the endomorphism --------------------------------------------------------
def test_bls_assume_subgroup_ladder_matches_oracle(engine, oracle):
    """ECCX_ASSUME_SUBGROUP: k = k1 + k2 x^2, k P = k1 P + k2 (beta X, -Y) for P in G1 -- same bytes as
    the reference algorithm for random scalars and for the scalars that stress the split: 0, 1,
    x^2 - 1, x^2, x^2 + 1 (k1 = 0 / k2 = 1 boundaries), r - 1, r, r + 1, 2^255, 2^256 - 1 (largest
    k2), and multiples of x^2 whose digits make the two halves meet (k1 P = +-k2 sigma'(P) cases come
    from k = j (x^2 + 1): k1 = k2 = j)."""
    from oracle import ecc_ref as R

    curve = "bls12_381_g1"
    x2 = R.BLS_X_ABS ** 2
    r = W.order(curve)
    special = [0, 1, 2, x2 - 1, x2, x2 + 1, 2 * x2, r - 1, r, r + 1, 1 << 255, (1 << 256) - 1,
               ((1 << 256) - 1) // x2 * x2, 5 * (x2 + 1), 16 * (x2 + 1), 17 * (x2 + 1), 31 * x2 + 31,
               (1 << 128) - 1, 1 << 127, (1 << 129) * x2 % (1 << 256)]
    n = 1500
    ks = bytearray(W.random_scalars(curve, n, seed=951).tobytes())
    for i, v in enumerate(special):
        ks[i * 32:(i + 1) * 32] = (v % (1 << 256)).to_bytes(32, "big")
    # some full-range scalars (above r, any 256-bit string is accepted)
    rng = np.random.Generator(np.random.PCG64(952))
    for i in range(len(special), len(special) + 200):
        ks[i * 32:(i + 1) * 32] = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
    ks = bytes(ks)
    pts = _bases(oracle, curve, n, seed=953)
    want = oracle.var(curve, ks, pts, threads=16)
    got = engine.scalarmul_var(curve, ks, pts, assume_subgroup=True)
    assert got[1] == want[1]
    assert got[0] == want[0]
    assert engine.scalarmul_var(curve, ks, pts) == got           # and the default ladder
    # the flag is accepted and ignored on curves without the endomorphism path
    k256 = W.random_scalars("p256r1", 64, seed=954).tobytes()
    p256 = _bases(oracle, "p256r1", 64, seed=955)
    assert engine.scalarmul_var("p256r1", k256, p256, assume_subgroup=True) == engine.scalarmul_var("p256r1", k256, p256)


def test_bls_assume_subgroup_full_size(engine, oracle):
    """BASELINE.json configs[4] at its single-GPU size through the endomorphism ladder: identical to
    the default ladder on all 2^20 units (bases r_i G are in G1), sampled against the oracle."""
    import torch

    curve = "bls12_381_g1"
    n = 1 << 20
    ks = W.random_scalars(curve, n, seed=961)
    d_k = torch.from_numpy(ks).cuda()
    d_r = torch.from_numpy(W.random_scalars(curve, n, seed=962)).cuda()
    pts, _ = engine.scalarmul_base_t(curve, d_r)
    out, flags = engine.scalarmul_var_t(curve, d_k, pts)
    out_g, flags_g = engine.scalarmul_var_t(curve, d_k, pts, assume_subgroup=True)
    torch.cuda.synchronize()
    assert torch.equal(out, out_g) and torch.equal(flags, flags_g)
    sample = np.random.Generator(np.random.PCG64(9)).choice(n, size=512, replace=False)
    sample.sort()
    t = torch.from_numpy(sample).cuda()
    want = oracle.var(curve, ks[sample].tobytes(), pts[t].cpu().numpy().tobytes(), threads=16)
    assert out_g[t].cpu().numpy().tobytes() == want[0] and flags_g[t].cpu().numpy().tobytes() == want[1]


def test_bls_subgroup_test_on_cofactor_points(engine, oracle):
    """The endomorphism test (g1.rs:90-109) on the points that make its ladders degenerate: points of
    small order (3, 11, 33: multiples hit the point at infinity and P itself inside [|x|]P), points
    of the full cofactor subgroup, G1 points, and sums of both -- against [r]P == infinity computed
    with textbook affine arithmetic."""
    from oracle import ecc_ref as R

    c = R.BLS12_381_G1
    x = -R.BLS_X_ABS
    h = (x - 1) ** 2 // 3
    assert (h * c.n - (c.p + 1 - (x + 1))) == 0      # #E(Fp) = p + 1 - t with t = x + 1
    pts = []
    xx = 1
    raw = []
    while len(raw) < 12:                             # arbitrary curve points (order h * r up to factors)
        xx += 1
        P = R.ref_w_decompress_xy(c, xx, bool(xx & 1))
        if P is not None:
            raw.append(P)
    G = (c.gx, c.gy)
    for P in raw:
        T = R.affine_mul(c, c.n, P)                  # in the cofactor subgroup
        pts.append(P)
        if T is not None:
            pts.append(T)
            for f in (3, 11, 33):
                S = R.affine_mul(c, h // f, T) if h % f == 0 else None
                if S is not None:
                    pts.append(S)                    # order divides f
            pts.append(R.affine_add(c, T, R.affine_mul(c, 12345, G)))   # mixed: not in G1
    pts += [R.affine_mul(c, k, G) for k in (1, 2, 3, c.n - 1, 0xDEADBEEF)]
    small = [P for P in pts if R.affine_mul(c, 33, P) is None]
    assert small, "no small-order point was generated"
    enc = b"".join(R.ref_g1_to_uncompressed(P) for P in pts)
    want_in = [R.affine_mul(c, c.n, P) is None for P in pts]
    assert any(want_in) and not all(want_in)
    assert want_in == [R.ref_g1_is_in_subgroup(P) for P in pts]
    out, flags = engine.point_decompress("bls12_381_g1", enc, uncompressed=True, check_subgroup=True)
    assert list(flags) == [0 if w else 2 for w in want_in]
    for i, (P, w) in enumerate(zip(pts, want_in)):
        rec = out[96 * i:96 * (i + 1)]
        assert rec == (P[0].to_bytes(48, "big") + P[1].to_bytes(48, "big") if w else bytes(96))


def test_bls_low_order_bases_among_ordinary_ones(engine, oracle):
    """The co-Z ladder marks units whose base point has order <= 16 (its table build degenerates) and
    the generic ladder redoes exactly those: points of order 3, 11, 33 and other cofactor-subgroup
    points interleaved with G1 points inside the same wavefronts -- every unit against the oracle."""
    from oracle import ecc_ref as R

    c = R.BLS12_381_G1
    x = -R.BLS_X_ABS
    h = (x - 1) ** 2 // 3
    odd = []
    xx = 1
    while len(odd) < 6:
        xx += 1
        P = R.ref_w_decompress_xy(c, xx, bool(xx & 1))
        if P is None:
            continue
        T = R.affine_mul(c, c.n, P)
        if T is None:
            continue
        odd.append(T)                                   # full cofactor order
        for f in (3, 11, 33):
            if h % f == 0:
                S = R.affine_mul(c, h // f, T)
                if S is not None:
                    odd.append(S)
    assert any(R.affine_mul(c, 3, P) is None for P in odd), "no point of order 3 among the test points"
    n = 640
    pts = bytearray(_bases(oracle, "bls12_381_g1", n, seed=971))
    where = list(range(5, n, 37))[: len(odd)]
    for i, P in zip(where, odd):
        pts[i * 96:(i + 1) * 96] = P[0].to_bytes(48, "big") + P[1].to_bytes(48, "big")
    pts = bytes(pts)
    ks = bytearray(W.random_scalars("bls12_381_g1", n, seed=972).tobytes())
    for j, i in enumerate(where):                        # small and structured scalars on the odd points too
        if j % 3 == 0:
            ks[i * 32:(i + 1) * 32] = (3 * 11 * (j + 1)).to_bytes(32, "big")
    ks = bytes(ks)
    want = oracle.var("bls12_381_g1", ks, pts, threads=16)
    got = engine.scalarmul_var("bls12_381_g1", ks, pts)
    assert got[1] == want[1] and got[0] == want[0]
    assert 1 in want[1]                                  # some multiples really are the point at infinity
    # under ECCX_ASSUME_SUBGROUP the G1 units are still right (the others are the caller's problem)
    got_g = engine.scalarmul_var("bls12_381_g1", ks, pts, assume_subgroup=True)
    for i in range(n):
        if i not in where:
            assert got_g[0][i * 96:(i + 1) * 96] == want[0][i * 96:(i + 1) * 96] and got_g[1][i] == want[1][i]


@pytest.mark.parametrize("curve", ["p256r1", "p384r1", "p521r1"])
def test_garbage_bases_do_not_disturb_their_neighbours(engine, oracle, curve):
    """The prime-order curves have no point of order <= 16, so the affine-table ladder's fix-up pass
    only ever sees bytes that are not a curve point (callers that skip ECCX_VALIDATE_POINTS): y = 0 makes
    the very first doubling degenerate, x = y = 0 every step.  Such units get SOME result (flag 0 or 1,
    never the internal marker), every other unit of the same wavefronts the oracle's; with validation
    they are rejected and nothing else changes."""
    fb, sb = _SIZES[curve]
    n = 700
    pts = bytearray(_bases(oracle, curve, n, seed=1201))
    bad = list(range(3, n, 61))
    for j, i in enumerate(bad):
        if j % 2 == 0:
            pts[i * 2 * fb + fb:(i + 1) * 2 * fb] = bytes(fb)           # (x, 0)
        else:
            pts[i * 2 * fb:(i + 1) * 2 * fb] = bytes(2 * fb)            # (0, 0)
    pts = bytes(pts)
    ks = W.random_scalars(curve, n, seed=1202).tobytes()
    good_pts = bytearray(pts)
    for i in bad:                                                       # the oracle gets a valid stand-in there
        good_pts[i * 2 * fb:(i + 1) * 2 * fb] = pts[0:2 * fb]
    want = oracle.var(curve, ks, bytes(good_pts), threads=16)
    got = engine.scalarmul_var(curve, ks, pts)
    chk = engine.scalarmul_var(curve, ks, pts, validate=True)
    for i in range(n):
        if i in bad:
            assert got[1][i] in (0, 1)
            assert chk[1][i] == 2 and chk[0][i * 2 * fb:(i + 1) * 2 * fb] == bytes(2 * fb)
        else:
            for r in (got, chk):
                assert r[0][i * 2 * fb:(i + 1) * 2 * fb] == want[0][i * 2 * fb:(i + 1) * 2 * fb] and r[1][i] == want[1][i]


# ---- host-buffer entry points: nothing allocated or freed after warm-up ------------------------------
def test_host_buffer_calls_allocate_nothing_after_warm_up(oracle):
    """What the Rust crate calls (rust/eccoxide-gpu/src/weierstrass.rs: the host-buffer forms): device-side copies of
    the arguments live in context-owned, grow-only buffers, so a steady stream of host calls neither allocates nor
    frees device memory -- eccx_device_bytes stays put across 100 mixed calls -- and still returns the oracle's bytes."""
    import eccoxide_amd as E

    curve, n = "p256r1", 2048
    ks = W.random_scalars(curve, n, seed=991).tobytes()
    with E.Engine(0) as eng:
        eng.prepare(curve, base=True, ct=True)
        eng.reserve(curve, n, var=True, ct=True, host=True)
        pts = eng.scalarmul_base(curve, ks)
        # one call of each kind: whatever grows, grows now
        eng.scalarmul_var(curve, ks, pts[0])
        eng.scalarmul_var(curve, ks, pts[0], ct_scan=True)
        eng.scalarmul_base(curve, ks, ct_scan=True)
        eng.point_add(curve, pts[0], pts[0])
        eng.double_scalarmul(curve, ks, ks, pts[0])
        enc = eng.point_compress(curve, pts[0], pts[1])
        eng.point_decompress(curve, enc)
        before = eng.device_bytes()
        want_v = oracle.var(curve, ks, pts[0], threads=16)
        for i in range(100):
            m = n - (i % 7) * 13
            if i % 5 == 0:
                got = eng.scalarmul_var(curve, ks[:m * 32], pts[0][:m * 64])
                assert got[0] == want_v[0][:m * 64]
            elif i % 5 == 1:
                assert eng.scalarmul_base(curve, ks[:m * 32])[0] == pts[0][:m * 64]
            elif i % 5 == 2:
                assert eng.scalarmul_var(curve, ks[:m * 32], pts[0][:m * 64], ct_scan=True)[0] == want_v[0][:m * 64]
            elif i % 5 == 3:
                assert eng.point_decompress(curve, enc[:m * 33])[0] == pts[0][:m * 64]
            else:
                eng.point_add(curve, pts[0][:m * 64], pts[0][:m * 64])
            assert eng.device_bytes() == before, i
