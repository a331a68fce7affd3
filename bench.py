#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (BASELINE.json).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  At N=1 the workload is BASELINE.json configs[1]: p256r1 variable-base
scalar multiplication, batch = 2^20 (scalars uniform in [1, n), bases r_i*G).  For N>1 the
driver launches one process per GPU (torch.distributed.run); every rank runs the same
per-GPU batch (weak scaling), there is no data-path collective, and every step's result bytes
are gathered to rank 0 over RCCL -- asynchronously, while the next batch computes into a second
set of buffers; all gathers have completed before the timed region ends.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      algorithmic HBM bytes (160 B per p256 unit) / kernel time
                vs the 8 TB/s HBM peak -- the metric asks for it; the path is integer-VALU
                bound, so the fraction is tiny by construction
  valu          the roofline that actually binds: multiply-accumulates issued per second
                against the v_mad_u64_u32 issue peak measured on this chip
  cpu_baseline  the oracle (C restatement of the reference algorithm, kind "port") timed on
                the host cores over a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def _var_muls(nw, dbl, inv):
    """field mul+sqr the saturated (Jacobian, signed 5-bit window) variable-base path issues per
    unit: window table (1 doubling + 14 additions + Z^2, Z^3 per entry), ceil((4*nw + 1)/5)
    windows of 5 doublings + 1 addition (no doublings for the top window), input conversion,
    and the batched normalisation (one Fermat inversion per 8 units)."""
    nwin = (4 * nw + 1 + 4) // 5
    return (dbl + 14 * 14 + 15 * 2) + nwin * 14 + (nwin - 1) * 5 * dbl + 2 + (inv + 7) // 8 + 9


def _sat(muls, pairs_per_mul):
    """multiplier instructions of a saturated-limb kernel: every limb product is the pair
    v_mad_u64_u32 + v_addc_co_u32 (fe.hpp)"""
    return {"pair": muls * pairs_per_mul, "mad": 0}


def _var_unsat(n, nz, sb, a0, inv, sat_pairs, mont=True, norm_u=8):
    """default variable-base path (kernels_unsat.hpp): n limbs of 28/29 bits, one
    v_mad_u64_u32 per limb product.  A product is n*n + n*nz mads (nz = non-zero reduction
    digits per Montgomery factor: those of p + 1 for P-256, the 4 signed terms of p + 1 for
    P-384, all n for BLS12-381,
    none for the Mersenne prime whose wrapped half shares the columns), a square n(n+1)/2 +
    n*nz.  Doubling = 4 products + 4 squares (a = -3) or 2 + 5 (a = 0), addition = 11 + 3;
    ceil((8*sb + 1)/5) signed windows (inv, sat_pairs and norm_u are kept for reference only)."""
    mul, sqr = n * n + n * nz, n * (n + 1) // 2 + n * nz
    nwin = (8 * sb + 1 + 4) // 5
    dbls, adds = 1 + (nwin - 1) * 5, 14 + nwin
    dm, ds = (2, 5) if a0 else (4, 4)
    conv = 5 if mont else 0                      # 2 products into the working form, 3 out of it
    n_mul = dm * dbls + 11 * adds + 15 + conv    # + Z^3 per table entry
    n_sqr = ds * dbls + 3 * adds + 15            # + Z^2 per table entry
    # the normalisation kernel works on the same unsaturated limbs: per unit a prefix product, the
    # 1/Z share, Z^-2, Z^-3, x and y (7 products + 1 square) and 2 conversions out; the inversion
    # itself is division steps, not multiplications
    n_mul += 9
    n_sqr += 1
    sat_pairs = 0
    norm = 0
    return {"mad": n_mul * mul + n_sqr * sqr, "pair": norm * sat_pairs}


WORKLOADS = {
    # name: (curve, op, per-GPU batch, algorithmic bytes per unit, multiplier instructions per unit)
    "p256r1_var_2^20": ("p256r1", "var", 1 << 20, 160, _var_unsat(9, 4, 32, 0, 383, 8 * 8 + 8 * 3, norm_u=16)),
    # fixed base, default path: 16-bit windows, 16 additions of 7 products (Edwards, 81 + 9 mads each)
    # or of 8 products + 3 squares (P-256) on unsaturated limbs, then the saturated normalisation
    "ed25519_base_2^20": ("ed25519", "base", 1 << 20, 96,
                          {"mad": (16 * 7 + 8) * (81 + 9), "pair": 0}),
    "p256r1_base_2^20": ("p256r1", "base", 1 << 20, 96,
                         {"mad": 16 * (8 * (81 + 36) + 3 * (45 + 36)) + 9 * (81 + 36) + (45 + 36), "pair": 0}),
    # unsaturated 9 x 29 ladder: per bit 5 products (81 + 9 mads), 4 squares (45 + 9), one small multiple (9 + 1)
    "p384r1_base_2^19": ("p384r1", "base", 1 << 19, 144,
                         {"mad": 24 * (8 * 252 + 3 * 161) + 9 * 252 + 161, "pair": 0}),
    "p521r1_base_2^19": ("p521r1", "base", 1 << 19, 198,
                         {"mad": 33 * (8 * 324 + 3 * 171) + 9 * 324 + 171, "pair": 0}),
    "bls12_381_g1_base_2^20": ("bls12_381_g1", "base", 1 << 20, 128,
                               {"mad": 16 * (8 * 392 + 3 * 301) + 9 * 392 + 301, "pair": 0}),
    "x25519_2^20": ("ed25519", "x25519", 1 << 20, 96,
                    {"mad": 256 * (5 * 90 + 4 * 54 + 10) + 6 * 90, "pair": 0}),
    # edwards25519 variable base: 52 signed windows of 4 x (4 squares + 3 products) + (4 + 4) + a
    # 7-product addition; table of 16 cached multiples (1 doubling, 14 additions of 8, 16 x 2d*T)
    "ed25519_var_2^20": ("ed25519", "var", 1 << 20, 160,
                         {"mad": (51 * 20 + 4) * 54 + (51 * 16 + 52 * 7 + 4 + 14 * 8 + 16 + 1 + 8) * 90, "pair": 0}),
    # verify shape u1*G + u2*Q: the variable-base ladder + 16 mixed additions (8 products + 3 squares)
    "p256r1_verify_2^20": ("p256r1", "dsm", 1 << 20, 192,
                           {"mad": _var_unsat(9, 4, 32, 0, 383, 88, norm_u=16)["mad"] + 16 * (8 * 117 + 3 * 81),
                            "pair": _var_unsat(9, 4, 32, 0, 383, 88, norm_u=16)["pair"]}),
    "p384r1_var_2^19": ("p384r1", "var", 1 << 19, 240, _var_unsat(14, 4, 48, 0, 575, 12 * 12 + 12 * 10)),
    "p521r1_var_2^19": ("p521r1", "var", 1 << 19, 330, _var_unsat(18, 0, 66, 0, 780, 17 * 17, mont=False)),
    "bls12_381_g1_var_2^20": ("bls12_381_g1", "var", 1 << 20, 224, _var_unsat(14, 14, 32, 1, 570, 2 * 12 * 12)),
}
# HBM bytes per launch measured with rocprofv3 PMC passes (tools/profile.sh; summaries under
# profiles/): FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for 16-byte-per-lane
# reads on gfx950, plus WRITE_SIZE, summed over the kernels of one step.  Counters cannot be
# read from inside this process, so the figure is the committed measurement of this very
# workload, not a live one; null where no profile has been taken.
MEASURED_TRAFFIC = {
    "p256r1_var_2^20": {"bytes": 2 * (6786535305 + 109929528) + 3698666166 + 70295648,
                        "fetch_raw": 6786535305 + 109929528, "write": 3698666166 + 70295648,
                        "source": "profiles/r01_p256r1_var_u29.json"},
    "ed25519_base_2^20": {"bytes": 2 * (1135026666 + 108984009) + 118721354 + 70266038,
                          "fetch_raw": 1135026666 + 108984009, "write": 118721354 + 70266038,
                          "source": "profiles/r01_ed25519_base.json"},
    "x25519_2^20": {"bytes": 2 * (31830262 + 106681508) + 118489737 + 36701348,
                    "fetch_raw": 31830262 + 106681508, "write": 118489737 + 36701348,
                    "source": "profiles/r01_x25519.json"},
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Integer-multiplier issue costs measured by tools/ubench/valu_rates.hip on MI355X
# (profiles/r01_valu_rates.jsonl): cycles one SIMD needs to issue the instruction for one
# wave.  A lone v_mad_u64_u32 (unsaturated limbs) costs 4.8-5.03; the saturated unit of work,
# the pair v_mad_u64_u32 + v_addc_co_u32, costs 8.62.  1024 SIMDs at 2.4 GHz.
CYC_MAD = 4.8
CYC_PAIR = 8.62
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="p256r1_var_2^20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 17)
    ap.add_argument("--variant", default="default", choices=["default", "mirror", "lds", "l2"],
                    help="default: fast kernels; mirror: reference-mirroring kernels; "
                         "lds: ed25519 fixed base with a signed 6-bit comb table resident in LDS; "
                         "l2: the reference's 4-bit comb read through L2")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched by hand without torch.distributed.run: start the ranks as a child job
        # (before anything in this process touches the GPU) and exit with its code
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    import eccoxide_amd as E
    from eccoxide_amd import workload as W
    from eccoxide_amd.dist import GatherPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    curve, op, n, alg_bytes, mult = WORKLOADS[args.workload]
    fb, sb = E.field_bytes(curve), E.scalar_bytes(curve)
    eng = E.Engine(dev.index)

    # synthetic inputs, resident in HBM before the timed region (every rank its own shard seed)
    ks = torch.from_numpy(W.random_scalars(curve, n, seed=10 + rank)).to(dev)
    ks2 = None
    if op in ("var", "dsm"):
        rs = torch.from_numpy(W.random_scalars(curve, n, seed=1000 + rank)).to(dev)
        pts, _ = eng.scalarmul_base_t(curve, rs)  # r_i * G: bases in the prime-order subgroup
        del rs
        if op == "dsm":
            ks2 = torch.from_numpy(W.random_scalars(curve, n, seed=2000 + rank)).to(dev)
    elif op == "x25519":
        rs = torch.from_numpy(W.random_scalars(curve, n, seed=1000 + rank)).to(dev)
        pts, _ = eng.x25519_t(rs)  # peer public keys X25519(r_i, 9)
        del rs
    else:
        eng.scalarmul_base_t(curve, ks[:256].contiguous())  # builds the comb table
        pts = None
    out_cols = 32 if op == "x25519" else 2 * fb
    # two sets of output buffers: with N > 1 the gather of batch i runs while batch i+1 computes
    outs = [torch.empty((n, out_cols), dtype=torch.uint8, device=dev) for _ in range(2)]
    flagss = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream(dev)
    pipe = GatherPipeline(n, out_cols, dev, slots=2)

    mirror = args.variant == "mirror"

    def step(slot):
        out, flags = outs[slot], flagss[slot]
        if op == "var":
            eng.scalarmul_var_t(curve, ks, pts, out, flags, stream=stream.cuda_stream, mirror=mirror)
        elif op == "dsm":
            eng.double_scalarmul_t(curve, ks, ks2, pts, out, flags, stream=stream.cuda_stream)
        elif op == "x25519":
            eng.x25519_t(ks, pts, out, flags, stream=stream.cuda_stream)
        else:
            eng.scalarmul_base_t(curve, ks, out, flags, stream=stream.cuda_stream, mirror=mirror,
                                 table_in_lds={"lds": True, "l2": False}.get(args.variant))

    def run(steps, events=None):
        """`steps` passes: compute batch i, start its gather, and only wait for a gather when
        its buffers are about to be reused; every gather has completed on return."""
        for i in range(steps):
            slot = i & 1
            pipe.finish(slot)            # the gather that last read this slot's buffers
            if events:
                events[i][0].record(stream)
            step(slot)
            if events:
                events[i][1].record(stream)
            pipe.start(slot, outs[slot], flagss[slot])
        pipe.finish(0)
        pipe.finish(1)

    run(args.warmup)
    torch.cuda.synchronize(dev)

    # kernel-only time of the dominant kernel, HIP events on the launch stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.steps, ev)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    last = (args.steps - 1) & 1
    out, flags = outs[last], flagss[last]

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # correctness gate on the timed buffers: sampled comparison with the oracle (rank 0)
    parity = None
    cpu = None
    if rank == 0:
        from tests import oracle_lib

        ora = oracle_lib.load()
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:256].sort().values.to(dev)
        s_k = ks[idx].cpu().numpy().tobytes()
        if op == "var":
            w_out, w_inf, _ = ora.var(curve, s_k, pts[idx].cpu().numpy().tobytes(), threads=8)
        elif op == "dsm":
            # u1*G and u2*Q from the C oracle, their sum with textbook affine arithmetic
            from oracle import ecc_ref as R

            c = R.CURVES[curve]
            a_out, a_inf, _ = ora.base(curve, s_k, threads=8)
            b_out, b_inf, _ = ora.var(curve, ks2[idx].cpu().numpy().tobytes(), pts[idx].cpu().numpy().tobytes(), threads=8)
            pb = 2 * fb
            dec = lambda buf, fl, i: None if fl[i] else (int.from_bytes(buf[i * pb:i * pb + fb], "big"),
                                                         int.from_bytes(buf[i * pb + fb:(i + 1) * pb], "big"))
            sums = [R.affine_add(c, dec(a_out, a_inf, i), dec(b_out, b_inf, i)) for i in range(len(a_inf))]
            w_out = b"".join(bytes(pb) if t is None else t[0].to_bytes(fb, "big") + t[1].to_bytes(fb, "big") for t in sums)
            w_inf = bytes(1 if t is None else 0 for t in sums)
        elif op == "x25519":
            w_out, w_inf = ora.x25519(s_k, pts[idx].cpu().numpy().tobytes(), threads=8)
        else:
            w_out, w_inf, _ = ora.base(curve, s_k, threads=8)
        parity = (out[idx].cpu().numpy().tobytes() == w_out) and (flags[idx].cpu().numpy().tobytes() == w_inf)
        if not args.no_cpu_baseline:
            cores = min(16, os.cpu_count() or 1)
            m = min(n, args.cpu_sample)
            c_k = ks[:m].cpu().numpy().tobytes()
            c_p = pts[:m].cpu().numpy().tobytes() if op in ("var", "x25519", "dsm") else None
            t1 = time.perf_counter()
            if op == "var":
                ora.var(curve, c_k, c_p, threads=cores)
            elif op == "dsm":   # the two scalar multiplications dominate; the final addition is not timed
                ora.base(curve, c_k, threads=cores)
                ora.var(curve, ks2[:m].cpu().numpy().tobytes(), c_p, threads=cores)
            elif op == "x25519":
                ora.x25519(c_k, c_p, threads=cores)
            else:
                ora.base(curve, c_k, threads=cores)
            dt = time.perf_counter() - t1
            cpu = {"value": m / dt, "unit": "scalarmuls/s", "cores": cores, "kind": "port",
                   "sample": f"first {m} units of the same {args.workload} batch, oracle/eccx_oracle.c "
                             f"(C restatement of the reference algorithm), {cores} threads, {dt:.2f} s"}

    if rank == 0:
        total_units = n * world * args.steps
        value = total_units / elapsed
        ach = alg_bytes * n / (kernel_ms * 1e-3) / 1e9
        # time the SIMDs must spend issuing the multiplier instructions of this batch (one
        # instruction serves the 64 units of a wave) against the kernel time
        issue_cycles = mult["mad"] * CYC_MAD + mult["pair"] * CYC_PAIR
        valu_frac = (n / 64) * issue_cycles / (kernel_ms * 1e-3 * SIMDS * CLOCK_HZ)
        mul_rate = (mult["mad"] + mult["pair"]) * n / (kernel_ms * 1e-3)
        line = {
            "metric": "variable-base scalarmuls/sec (batch) per GPU + achieved HBM GB/s vs roofline"
            if op == "var" else ("double-scalar u1*G + u2*Q /sec (batch) per GPU + achieved HBM GB/s vs roofline" if op == "dsm" else ("X25519 scalarmuls/sec (batch) per GPU + achieved HBM GB/s vs roofline" if op == "x25519"
                                 else "fixed-base scalarmuls/sec (batch) per GPU + achieved HBM GB/s vs roofline")),
            "value": value,
            "unit": "scalarmuls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": args.workload, "curve": curve, "op": op, "batch_per_gpu": n,
                       "global_batch": n * world, "parallelism": f"shard{world}" if world > 1 else "single",
                       "gather": "rccl gather to rank 0, overlapped with the next batch (all complete inside the timed region)" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "traffic": (MEASURED_TRAFFIC[args.workload]["bytes"]
                                     if args.workload in MEASURED_TRAFFIC and args.variant == "default" else None),
                         "traffic_detail": MEASURED_TRAFFIC.get(args.workload) if args.variant == "default" else None,
                         "kernel_ms": kernel_ms, "alg_bytes_per_unit": alg_bytes,
                         "alg_bytes_per_launch": alg_bytes * n,
                         "note": "integer-VALU bound path, see valu; traffic above the algorithmic bytes is "
                                 "the per-lane window table of the variable-base ladder / the random reads of the "
                                 "16-bit-window comb table of the fixed-base path (DESIGN.md §6)"},
            "valu": None if args.variant != "default" else {"bound": "integer multiplier issue (v_mad_u64_u32; + v_addc_co_u32 in saturated kernels)",
                     "achieved": mul_rate / 1e12, "unit": "T limb-products/s",
                     "frac": valu_frac,
                     "mads_per_unit": mult["mad"], "mad_addc_pairs_per_unit": mult["pair"],
                     "issue_cycles_per_wave": issue_cycles,
                     "note": "frac = share of SIMD issue time spent on multiplier instructions at the "
                             "measured issue cost (4.8 cycles per mad, 8.62 per mad+addc pair)"},
            "cpu_baseline": cpu,
            "parity_sample_ok": parity,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
