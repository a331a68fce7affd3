#!/usr/bin/env python3
"""Throughput of the point codecs (eccx_point_compress_dev / eccx_point_decompress_dev) and of the
group law (eccx_point_add_dev) on one GPU, inputs resident in HBM: one JSON line per curve and direction.

usage: python tools/bench_codec.py [--log2n 20] [--steps 10]
`achieved` = algorithmic bytes (encoding + x||y + flag byte) per launch / average launch time
measured with events on the launch stream; a sample of every timed output is checked against
the oracle (oracle/ecc_ref.py)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch

    import eccoxide_amd as E
    from eccoxide_amd import workload as W
    from oracle import ecc_ref as R

    dev = torch.device("cuda", 0)
    eng = E.Engine(0)
    n = 1 << args.log2n
    stream = torch.cuda.current_stream(dev)
    for curve in ["p256r1", "p384r1", "p521r1", "bls12_381_g1", "ed25519"]:
        fb, eb = E.field_bytes(curve), eng.compressed_bytes(curve)
        ks = torch.from_numpy(W.random_scalars(curve, n, seed=3)).to(dev)
        pts, fl = eng.scalarmul_base_t(curve, ks)
        inf = None if curve == "ed25519" else fl
        enc = eng.point_compress_t(curve, pts, inf)
        back, bfl = eng.point_decompress_t(curve, enc)
        torch.cuda.synchronize(dev)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:128].sort().values.to(dev)
        s_pts = pts[idx].cpu().numpy().tobytes()
        ok = enc[idx].cpu().numpy().tobytes() == R.point_compress_bytes(curve, s_pts) and torch.equal(back, pts) \
            and int(bfl.sum()) == 0
        other, ofl = eng.scalarmul_base_t(curve, torch.from_numpy(W.random_scalars(curve, n, seed=4)).to(dev))
        sums, sfl = eng.point_add_t(curve, pts, other)
        for name, fn in (("add", lambda: eng.point_add_t(curve, pts, other, sums, sfl)),
                         ("compress", lambda: eng.point_compress_t(curve, pts, inf, enc)),
                         ("decompress", lambda: eng.point_decompress_t(curve, enc, back, bfl)),
                         ("decompress+subgroup", (lambda: eng.point_decompress_t(curve, enc, back, bfl, check_subgroup=True))
                          if curve == "bls12_381_g1" else None)):
            if fn is None:
                continue
            fn()
            torch.cuda.synchronize(dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(stream)
                fn()
                b.record(stream)
            torch.cuda.synchronize(dev)
            ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
            alg = 6 * fb + 1 if name == "add" else eb + 2 * fb + 1
            print(json.dumps({"metric": f"{curve} point {name}", "value": n / (ms * 1e-3), "unit": "points/s", "n": n,
                              "kernel_ms": ms, "alg_bytes_per_unit": alg,
                              "roofline": {"bound": "hbm", "achieved": alg * n / (ms * 1e-3) / 1e9, "peak": 8000.0,
                                           "unit": "GB/s", "frac": alg * n / (ms * 1e-3) / 1e9 / 8000.0},
                              "parity_sample_ok": bool(ok)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
