#!/usr/bin/env python3
"""Differential soak on the GPU: the default kernels (Jacobian / extended coordinates, signed
windows, wide combs, fused double-scalar) against the reference-mirroring kernels (complete
formulas in homogeneous coordinates, 4-bit windows, 4-bit comb and group law on saturated
canonical limbs) on random batches of random sizes, all curves.

    python tools/soak.py [seconds] [seed]

Scalars mix uniform values with sparse / edge patterns (few set bits, all-ones runs, values around
the group order) to provoke the special cases of the Jacobian ladder and the combs.  Exits non-zero
on the first mismatch.  Not part of the test suite (it runs for minutes); no oracle involved.
"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import eccoxide_amd as E
from eccoxide_amd import workload as W

CURVES = ["p256r1", "p384r1", "p521r1", "bls12_381_g1", "ed25519"]


def scalars(curve, n, rng):
    sb = E.scalar_bytes(curve)
    order = W.order(curve)
    top = 8 * sb if curve != "p521r1" else 521
    out = []
    for _ in range(n):
        r = rng.random()
        if r < 0.5:
            k = rng.randrange(1 << top)
        elif r < 0.65:                       # sparse
            k = 0
            for _ in range(rng.randrange(1, 6)):
                k |= 1 << rng.randrange(top)
        elif r < 0.8:                        # runs of ones / zeros at window boundaries
            k = ((1 << rng.randrange(1, top)) - 1) << rng.randrange(0, 40)
            k &= (1 << top) - 1
        elif r < 0.9:                        # around the order and its multiples
            k = (order * rng.randrange(0, 3) + rng.randrange(-40, 40)) % (1 << top)
        else:                                # single 16-bit window digits (comb entries)
            k = rng.randrange(1, 1 << 16) << (16 * rng.randrange(0, (top + 15) // 16))
            k &= (1 << top) - 1
        out.append(k.to_bytes(sb, "big"))
    return torch.frombuffer(bytearray(b"".join(out)), dtype=torch.uint8).reshape(n, sb)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260401)
    dev = torch.device("cuda", 0)
    eng = E.Engine(0)
    t0 = last = time.time()
    rounds = units = 0
    while time.time() - t0 < budget:
        curve = rng.choice(CURVES)
        n = rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 4097, rng.randrange(1, 20000)])
        k1 = scalars(curve, n, rng).to(dev)
        k2 = scalars(curve, n, rng).to(dev)
        # bases: multiples of G (fixed-base path, default vs mirror), then reused as variable bases
        b_def, f_def = eng.scalarmul_base_t(curve, k1)
        b_mir, f_mir = eng.scalarmul_base_t(curve, k1, mirror=True)
        assert torch.equal(b_def, b_mir) and torch.equal(f_def, f_mir), ("base", curve, n, rounds)
        if curve != "ed25519":
            # replace points at infinity (no affine form) by the generator before using them as bases
            inf = f_def != 0
            if bool(inf.any()):
                g, _ = eng.scalarmul_base_t(curve, torch.frombuffer(bytearray((1).to_bytes(E.scalar_bytes(curve), "big")), dtype=torch.uint8).reshape(1, -1).to(dev))
                b_def[inf] = g[0]
        v_def, vf_def = eng.scalarmul_var_t(curve, k2, b_def)
        v_mir, vf_mir = eng.scalarmul_var_t(curve, k2, b_def, mirror=True)
        assert torch.equal(v_def, v_mir) and torch.equal(vf_def, vf_mir), ("var", curve, n, rounds)
        if curve == "bls12_381_g1":
            # the bases are multiples of G, so the endomorphism ladder must agree with the co-Z ladder
            v_glv, vf_glv = eng.scalarmul_var_t(curve, k2, b_def, assume_subgroup=True)
            assert torch.equal(v_glv, v_def) and torch.equal(vf_glv, vf_def), ("glv", curve, n, rounds)
        if rounds % 3 == 0:
            # ECCX_CT_SCAN: the secret-scalar kernels (scanning ladder, scanning comb, lane-gather comb), and every
            # sixth round the reference-mirroring kernels with the scan
            v_ct, vf_ct = eng.scalarmul_var_t(curve, k2, b_def, ct_scan=True)
            assert torch.equal(v_ct, v_def) and torch.equal(vf_ct, vf_def), ("ct var", curve, n, rounds)
            if curve == "bls12_381_g1":  # bases in G1: the endomorphism ladder in secret-scalar form
                v_cs, vf_cs = eng.scalarmul_var_t(curve, k2, b_def, ct_scan=True, assume_subgroup=True)
                assert torch.equal(v_cs, v_def) and torch.equal(vf_cs, vf_def), ("ct subgroup var", curve, n, rounds)
            b_ct, f_ct = eng.scalarmul_base_t(curve, k1, ct_scan=True)
            assert torch.equal(b_ct, b_mir) and torch.equal(f_ct, f_mir), ("ct base", curve, n, rounds)
            b_cg, f_cg = eng.scalarmul_base_t(curve, k1, ct_gather=True)
            assert torch.equal(b_cg, b_mir) and torch.equal(f_cg, f_mir), ("ct gather base", curve, n, rounds)
            if rounds % 6 == 0:
                v_cm, vf_cm = eng.scalarmul_var_t(curve, k2, b_def, ct_scan=True, mirror=True)
                assert torch.equal(v_cm, v_def) and torch.equal(vf_cm, vf_def), ("ct mirror var", curve, n, rounds)
                b_cm, f_cm = eng.scalarmul_base_t(curve, k1, ct_scan=True, mirror=True)
                assert torch.equal(b_cm, b_mir) and torch.equal(f_cm, f_mir), ("ct mirror base", curve, n, rounds)
        # fused double-scalar against comb + ladder + complete addition
        for subtract in (False, True):
            d_out, d_fl = eng.double_scalarmul_t(curve, k1, k2, b_def, subtract=subtract)
            # edwards25519: the neutral element is an ordinary affine point, no infinity flags
            a_inf = None if curve == "ed25519" else f_mir.cpu().numpy().tobytes()
            b_inf = None if curve == "ed25519" else vf_mir.cpu().numpy().tobytes()
            s_out, s_fl = eng.point_add(curve, b_mir.cpu().numpy().tobytes(), v_mir.cpu().numpy().tobytes(),
                                        a_inf=a_inf, b_inf=b_inf, subtract=subtract, mirror=True)
            assert d_out.cpu().numpy().tobytes() == s_out and d_fl.cpu().numpy().tobytes() == s_fl, ("dsm", curve, n, subtract, rounds)
            # the default (unsaturated) group law against the saturated one
            u_out, u_fl = eng.point_add(curve, b_mir.cpu().numpy().tobytes(), v_mir.cpu().numpy().tobytes(),
                                        a_inf=a_inf, b_inf=b_inf, subtract=subtract)
            assert (u_out, u_fl) == (s_out, s_fl), ("point_add", curve, n, subtract, rounds)
        # wire formats: compress -> decompress gives the points back; the other sign gives the
        # negated point (their sum is the neutral element), except where the flipped encoding is
        # not canonical (edwards25519 x = 0) and must be rejected
        ed = curve == "ed25519"
        enc = eng.point_compress_t(curve, v_def, None if ed else vf_def)
        back, bfl = eng.point_decompress_t(curve, enc)
        assert torch.equal(back, v_def) and torch.equal(bfl, torch.zeros_like(vf_def) if ed else vf_def), ("codec", curve, n, rounds)
        flip = enc.clone()
        if ed:
            flip[:, 31] ^= 0x80
        elif curve == "bls12_381_g1":
            flip[:, 0] ^= 0x20
        else:
            flip[:, 0] ^= 0x01
        neg, nfl = eng.point_decompress_t(curve, flip)
        if os.environ.get("SOAK_DEBUG") and n <= 4:
            print("debug", curve, n, "points", v_def.cpu().numpy().tobytes().hex(), "flags", vf_def.cpu().tolist(),
                  "enc", enc.cpu().numpy().tobytes().hex(), "flip flags", nfl.cpu().tolist(), flush=True)
        fb = E.field_bytes(curve)
        if ed:
            x_zero = ~(v_def[:, :fb] != 0).any(dim=1)
            assert torch.equal(nfl == 2, x_zero) and not bool((nfl == 1).any()), ("codec flip", curve, n, rounds)
            keep = (~x_zero).cpu()
        else:
            was_inf = vf_def != 0   # 0x00.. ^ 1 / 0xC0.. ^ 0x20 are not encodings
            assert torch.equal(nfl == 2, was_inf) and not bool((nfl == 1).any()), ("codec flip", curve, n, rounds)
            keep = (~was_inf).cpu()
        if bool(keep.any()):
            a_b = v_def.cpu()[keep].contiguous().numpy().tobytes()
            n_b = neg.cpu()[keep].contiguous().numpy().tobytes()
            _, zfl = eng.point_add(curve, a_b, n_b)
            assert zfl == b"\x01" * int(keep.sum()), ("codec negation", curve, n, rounds)
        # validation: corrupt some bases (off-curve / non-canonical) and require identical rejection
        # (flag 2, zero bytes) from both stacks and from the fused kernel
        if rounds % 4 == 0:
            bad = b_def.clone()
            sel = torch.rand(n, device=dev) < 0.3
            bad[sel, rng.randrange(bad.shape[1])] ^= 1 << rng.randrange(8)
            if rng.random() < 0.5:
                bad[sel & (torch.rand(n, device=dev) < 0.3)] = 0xFF   # non-canonical coordinates
            o1, f1 = eng.scalarmul_var_t(curve, k2, bad, validate=True)
            o2, f2 = eng.scalarmul_var_t(curve, k2, bad, validate=True, mirror=True)
            if not (torch.equal(o1, o2) and torch.equal(f1, f2)):
                diff = ((o1 != o2).any(dim=1) | (f1 != f2)).nonzero().flatten().cpu().tolist()
                for i in diff[:4]:
                    print("validate mismatch", curve, "unit", i, "scalar", k2[i].cpu().numpy().tobytes().hex(),
                          "base", bad[i].cpu().numpy().tobytes().hex(), "default", o1[i].cpu().numpy().tobytes().hex(), int(f1[i]),
                          "mirror", o2[i].cpu().numpy().tobytes().hex(), int(f2[i]), flush=True)
            assert torch.equal(o1, o2) and torch.equal(f1, f2), ("validate", curve, n, rounds)
            # without validation the corrupted units are garbage in, garbage out -- but only they:
            # every untouched unit must come out as before from both stacks
            o4, f4 = eng.scalarmul_var_t(curve, k2, bad)
            o5, f5 = eng.scalarmul_var_t(curve, k2, bad, mirror=True)
            keep = ~sel
            assert torch.equal(o4[keep], v_def[keep]) and torch.equal(f4[keep], vf_def[keep]), ("garbage neighbours", curve, n, rounds)
            assert torch.equal(o5[keep], v_def[keep]) and torch.equal(f5[keep], vf_def[keep]), ("garbage neighbours mirror", curve, n, rounds)
            o3, f3 = eng.double_scalarmul_t(curve, torch.zeros_like(k1), k2, bad, validate=True)
            rej = f1 == 2
            assert torch.equal(f3 == 2, rej) and torch.equal(o3[~rej], o1[~rej]) and not bool(o3[rej].any()), ("validate dsm", curve, n, rounds)
        rounds += 1
        units += n
        if time.time() - last > 30:   # a heartbeat: long silent runs look hung to a job runner
            last = time.time()
            print(f"  {rounds} rounds, {units} units, {last - t0:.0f} s", flush=True)
    print(f"soak ok: {rounds} rounds, {units} units per operation, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
