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


def _var_unsat(n, nz, sb, a0, inv, sat_pairs, mont=True, norm_u=8, glv_bits=0, merged_y3=False, inv30=(0, 0, 0), wb=5, full_windows=0):
    """v_mad_u64_u32 / v_mad_i64_i32 per unit of the default variable-base path (kernels_coz.hpp in front
    of kernels_unsat.hpp): n limbs of 28/29 bits, one mad per limb product.  A product is n*n mads, a
    square n(n+1)/2, a Montgomery reduction n*nz (nz = non-zero reduction digits per Montgomery factor:
    those of p + 1 for P-256, the 4 signed terms of p + 1 for P-384, all n for BLS12-381, none for the
    Mersenne prime whose wrapped half shares the columns).
    Window table of 16 affine entries: 2P from the affine P (2 products + 4 squares), 14 co-Z additions
    (4 products + 2 squares; a = -3: one more product for the running denominator), one pass to the
    common denominator (4 products + 1 square per entry) -- on the a = -3 curves behind one division-step
    inversion per unit: inv30 = (30-bit limbs, non-zero limbs of p, batches of 30 steps), per batch
    8 mads per limb for the two 2x2 matrix updates + 2 per non-zero limb of p, and 2 products + 1
    square for the conversions and entry 16.
    Ladder: ceil((bits + 1)/5) signed windows of 5 doublings + 1 mixed addition (8 products + 3 squares);
    the top window's entry is loaded, not added.
    a = -3: doubling 4 products + 4 squares; merged_y3 (P-384): Y3 takes one reduction for two products
    in the doubling and in the addition.
    a = 0 (BLS12-381): doubling 3 products + 4 squares with 6 reductions, additions with 10 (Y3 merged);
    glv_bits > 0: the endomorphism form, two half-length scalars, 2 additions per window and one more
    product per entry (beta x).
    wb: window width (5 default; 4 for the secret-scalar form, whose table has 2^(wb-1) entries);
    full_windows: the bottom windows of the secret-scalar form that also compute 2 * entry from its affine
    coordinates (2 products + 4 squares; kernels_coz.hpp coz_full_windows()).
    (inv, sat_pairs and norm_u are kept for reference only)"""
    prod, sq, red = n * n, n * (n + 1) // 2, n * nz

    def cost(p, s, r=None):
        return p * prod + s * sq + (p + s if r is None else r) * red

    conv = cost(5, 0) if mont else 0                     # 2 products into the working form, 3 out of it
    norm = cost(9, 1)                                    # normalisation kernel: 7 products + 1 square + 2 conversions out
    bits = glv_bits if glv_bits else 8 * sb
    nwin = (bits + 1 + wb - 1) // wb
    tbl = 1 << (wb - 1)
    per_entry = cost(4, 1) + (cost(1, 0) if glv_bits else 0)
    if a0:
        dbl, madd = cost(3, 4, 6), cost(8, 3, 10)        # doubling: X*B, E*t, Y*Z + X^2, Y^2, E^2 and -2(2B)^2 as a square; 6 reductions
        build = cost(2, 4) + (tbl - 2) * cost(4, 2) + (tbl - 1) * per_entry + (cost(1, 0) if glv_bits else 0)   # (+ beta x of the top entry)
        tail = cost(1, 0)                                # Z *= zeta
    else:
        dbl, madd = cost(4, 4), cost(8, 3)
        if merged_y3:   # P-384: Y3 is one reduction for two products (signed columns)
            dbl, madd = cost(4, 4, 7), cost(8, 3, 10)
        n30, nz30, batches = inv30
        inversion = batches * (8 * n30 + 2 * nz30) + (cost(2, 0) if mont else 0)
        build = cost(2, 4) + (tbl - 2) * cost(5, 2) + inversion + cost(3, 1) + (tbl - 1) * per_entry
        tail = 0
    total = build + (nwin - 1) * wb * dbl + (nwin * (2 if glv_bits else 1) - 1) * madd + tail + full_windows * cost(2, 4)
    return {"mad": total + conv + norm, "pair": 0}


def _var_generic(n, nz, sb, mont=True):
    """The generic ladder alone (kernels_unsat.hpp), as the fused verification kernel runs it: Jacobian
    table entries with cached Z^2, Z^3 -- even entries by doubling, odd ones by addition (8 doublings + 7
    additions of 11 products + 3 squares, 1 product + 1 square per entry), then per window 5 doublings
    + 1 addition."""
    prod, sq, red = n * n, n * (n + 1) // 2, n * nz

    def cost(p, s):
        return p * prod + s * sq + (p + s) * red

    nwin = (8 * sb + 1 + 4) // 5
    dbl, add = cost(4, 4), cost(11, 3)
    total = 8 * dbl + 7 * add + 15 * cost(1, 1) + (nwin - 1) * 5 * dbl + (nwin - 1) * add
    return {"mad": total + (cost(5, 0) if mont else 0) + cost(9, 1), "pair": 0}


WORKLOADS = {
    # name: (curve, op, per-GPU batch, algorithmic bytes per unit, multiplier instructions per unit)
    "p256r1_var_2^20": ("p256r1", "var", 1 << 20, 160, _var_unsat(9, 4, 32, 0, 383, 8 * 8 + 8 * 3, norm_u=16, inv30=(9, 7, 20))),
    # fixed base, default path: 16-bit windows, 16 additions of 7 products (Edwards, 81 + 9 mads each)
    # or of 8 products + 3 squares (P-256) on unsaturated limbs, then the saturated normalisation
    "ed25519_base_2^20": ("ed25519", "base", 1 << 20, 96,
                          {"mad": (15 * 7 + 1 + 8) * (81 + 9), "pair": 0}),   # window 0 is loaded (1 product), 15 additions
    "p256r1_base_2^20": ("p256r1", "base", 1 << 20, 96,
                         {"mad": 15 * (8 * (81 + 36) + 3 * (45 + 36)) + 9 * (81 + 36) + (45 + 36), "pair": 0}),   # window 0 is loaded
    # unsaturated 9 x 29 ladder: per bit 5 products (81 + 9 mads), 4 squares (45 + 9), one small multiple (9 + 1)
    "p384r1_base_2^19": ("p384r1", "base", 1 << 19, 144,
                         {"mad": 23 * (8 * 252 + 3 * 161) + 9 * 252 + 161, "pair": 0}),
    "p521r1_base_2^19": ("p521r1", "base", 1 << 19, 198,
                         {"mad": 32 * (8 * 324 + 3 * 171) + 9 * 324 + 171, "pair": 0}),
    "bls12_381_g1_base_2^20": ("bls12_381_g1", "base", 1 << 20, 128,
                               {"mad": 15 * (8 * 196 + 3 * 105 + 10 * 196) + 9 * 392 + 301, "pair": 0}),   # Y3 merged: 10 reductions per addition
    "x25519_2^20": ("ed25519", "x25519", 1 << 20, 96,
                    {"mad": 256 * (5 * 90 + 4 * 54 + 10) + 6 * 90, "pair": 0}),
    # edwards25519 variable base: 52 signed windows of 4 x (4 squares + 3 products) + (4 + 4) + a
    # 7-product addition; table of 16 cached multiples (1 doubling, 14 additions of 8, 16 x 2d*T)
    "ed25519_var_2^20": ("ed25519", "var", 1 << 20, 160,
                         {"mad": (51 * 20 + 4) * 54 + (51 * 16 + 52 * 7 + 4 + 14 * 8 + 16 + 1 + 8) * 90, "pair": 0}),
    # verify shape u1*G + u2*Q: the variable-base ladder + 16 mixed additions (8 products + 3 squares)
    "p256r1_verify_2^20": ("p256r1", "dsm", 1 << 20, 192,
                           {"mad": _var_unsat(9, 4, 32, 0, 383, 88, norm_u=16, inv30=(9, 7, 20))["mad"] + 16 * (8 * 117 + 3 * 81), "pair": 0}),
    "p384r1_var_2^19": ("p384r1", "var", 1 << 19, 240, _var_unsat(14, 4, 48, 0, 575, 12 * 12 + 12 * 10, merged_y3=True, inv30=(13, 12, 37))),
    "p521r1_var_2^19": ("p521r1", "var", 1 << 19, 330, _var_unsat(18, 0, 66, 0, 780, 17 * 17, mont=False, inv30=(18, 18, 51))),
    "bls12_381_g1_var_2^20": ("bls12_381_g1", "var", 1 << 20, 224, _var_unsat(14, 14, 32, 1, 570, 2 * 12 * 12)),
}
def _base_ct(n, nz, sb, nbits, w, merged_y3=False):
    """Secret-scalar fixed base (kernels_ct.hpp k_scalarmul_base_ct): ceil((8 SB + 1) / w) signed windows, every one but
    the first a mixed XYZZ addition (8 products + 2 squares; Y3 in one reduction where the field merges it), the top
    ct_unsafe_windows() also 2 * entry from affine coordinates (4 products + 3 squares), 2 products to the Jacobian row,
    then the normalisation as in _var_unsat."""
    prod, sq, red = n * n, n * (n + 1) // 2, n * nz

    def cost(p, s, r=None):
        return p * prod + s * sq + (p + s if r is None else r) * red

    nwin = (8 * sb + 1 + w - 1) // w
    unsafe = nwin - (nbits + w - 1) // w + 1
    return {"mad": (nwin - 1) * cost(8, 2, 9 if merged_y3 else None) + unsafe * cost(4, 3) + cost(2, 0) + cost(9, 1), "pair": 0}


def _ed_base_ct(w):
    """edwards25519 fixed base, secret scalars: ceil(257 / w) windows, the first loaded (1 product), the others complete
    additions of 7 products (81 + 9 mads), the normalisation (5 products)"""
    return {"mad": ((((257 + w - 1) // w) - 1) * 7 + 1 + 5) * 90, "pair": 0}


# multiplier instructions of the non-default variants that have a count of their own.  Window widths as compiled:
# ECCX_CT_VAR_BITS 4, ECCX_CT_ED_VAR_BITS 3, ECCX_CT_BASE_BITS 6 (edwards25519: 5), ECCX_CT_GATHER_BITS 7.
VARIANT_MULT = {
    ("bls12_381_g1_var_2^20", "glv"): _var_unsat(14, 14, 32, 1, 570, 0, glv_bits=129),
    ("p256r1_var_2^20", "ct"): _var_unsat(9, 4, 32, 0, 383, 0, inv30=(9, 7, 20), wb=4, full_windows=1),
    ("p384r1_var_2^19", "ct"): _var_unsat(14, 4, 48, 0, 575, 0, merged_y3=True, inv30=(13, 12, 37), wb=4, full_windows=1),
    ("p521r1_var_2^19", "ct"): _var_unsat(18, 0, 66, 0, 780, 0, mont=False, inv30=(18, 18, 51), wb=4, full_windows=3),
    ("bls12_381_g1_var_2^20", "ct"): _var_unsat(14, 14, 32, 1, 570, 0, wb=4, full_windows=64),
    # the two-half ladder: 33 windows of 4 doublings + 2 additions, beta x per second-half lookup (33 products; the count
    # of _var_unsat has it per table entry, 8), both additions of the bottom window with the doubled entry
    ("bls12_381_g1_var_2^20", "ctsub"): {"mad": _var_unsat(14, 14, 32, 1, 570, 0, glv_bits=129, wb=4, full_windows=2)["mad"] + (33 - 8) * 2 * 196, "pair": 0},
    # edwards25519, 86 signed 3-bit windows: 85 x 3 doublings (4 squares + 3 products, the last of a window + 1), 86
    # additions of 6 products, the 4-entry table (1 doubling, 2 additions of 8, 4 x 2d T, to Z = 1: 15 products
    # + one division-step inversion), the normalisation (5 products)
    ("ed25519_var_2^20", "ct"): {"mad": (85 * 12 + 4) * 54 + (85 * 10 + 86 * 6 + 1 + 4 + 16 + 4 + 15 + 5) * 90 + 20 * (8 * 9 + 2), "pair": 0},
    ("p256r1_base_2^20", "ct"): _base_ct(9, 4, 32, 256, 6), ("p256r1_base_2^20", "ctg"): _base_ct(9, 4, 32, 256, 7),
    ("p384r1_base_2^19", "ct"): _base_ct(14, 4, 48, 384, 6, True), ("p384r1_base_2^19", "ctg"): _base_ct(14, 4, 48, 384, 7, True),
    ("p521r1_base_2^19", "ct"): _base_ct(18, 0, 66, 521, 6), ("p521r1_base_2^19", "ctg"): _base_ct(18, 0, 66, 521, 7),
    ("bls12_381_g1_base_2^20", "ct"): _base_ct(14, 14, 32, 255, 6, True), ("bls12_381_g1_base_2^20", "ctg"): _base_ct(14, 14, 32, 255, 7, True),
    ("ed25519_base_2^20", "ct"): _ed_base_ct(5), ("ed25519_base_2^20", "ctg"): _ed_base_ct(7),
}
# HBM bytes per launch measured with rocprofv3 PMC passes (tools/profile_all.sh -> tools/prof_summary.py;
# summaries committed under profiles/): FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for
# 16-byte-per-lane reads on gfx950, plus WRITE_SIZE, summed over the kernels of one step.  Counters
# cannot be read from inside this process, so `roofline.traffic` is loaded from the committed summary
# of this very workload (named in traffic_detail.source); null when that file is absent.
PROFILE_ROUNDS = ("r03", "r02")  # newest summary of the workload that exists
# kernels of one step per (op, variant): for each, substrings of the kernel name in the summary, newest spelling first
# (round 3 added the window width and the secret-scalar flag to the ladder's template arguments)
STEP_KERNELS = {
    # variable base: the affine-table ladder, the generic ladder as its fix-up pass (reads the flags, redoes
    # marked units: none in these workloads), the normalisation
    ("var", "default"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::{G}, false, false, 5, false>", "k_scalarmul_coz_unsat<eccx::{U}, eccx::{G}, false, false>"],
                         ["k_scalarmul_var_unsat<eccx::{U}, false>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("var", "glv"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::{G}, true, false, 5, false>", "k_scalarmul_coz_unsat<eccx::{U}, eccx::{G}, true, false>"],
                     ["k_scalarmul_var_unsat<eccx::{U}, false>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("var", "mirror"): [["k_scalarmul_var_mirror_unsat<eccx::{U}>"], ["k_batch_to_affine<eccx::{S}, 0,"]],
    # secret scalars: the scanning affine-table ladder, the normalisation, the mirror ladder as fix-up pass
    ("var", "ct"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::NoGlv, false, false, 4, true>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"],
                    ["k_scalarmul_var_mirror_unsat<eccx::{U}>"]],
    # secret scalars, bases vouched to be in the prime-order subgroup (bls12_381_g1)
    ("var", "ctsub"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::{G}, true, false, 4, true>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"],
                       ["k_scalarmul_var_mirror_unsat<eccx::{U}>"]],
    ("dsm", "default"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::NoGlv, false, true, 5, false>", "k_scalarmul_coz_unsat<eccx::{U}, eccx::NoGlv, false, true>"],
                         ["k_scalarmul_var_unsat<eccx::{U}, true>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("dsm", "xonly"): [["k_scalarmul_coz_unsat<eccx::{U}, eccx::NoGlv, false, true, 5, false>"], ["k_scalarmul_var_unsat<eccx::{U}, true>"],
                       ["k_batch_to_affine_unsat<eccx::{U}, 4,"]],
    ("base", "default"): [["k_scalarmul_base_unsat<eccx::{U}>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("base", "ct"): [["k_scalarmul_base_ct<eccx::{U}, false>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("base", "ctg"): [["k_scalarmul_base_ct<eccx::{U}, true>"], ["k_batch_to_affine_unsat<eccx::{U}, 1,"]],
    ("x25519", "default"): [["k_x25519_ladder_unsat<eccx::ED25519U>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 3,"]],
}
ED_STEP_KERNELS = {
    ("var", "default"): [["k_ed_scalarmul_var_unsat<eccx::ED25519U, false, 5, false>", "k_ed_scalarmul_var_unsat<eccx::ED25519U, false>"],
                         ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("var", "ct"): [["k_ed_scalarmul_var_unsat<eccx::ED25519U, false, 3, true>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("dsm", "default"): [["k_ed_scalarmul_var_unsat<eccx::ED25519U, true, 5, false>", "k_ed_scalarmul_var_unsat<eccx::ED25519U, true>"],
                         ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("base", "default"): [["k_ed_scalarmul_base_unsat<eccx::ED25519U>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("base", "lds"): [["k_ed_scalarmul_base_lds6<eccx::ED25519U>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("base", "ct"): [["k_ed_scalarmul_base_ct<eccx::ED25519U, false>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
    ("base", "ctg"): [["k_ed_scalarmul_base_ct<eccx::ED25519U, true>"], ["k_batch_to_affine_unsat<eccx::ED25519U, 2,"]],
}
# (S, U, G): saturated struct, unsaturated struct, the GLV argument the default ladder is instantiated with
CURVE_STRUCTS = {"p256r1": ("P256", "P256U", "NoGlv"), "p384r1": ("P384", "P384U", "NoGlv"), "p521r1": ("P521", "P521U", "NoGlv"),
                 "bls12_381_g1": ("BLS12_381", "BLS12_381U", "BLS12_381_GLV"), "ed25519": ("ED25519", "ED25519U", "NoGlv")}


def profile_path(workload, variant):
    """The committed rocprofv3 summary of this workload and variant: the newest round that has one."""
    for rnd in PROFILE_ROUNDS:
        tag = rnd + "_" + workload.replace("^", "")
        if variant != "default":
            tag += "_" + variant
        rel = os.path.join("profiles", tag + ".json")
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return None


def _step_entries(workload, curve, op, variant):
    """(summary path, [(pmc key, counters)] for the kernels of one step) or (None, None)"""
    rel = profile_path(workload, variant)
    table = ED_STEP_KERNELS if curve == "ed25519" and op != "x25519" else STEP_KERNELS
    pats = table.get((op, variant))
    if rel is None or not pats:
        return None, None
    with open(os.path.join(ROOT, rel)) as f:
        prof = json.load(f)
    S, U, G = CURVE_STRUCTS[curve]
    found = []
    for alts in pats:
        best = None
        for alt in alts:
            pat = alt.format(S=S, U=U, G=G)
            for key, v in prof.get("pmc", {}).items():
                if pat in key and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    # the (kernel, grid) entry with the most dispatches: the timed launches
                    rank = (v["FETCH_SIZE"]["dispatches"], v["hbm_read_bytes_per_dispatch_raw"])
                    if best is None or rank > best[0]:
                        best = (rank, key, v)
            if best is not None:
                break
        if best is None:
            return rel, None
        found.append((best[1], best[2]))
    return rel, found


def measured_traffic(workload, curve, op, variant):
    """HBM bytes of one step from the committed rocprofv3 PMC summary of this workload: for each
    kernel of the step FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE.  None if there is no summary."""
    rel, found = _step_entries(workload, curve, op, variant)
    if not found:
        return None
    fetch = sum(v["hbm_read_bytes_per_dispatch_raw"] for _, v in found)
    write = sum(v["hbm_write_bytes_per_dispatch"] for _, v in found)
    return {"bytes": int(2 * fetch + write), "fetch_raw": int(fetch), "write": int(write), "source": rel, "kernels": [k for k, _ in found]}


def measured_clock(workload, curve, op, variant):
    """Effective shader clock of the step's dominant kernel and its cycles per vector instruction, from the same
    summary (tools/prof_summary.py: GRBM_GUI_ACTIVE / 8 / duration of the same dispatch).  None before round 3's
    summaries exist for the workload."""
    rel, found = _step_entries(workload, curve, op, variant)
    if not found:
        return None
    key, v = found[0]
    clk = v.get("effective_clock_ghz")
    if not clk:
        return None
    return {"hz": clk["median"] * 1e9, "source": rel, "kernel": key, "how": clk.get("how"),
            "cycles_per_valu_inst": (v.get("cycles_per_valu_inst") or {}).get("median")}


HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Integer-multiplier issue costs measured by tools/ubench/valu_rates.hip on MI355X
# (profiles/r01_valu_rates.jsonl): cycles one SIMD needs to issue the instruction for one
# wave.  A lone v_mad_u64_u32 (unsaturated limbs) costs 4.8-5.03; the saturated unit of work,
# the pair v_mad_u64_u32 + v_addc_co_u32, costs 8.62.  1024 SIMDs at 2.4 GHz.
CYC_MAD = 4.8
CYC_PAIR = 8.62
SIMDS = 256 * 4
CLOCK_HZ_NOMINAL = 2.4e9  # only where no summary with GRBM_GUI_ACTIVE exists for the workload: the chip holds about 2.25 GHz on these kernels


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """Host threads this process can really run at once: the smaller of the online CPUs, the affinity
    mask and the cgroup CPU quota (a GPU box of this pool exposes 256 hardware threads and grants a
    share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def host_path(eng, curve, op, n, ks, pts, dev_out, opts_bits, ks2=None):
    """The same batch through the HOST-buffer entry point (what a caller without device buffers -- the Rust crate --
    uses): pageable host memory in, pageable host memory out, PCIe copies and the final synchronisation included.
    Never part of `value` (reported separately).  The first call sizes the context's device-side
    buffers; the best of the next three is reported, and its output must equal the device-resident run's."""
    import ctypes

    import numpy as np

    import eccoxide_amd as E

    cid, fb, sb = E.curve_id(curve), E.field_bytes(curve), E.scalar_bytes(curve)
    lib, ctx = eng._lib, eng._ctx
    hk = np.ascontiguousarray(ks.cpu().numpy())
    hk2 = np.ascontiguousarray(ks2.cpu().numpy()) if op == "dsm" else None
    hp = np.ascontiguousarray(pts.cpu().numpy()) if op in ("var", "dsm", "x25519") else None
    ho = np.empty(tuple(dev_out.shape), dtype=np.uint8)
    hf = np.empty((n,), dtype=np.uint8)
    eng.reserve(curve, n, var=False, host=True)

    def call():
        t = time.perf_counter()
        if op == "var":
            rc = lib.eccx_scalarmul_var(ctx, cid, n, hk.ctypes.data, hp.ctypes.data, ho.ctypes.data, hf.ctypes.data, None, opts_bits)
        elif op == "dsm":
            rc = lib.eccx_double_scalarmul(ctx, cid, n, hk.ctypes.data, hk2.ctypes.data, hp.ctypes.data, ho.ctypes.data, hf.ctypes.data, opts_bits)
        elif op == "x25519":
            rc = lib.eccx_x25519(ctx, n, hk.ctypes.data, hp.ctypes.data, ho.ctypes.data, hf.ctypes.data, 0)
        else:
            rc = lib.eccx_scalarmul_base(ctx, cid, n, hk.ctypes.data, ho.ctypes.data, hf.ctypes.data, None, opts_bits)
        dt = time.perf_counter() - t
        eng._check(rc)
        return dt

    call()
    best = min(call() for _ in range(3))
    same = bool((ho == dev_out.cpu().numpy()).all())
    pcie = int(hk.nbytes + (hk2.nbytes if hk2 is not None else 0) + (hp.nbytes if hp is not None else 0) + ho.nbytes + hf.nbytes)
    return {"ms": best * 1e3, "units_per_s": n / best, "pcie_bytes": pcie, "matches_device_run": same,
            "entry_point": {"var": "eccx_scalarmul_var", "dsm": "eccx_double_scalarmul", "x25519": "eccx_x25519"}.get(op, "eccx_scalarmul_base"),
            "note": "host-buffer call on the same batch: pageable host memory, H2D + kernels + D2H + synchronisation; "
                    "not part of value (tools/hostbench: the same measurement from C, profiles/r03_hostbench.jsonl)"}


def cpu_baseline(ora, curve, op, args, n, ks, ks2, pts):
    """The oracle (C restatement of the reference algorithm: RCB complete formulas, fixed 4-bit
    window, 64-bit-limb Montgomery with unsigned __int128 -- kind "port") timed on this box's host
    cores, in the shape SURVEY.md §8(d) asks for: the first 1024 units of the batch (BASELINE.json
    configs[0]) on ONE thread and on ALL host cores, with nproc, the CPU model and ns per
    operation; plus a larger sample on all cores, which is the `value` reported (1024 units over
    hundreds of threads mostly measures thread start-up).  The authentic number would be the
    reference's own `cargo bench -- <curve>::point::scalar_mul` (benches/curves.rs:247-265):
    unavailable, there is no Rust toolchain on the box."""
    nproc = os.cpu_count() or 1
    usable = _usable_cores()

    def timed(m, threads):
        c_k = ks[:m].cpu().numpy().tobytes()
        c_p = pts[:m].cpu().numpy().tobytes() if op in ("var", "x25519", "dsm") else None
        c_k2 = ks2[:m].cpu().numpy().tobytes() if op == "dsm" else None
        t1 = time.perf_counter()
        if op == "var":
            ora.var(curve, c_k, c_p, threads=threads)
        elif op == "dsm":   # the two scalar multiplications dominate; the final addition is not timed
            ora.base(curve, c_k, threads=threads)
            ora.var(curve, c_k2, c_p, threads=threads)
        elif op == "x25519":
            ora.x25519(c_k, c_p, threads=threads)
        else:
            ora.base(curve, c_k, threads=threads)
        dt = time.perf_counter() - t1
        return {"units": m, "threads": threads, "seconds": dt, "ops_per_s": m / dt, "ns_per_op": dt / m * 1e9}

    small = min(n, 1024)
    one = timed(small, 1)
    allc = timed(small, nproc)
    big_m = min(n, args.cpu_sample)
    # the large sample on every hardware thread, and -- where the process is granted fewer CPUs than
    # the machine has, or the quota cannot be read -- on that share too; the faster one is `value`
    runs = [timed(big_m, nproc)]
    for t in sorted({usable, min(16, nproc)} - {nproc}):
        runs.append(timed(big_m, t))
    big = max(runs, key=lambda r: r["ops_per_s"])
    return {"value": big["ops_per_s"], "unit": "scalarmuls/s", "cores": big["threads"], "kind": "port",
            "sample": f"first {big_m} units of the same {args.workload} batch, oracle/eccx_oracle.c (C restatement of the "
                      f"reference algorithm), {big['threads']} threads, {big['seconds']:.2f} s",
            "nproc": nproc, "usable_cores": usable, "cpu_model": _cpu_model(),
            "n1024_1thread": one, "n1024_allcores": allc, "large_allcores": runs[0], "large_runs": runs,
            "reference_cargo_bench": "unavailable (no Rust toolchain / crates.io on the box; "
                                     "would be benches/curves.rs:247-265)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="p256r1_var_2^20", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--cpu-sample", type=int, default=1 << 17)
    ap.add_argument("--variant", default="default", choices=["default", "mirror", "lds", "l2", "ct", "ctg", "ctsub", "glv", "xonly"],
                    help="default: fast kernels; mirror: reference-mirroring kernels; "
                         "lds: ed25519 fixed base with a signed 6-bit comb table resident in LDS; "
                         "l2: the reference's 4-bit comb read through L2; "
                         "ct: ECCX_CT_SCAN (secret-scalar kernels, every table entry read); ctg: with ECCX_CT_GATHER; "
                         "ctsub: ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP (bls12_381_g1, bases in G1); "
                         "glv: ECCX_ASSUME_SUBGROUP (bls12_381_g1 endomorphism ladder)")
    args = ap.parse_args()
    # ECCX_FORCE_DIST=1: initialise RCCL and run the gather pipeline even at world size 1, so that a
    # one-GPU box exercises process-group init, the asynchronous gather on RCCL's stream and its
    # ordering against the engine's kernels
    force_dist = os.environ.get("ECCX_FORCE_DIST", "0") == "1"

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
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)
    curve, op, n, alg_bytes, mult = WORKLOADS[args.workload]
    mult = VARIANT_MULT.get((args.workload, args.variant), mult)
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
    out_cols = 32 if op == "x25519" else (fb if args.variant == "xonly" else 2 * fb)
    if args.variant == "xonly":
        alg_bytes -= fb  # u1 + u2 + Q in, x out: 160 bytes per p256r1 unit
    # two sets of output buffers: with N > 1 the gather of batch i runs while batch i+1 computes
    outs = [torch.empty((n, out_cols), dtype=torch.uint8, device=dev) for _ in range(2)]
    flagss = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream(dev)
    pipe = GatherPipeline(n, out_cols, dev, slots=2, force=force_dist)

    mirror = args.variant == "mirror"
    ct = args.variant in ("ct", "ctg", "ctsub")
    ctg = args.variant == "ctg"
    xonly = args.variant == "xonly"
    if xonly and op != "dsm":
        sys.exit("--variant xonly applies to the verify workloads")
    glv = args.variant in ("glv", "ctsub")
    # one-time costs out of the timed region (and out of the _dev calls): tables + scratch
    if op in ("base", "dsm"):
        eng.prepare(curve, base=True, base_lds=args.variant == "lds", ct=ct and not ctg, ct_gather=ctg)
    eng.reserve(curve, n, var=op in ("var", "dsm"), mirror=mirror or ct, ct=ct)

    def step(slot):
        out, flags = outs[slot], flagss[slot]
        if op == "var":
            eng.scalarmul_var_t(curve, ks, pts, out, flags, stream=stream.cuda_stream, mirror=mirror, ct_scan=ct,
                                assume_subgroup=glv)
        elif op == "dsm":
            eng.double_scalarmul_t(curve, ks, ks2, pts, out, flags, stream=stream.cuda_stream, x_only=xonly)
        elif op == "x25519":
            eng.x25519_t(ks, pts, out, flags, stream=stream.cuda_stream)
        else:
            eng.scalarmul_base_t(curve, ks, out, flags, stream=stream.cuda_stream, mirror=mirror, ct_scan=ct, ct_gather=ctg,
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
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.steps, ev)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    last = (args.steps - 1) & 1
    out, flags = outs[last], flagss[last]

    if use_dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # correctness gate on the timed buffers: sampled comparison with the oracle (rank 0).  With a
    # gather (N > 1 or ECCX_FORCE_DIST) the bytes checked are the ones that ARRIVED on the root:
    # rank 0's own slice and one peer's slice of the gathered buffer, the peer's inputs regenerated
    # from its seeds.
    parity = None
    cpu = None
    host = None
    opts_bits = (E.engine.MIRROR_REFERENCE if mirror else 0) | (E.engine.CT_SCAN if ct else 0) | (E.engine.CT_GATHER if ctg else 0) | \
                (E.engine.ASSUME_SUBGROUP if glv else 0) | ({"lds": E.engine.TABLE_IN_LDS, "l2": E.engine.TABLE_IN_L2}.get(args.variant, 0))
    if rank == 0:
        from tests import oracle_lib

        ora = oracle_lib.load()

        def expected(s_k, s_k2, s_pts):
            if op == "var":
                w_out, w_inf, _ = ora.var(curve, s_k, s_pts, threads=8)
            elif op == "dsm":
                # u1*G and u2*Q from the C oracle, their sum with textbook affine arithmetic
                from oracle import ecc_ref as R

                c = R.CURVES[curve]
                a_out, a_inf, _ = ora.base(curve, s_k, threads=8)
                b_out, b_inf, _ = ora.var(curve, s_k2, s_pts, threads=8)
                pb = 2 * fb
                le = curve == "ed25519"
                order = "little" if le else "big"
                dec = lambda buf, fl, i: None if (fl[i] and not le) else (int.from_bytes(buf[i * pb:i * pb + fb], order),
                                                                         int.from_bytes(buf[i * pb + fb:(i + 1) * pb], order))
                sums = [R.affine_add(c, dec(a_out, a_inf, i), dec(b_out, b_inf, i)) for i in range(len(a_inf))]
                if xonly:
                    w_out = b"".join(bytes(fb) if t is None else t[0].to_bytes(fb, order) for t in sums)
                else:
                    w_out = b"".join(bytes(pb) if t is None else t[0].to_bytes(fb, order) + t[1].to_bytes(fb, order) for t in sums)
                w_inf = bytes(1 if t is None else 0 for t in sums)
            elif op == "x25519":
                w_out, w_inf = ora.x25519(s_k, s_pts, threads=8)
            else:
                w_out, w_inf, _ = ora.base(curve, s_k, threads=8)
            return w_out, w_inf

        idx_cpu = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:256].sort().values
        idx = idx_cpu.to(dev)
        g_out, g_flags = pipe.result(last)      # the gathered buffers on the root (the local ones without a gather)
        w_out, w_inf = expected(ks[idx].cpu().numpy().tobytes(),
                                ks2[idx].cpu().numpy().tobytes() if ks2 is not None else None,
                                pts[idx].cpu().numpy().tobytes() if pts is not None else None)
        parity = (g_out[idx].cpu().numpy().tobytes() == w_out) and (g_flags[idx].cpu().numpy().tobytes() == w_inf)
        parity = parity and (out[idx].cpu().numpy().tobytes() == w_out)
        if world > 1:
            peer = world - 1
            sel = idx_cpu.numpy()
            p_k = W.random_scalars(curve, n, seed=10 + peer)[sel].tobytes()
            p_r = W.random_scalars(curve, n, seed=1000 + peer)[sel].tobytes()
            p_k2 = W.random_scalars(curve, n, seed=2000 + peer)[sel].tobytes() if op == "dsm" else None
            if op in ("var", "dsm"):
                p_pts = ora.base(curve, p_r, threads=8)[0]
            elif op == "x25519":
                p_pts = ora.x25519(p_r, None, threads=8)[0]
            else:
                p_pts = None
            pw_out, pw_inf = expected(p_k, p_k2, p_pts)
            pidx = idx + peer * n
            parity = parity and (g_out[pidx].cpu().numpy().tobytes() == pw_out) and (g_flags[pidx].cpu().numpy().tobytes() == pw_inf)
        if world == 1 and not args.no_cpu_baseline:  # a reported baseline, timed at N = 1 only (the other ranks would wait)
            cpu = cpu_baseline(ora, curve, op, args, n, ks, ks2, pts)
        if world == 1 and not args.no_host_path:
            host = host_path(eng, curve, op, n, ks, pts, out, opts_bits | (E.engine.OUT_X_ONLY if xonly else 0), ks2)

    if rank == 0:
        total_units = n * world * args.steps
        value = total_units / elapsed
        ach = alg_bytes * n / (kernel_ms * 1e-3) / 1e9
        # time the SIMDs must spend issuing the multiplier instructions of this batch (one
        # instruction serves the 64 units of a wave) against the kernel time
        issue_cycles = mult["mad"] * CYC_MAD + mult["pair"] * CYC_PAIR
        clock = measured_clock(args.workload, curve, op, args.variant)
        clock_hz = clock["hz"] if clock else CLOCK_HZ_NOMINAL
        valu_frac = (n / 64) * issue_cycles / (kernel_ms * 1e-3 * SIMDS * clock_hz)
        mul_rate = (mult["mad"] + mult["pair"]) * n / (kernel_ms * 1e-3)
        traffic = measured_traffic(args.workload, curve, op, args.variant)
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
            "config": {"workload": args.workload, "variant": args.variant, "curve": curve, "op": op, "batch_per_gpu": n,
                       "global_batch": n * world, "parallelism": f"shard{world}" if world > 1 else "single",
                       "gather": "rccl gather to rank 0, overlapped with the next batch (all complete inside the timed region)" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_detail": traffic,
                         "kernel_ms": kernel_ms, "alg_bytes_per_unit": alg_bytes,
                         "alg_bytes_per_launch": alg_bytes * n,
                         "note": "integer-VALU bound path, see valu; traffic above the algorithmic bytes is "
                                 "the per-lane window table of the variable-base ladder / the random reads of the "
                                 "16-bit-window comb table of the fixed-base path (DESIGN.md §6)"},
            "valu": None if (args.variant != "default" and (args.workload, args.variant) not in VARIANT_MULT) else {"bound": "integer multiplier issue (v_mad_u64_u32; + v_addc_co_u32 in saturated kernels)",
                     "achieved": mul_rate / 1e12, "unit": "T limb-products/s",
                     "frac": valu_frac,
                     "mads_per_unit": mult["mad"], "mad_addc_pairs_per_unit": mult["pair"],
                     "issue_cycles_per_wave": issue_cycles,
                     "clock": clock if clock else {"hz": CLOCK_HZ_NOMINAL, "source": "nominal (no rocprofv3 summary with GRBM_GUI_ACTIVE for this workload)"},
                     "note": "frac = share of SIMD issue time spent on multiplier instructions at the "
                             "measured issue cost (4.8 cycles per mad, 8.62 per mad+addc pair)"},
            "cpu_baseline": cpu,
            "host_path": host,
            "parity_sample_ok": parity,
        }
        if force_dist:
            line["config"]["gather"] = "rccl gather forced at world size 1 (ECCX_FORCE_DIST)"
        if not parity:
            # a fast wrong answer is not a result: no value, non-zero exit
            line["value"] = None
            line["invalid"] = "sampled results differ from the oracle"
        print(json.dumps(line), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not parity:
        sys.exit(3)


if __name__ == "__main__":
    main()
