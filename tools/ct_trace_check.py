#!/usr/bin/env python3
"""Dynamic check of the secret-scalar kernels: does the executed instruction stream depend on the scalars?

The static side is the branch census (tools/isa_histogram.py --branches).  This is the dynamic side: the same batch
of base points is multiplied by six scalar sets -- random, all zero, all ones (0xff..), order - 1,
one random scalar repeated, random again with another seed; after one warm-up pass the six run forward and then backward -- and the hardware's per-dispatch instruction counters (rocprofv3 PMC: vector, scalar,
vector-memory read / write, LDS and scalar-memory instructions, waves) are compared kernel by kernel.  A kernel whose
control flow and memory instruction stream do not depend on the scalars executes EXACTLY the same number of each,
whatever the scalars; the public-scalar kernels run beside them as a control (they skip work on wave-uniform
conditions, and their counts move).

    # on the GPU box:
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \\
        SQ_INSTS_LDS SQ_INSTS_SMEM -d $REPO/gpurun_out/ct_trace -o pmc -- python3 $REPO/tools/ct_trace_check.py run
    python3 tools/ct_trace_check.py summarize gpurun_out/ct_trace > gpurun_out/r03_ct_instruction_counts.json
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 1 << 16
PATTERNS = ["random", "zero", "ones", "order_minus_1", "one_scalar_repeated", "random_other_seed"]
SEQUENCE = PATTERNS + PATTERNS[::-1]  # forward, then backward: a drift of the clock over the run cancels in a pattern's two durations
CURVES = ["p256r1", "p384r1", "p521r1", "bls12_381_g1", "ed25519"]
COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"]
# what is launched per (curve, pattern), in this order; `kernels`: name fragments of the dispatches that belong to the call
CALLS = [
    ("ct fixed base, scan", dict(base=True, ct_scan=True), ["scalarmul_base_ct<", "batch_to_affine_unsat"]),
    ("ct fixed base, gather", dict(base=True, ct_gather=True), ["scalarmul_base_ct<", "batch_to_affine_unsat"]),
    ("ct variable base", dict(base=False, ct_scan=True), ["scalarmul_coz_unsat<", "ed_scalarmul_var_unsat<", "batch_to_affine_unsat", "var_mirror_unsat"]),
    # bls12_381_g1 only: bases vouched to be in G1 (ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP), the two-half ladder <..., BLS12_381_GLV, true, ...>
    ("ct variable base, bases in the subgroup", dict(base=False, ct_scan=True, assume_subgroup=True), ["scalarmul_coz_unsat<"]),
    ("public fixed base (control)", dict(base=True), ["scalarmul_base_unsat<", "batch_to_affine_unsat"]),
    ("public variable base (control)", dict(base=False), ["scalarmul_coz_unsat<", "ed_scalarmul_var_unsat<", "scalarmul_var_unsat<", "batch_to_affine_unsat"]),
]


def calls_of(curve):
    return [c for c in CALLS if "assume_subgroup" not in c[1] or curve == "bls12_381_g1"]


def scalars(curve, pattern, n):
    import numpy as np

    from eccoxide_amd import workload as W

    sb = {"p256r1": 32, "p384r1": 48, "p521r1": 66, "bls12_381_g1": 32, "ed25519": 32}[curve]
    if pattern == "random":
        return W.random_scalars(curve, n, seed=77)
    if pattern == "random_other_seed":
        return W.random_scalars(curve, n, seed=79)
    if pattern == "zero":
        return np.zeros((n, sb), dtype=np.uint8)
    if pattern == "ones":
        return np.full((n, sb), 0xFF, dtype=np.uint8)
    if pattern == "order_minus_1":
        return np.tile(np.frombuffer((W.order(curve) - 1).to_bytes(sb, "big"), dtype=np.uint8), (n, 1))
    return np.tile(W.random_scalars(curve, 1, seed=78), (n, 1))


def run():
    import torch

    import eccoxide_amd as E
    from eccoxide_amd import workload as W

    dev = torch.device("cuda", 0)
    with E.Engine(0) as eng:
        for curve in CURVES:
            eng.prepare(curve, base=True, ct=True, ct_gather=True)
            eng.reserve(curve, N, var=True, ct=True)
            pts, _ = eng.scalarmul_base_t(curve, torch.from_numpy(W.random_scalars(curve, N, seed=76)).to(dev))
            torch.cuda.synchronize()
            for pattern in PATTERNS + SEQUENCE:  # one untimed pass first: the clock settles before the dispatches that are compared
                ks = torch.from_numpy(scalars(curve, pattern, N)).to(dev)
                for label, kw, _ in calls_of(curve):
                    kw = dict(kw)
                    if kw.pop("base"):
                        eng.scalarmul_base_t(curve, ks, **kw)
                    else:
                        eng.scalarmul_var_t(curve, ks, pts, **kw)
                    torch.cuda.synchronize()
    print("ct_trace_check run ok")


def summarize(root):
    dbs = glob.glob(os.path.join(root, "**", "*.db"), recursive=True)
    db = sqlite3.connect(dbs[0])
    rows = db.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection").fetchall()
    disp = {}
    for d, name, grid, counter, value, duration in rows:
        rec = disp.setdefault(d, {"kernel": name.replace("void ", "").split("(")[0], "grid": grid, "c": {}, "us": duration / 1e3})
        rec["c"][counter] = rec["c"].get(counter, 0.0) + value
    order = [disp[d] for d in sorted(disp)]
    # replay the launch order of run(): per curve, after the setup dispatches, PATTERNS x CALLS; dispatches are matched
    # to a call by kernel-name fragment, in order
    out = {"units_per_dispatch": N, "patterns": PATTERNS, "counters": COUNTERS, "calls": {}}
    struct = {"p256r1": "P256U", "p384r1": "P384U", "p521r1": "P521U", "bls12_381_g1": "BLS12_381U", "ed25519": "ED25519U"}
    for curve in CURVES:
        mine = [r for r in order if f"eccx::{struct[curve]}" in r["kernel"] and r["grid"] >= 256]
        for label, kw, frags in calls_of(curve):
            want_ct = "ct_scan" in kw or "ct_gather" in kw
            gather = "ct_gather" in kw
            subgroup = "assume_subgroup" in kw
            per_kernel = {}
            for r in mine:
                k = r["kernel"]
                if not any(f in k for f in frags):
                    continue
                is_ct_kernel = ("_ct<" in k) or k.endswith(", true>") and ("coz_unsat" in k or "ed_scalarmul_var_unsat" in k)
                if "batch_to_affine" in k or "var_mirror" in k:
                    continue  # shared by every call: compared separately below
                if want_ct != is_ct_kernel:
                    continue
                if "scalarmul_base_ct<" in k and (k.endswith(", true>") != gather):
                    continue
                if want_ct and "coz_unsat" in k and (("PrimeOrderBases" in k or "_GLV" in k) != subgroup):
                    continue
                per_kernel.setdefault(k, []).append(dict(r["c"], duration_us=r["us"]))
            entry = {}
            for k, runs in per_kernel.items():
                # the timed region launches each kernel once per pattern (the first len(PATTERNS) dispatches after setup)
                runs = runs[-len(SEQUENCE):]
                same = all(all(run.get(c) == runs[0].get(c) for c in COUNTERS) for run in runs)
                us = [(runs[i]["duration_us"] + runs[len(SEQUENCE) - 1 - i]["duration_us"]) / 2 for i in range(len(PATTERNS))]
                # GRBM_GUI_ACTIVE counts cycles on each of the 8 XCDs: busy cycles of the dispatch = sum / 8
                cyc = [(runs[i].get("GRBM_GUI_ACTIVE", 0) + runs[len(SEQUENCE) - 1 - i].get("GRBM_GUI_ACTIVE", 0)) / 16 for i in range(len(PATTERNS))]
                # two dispatches per pattern (run forward, then backward), counters attached: durations are indicative only
                # (they include the clock the chip chose for that dispatch), the instruction counts are the evidence
                entry[k] = {"identical_across_scalar_patterns": same,
                            "duration_us_per_pattern": dict(zip(PATTERNS, us)),
                            "duration_spread": (max(us) - min(us)) / min(us) if min(us) > 0 else None,
                            "busy_cycles_per_pattern": dict(zip(PATTERNS, cyc)),
                            "busy_cycles_spread": (max(cyc) - min(cyc)) / min(cyc) if min(cyc) > 0 else None,
                            "clock_ghz_per_pattern": {p: (c / (u * 1e3) if u > 0 else None) for p, c, u in zip(PATTERNS, cyc, us)},
                            "per_pattern": {p: {c: run.get(c) for c in COUNTERS} for p, run in zip(PATTERNS, runs)},
                            "differing_dispatches": [] if same else [
                                {"position": i, "pattern": SEQUENCE[i] if len(runs) == len(SEQUENCE) else None,
                                 "counters": {c: run.get(c) for c in COUNTERS if run.get(c) != runs[0].get(c)}}
                                for i, run in enumerate(runs) if any(run.get(c) != runs[0].get(c) for c in COUNTERS)]}
            out["calls"].setdefault(curve, {})[label] = entry
        # the normalisation kernel runs behind every call: its last len(PATTERNS) x len(CALLS) dispatches of this curve are
        # the timed ones (earlier ones belong to the table builds and the base-point batch)
        norm = [r["c"] for r in mine if "batch_to_affine_unsat" in r["kernel"]][-len(SEQUENCE) * len(calls_of(curve)):]
        if norm:
            same = all(all(run.get(c) == norm[0].get(c) for c in COUNTERS) for run in norm[1:])
            out["calls"][curve]["normalisation (behind every call above, every pattern)"] = {
                "dispatches": len(norm), "identical_across_scalar_patterns": same, "first": {c: norm[0].get(c) for c in COUNTERS},
                "distinct_SQ_INSTS_VALU": sorted({run.get("SQ_INSTS_VALU") for run in norm})}
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")
    verdict = []
    for curve, calls in out["calls"].items():
        for label, entry in calls.items():
            if label.startswith("ct "):
                for k, e in entry.items():
                    verdict.append((curve, label, k, e["identical_across_scalar_patterns"]))
            elif label.startswith("normalisation"):
                verdict.append((curve, label, "k_batch_to_affine_unsat", entry["identical_across_scalar_patterns"]))
    for curve, calls in out["calls"].items():
        for label, entry in calls.items():
            if label.startswith("normalisation"):
                continue
            for k, e in entry.items():
                sys.stderr.write("  %-14s %-32s duration over the scalar patterns: %8.1f .. %8.1f us (spread %.1f %%), busy cycles spread %.1f %%%s\n" % (
                    curve, label, min(e["duration_us_per_pattern"].values()), max(e["duration_us_per_pattern"].values()),
                    100 * e["duration_spread"], 100 * (e["busy_cycles_spread"] or 0), "" if label.startswith("ct ") else "   <- control"))
    bad = [v for v in verdict if not v[3]]
    sys.stderr.write(f"secret-scalar kernels checked: {len(verdict)}, with scalar-dependent instruction counts: {len(bad)}\n")
    for v in bad:
        sys.stderr.write("  DIFFERS: %s / %s / %s\n" % v[:3])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        sys.exit(__doc__)
