#!/usr/bin/env python3
"""Tables of DESIGN.md section 6, generated from the committed rocprofv3 summaries so that the prose cannot drift
from the profiles it cites (round 2's traffic table had).

    python tools/gen_design_tables.py            # print the block
    python tools/gen_design_tables.py --write    # replace the block between the GENERATED markers in DESIGN.md

Every number comes from profiles/<round>_<workload>[_<variant>].json through the same lookups bench.py uses
(measured_traffic, measured_clock): per step, the kernels' FETCH_SIZE x 2 (the gfx950 correction) + WRITE_SIZE; the
dominant kernel's duration, its clock (GRBM_GUI_ACTIVE / 8 / duration of the same dispatch), SQ_INSTS_VALU and the cycles
a SIMD spends per vector instruction; its resource line from the code object.  tests/test_design_tables.py fails when
DESIGN.md and the profiles disagree.
"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN = "<!-- BEGIN GENERATED: tools/gen_design_tables.py -->"
END = "<!-- END GENERATED -->"

ROWS = [  # (workload, variant, label)
    ("p256r1_var_2^20", "default", "p256r1 variable base 2^20 (headline)"),
    ("p256r1_var_2^20", "ct", "p256r1 variable base 2^20, ECCX_CT_SCAN"),
    ("p384r1_var_2^19", "default", "p384r1 variable base 2^19"),
    ("p521r1_var_2^19", "default", "p521r1 variable base 2^19"),
    ("bls12_381_g1_var_2^20", "default", "bls12_381_g1 variable base 2^20"),
    ("bls12_381_g1_var_2^20", "glv", "bls12_381_g1 variable base 2^20, ECCX_ASSUME_SUBGROUP"),
    ("bls12_381_g1_var_2^20", "ct", "bls12_381_g1 variable base 2^20, ECCX_CT_SCAN"),
    ("bls12_381_g1_var_2^20", "ctsub", "bls12_381_g1 variable base 2^20, ECCX_CT_SCAN + ECCX_ASSUME_SUBGROUP"),
    ("ed25519_var_2^20", "default", "ed25519 variable base 2^20"),
    ("ed25519_var_2^20", "ct", "ed25519 variable base 2^20, ECCX_CT_SCAN"),
    ("p256r1_verify_2^20", "default", "p256r1 verify shape 2^20"),
    ("p256r1_base_2^20", "default", "p256r1 mul_base 2^20"),
    ("p256r1_base_2^20", "ct", "p256r1 mul_base 2^20, ECCX_CT_SCAN (scan)"),
    ("p256r1_base_2^20", "ctg", "p256r1 mul_base 2^20, ECCX_CT_GATHER"),
    ("ed25519_base_2^20", "default", "ed25519 mul_base 2^20"),
    ("ed25519_base_2^20", "ct", "ed25519 mul_base 2^20, ECCX_CT_SCAN (scan)"),
    ("ed25519_base_2^20", "ctg", "ed25519 mul_base 2^20, ECCX_CT_GATHER"),
    ("ed25519_base_2^20", "lds", "ed25519 mul_base 2^20, table in LDS"),
    ("x25519_2^20", "default", "X25519 2^20"),
]


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def block():
    b = load_bench()
    out = [BEGIN,
           "| Workload | summary | dominant kernel: median ms, VGPRs / scratch B / LDS B, waves per SIMD | clock GHz | VALU wave-instructions per launch | cycles per VALU instruction | HBM bytes per step (FETCH x 2 + WRITE) | algorithmic bytes | ratio |",
           "|---|---|---|---|---|---|---|---|---|"]
    for workload, variant, label in ROWS:
        curve, op, n, alg, _ = b.WORKLOADS[workload]
        rel, found = b._step_entries(workload, curve, op, variant)
        if not found:
            continue
        prof = json.load(open(os.path.join(ROOT, rel)))
        key, v = found[0]
        name, grid = key.split(" grid=")
        row = next((r for r in prof["kernel_trace"] if r["kernel"] == name and str(r["grid_threads"]) == grid), None) or \
            next(r for r in prof["kernel_trace"] if r["kernel"] == name)
        res = row.get("resources")
        resline = (f"{res['vgpr']} / {res['scratch_bytes']} / {res['lds_bytes']}, {res['waves_per_simd_by_registers']}" if res
                   else f"(rocprofv3: {row.get('vgpr', '?')}) / {row.get('scratch_bytes', '?')} / {row.get('lds_bytes', '?')}, ?")
        t = b.measured_traffic(workload, curve, op, variant)
        c = b.measured_clock(workload, curve, op, variant)
        valu = v.get("SQ_INSTS_VALU", {}).get("avg_per_dispatch")
        algb = alg * n
        out.append("| %s | `%s` | %.2f ms, %s | %s | %s | %s | %.2f GB (%.2f x 2 + %.2f) | %.0f MB | %.0f |" % (
            label, os.path.basename(rel), row["median_us"] / 1e3, resline,
            ("%.2f" % (c["hz"] / 1e9)) if c else "n/a (round-2 summary)",
            ("%.3g" % valu) if valu else "n/a",
            ("%.2f" % c["cycles_per_valu_inst"]) if c and c.get("cycles_per_valu_inst") else "n/a",
            t["bytes"] / 1e9, t["fetch_raw"] / 1e9, t["write"] / 1e9, algb / 1e6, t["bytes"] / algb))
    out.append(END)
    return "\n".join(out)


def main():
    txt = block()
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        s = open(path).read()
        a, z = s.index(BEGIN), s.index(END) + len(END)
        open(path, "w").write(s[:a] + txt + s[z:])
    else:
        print(txt)


if __name__ == "__main__":
    main()
