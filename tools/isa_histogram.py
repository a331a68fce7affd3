#!/usr/bin/env python3
"""Per-body instruction histograms of a kernel, from the gfx950 assembly hipcc emits.

    python tools/isa_histogram.py k_p256.hip 'k_scalarmul_var_unsat<eccx::P256U, false>' > profiles/r02_isa_p256r1_var.txt

Compiles eccoxide_amd/csrc/<unit> with the library's flags (device code only, -S), cuts the named
kernel into basic blocks and prints, for every block with at least --min-mads multiplies, the
instruction mix and its weight at the issue costs measured on MI355X
(profiles/r01_valu_rates.jsonl: v_mad_u64_u32 5.03 cycles, three-operand / 64-bit VALU 4.2, two-operand
VALU 2.5 per wave instruction per SIMD).  The big blocks of the ladders are the doubling body, the
addition body (often split in two by a branch) and the table-entry body.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eccoxide_amd", "csrc")
COST = {"v_mad_u64_u32": 5.03, "v_mad_i64_i32": 5.1, "v_lshrrev_b64": 4.2, "v_ashrrev_i64": 4.2, "v_lshlrev_b64": 4.2,
        "v_lshl_add_u64": 4.5, "v_add3_u32": 4.2, "v_alignbit_b32": 4.2, "v_lshl_add_u32": 4.2, "v_or3_b32": 4.2,
        "v_bitop3_b32": 4.2, "v_xad_u32": 4.2, "v_mul_lo_u32": 4.35, "v_mul_hi_u32": 4.2, "v_cndmask_b32_e64": 4.2}
DEFAULT_COST = 2.5


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return dict(zip(names, out.strip().split("\n")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit")
    ap.add_argument("kernel", help="substring of the demangled kernel name")
    ap.add_argument("--min-mads", type=int, default=150)
    ap.add_argument("--asm", help="reuse an assembly file instead of compiling")
    ap.add_argument("--branches", action="store_true",
                    help="list every conditional branch of the kernel with the instruction that produced its condition "
                         "(the census behind the secret-scalar kernels' claim: no branch on scalar-derived data)")
    ap.add_argument("--define", "-D", action="append", default=[], help="extra -D macro for the compile")
    args = ap.parse_args()
    asm = args.asm
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(), "unit.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm",
                               "-pragma-unroll-threshold=1000000", "-S", "--cuda-device-only", os.path.join(CSRC, args.unit),
                               "-o", asm] + ["-D" + d for d in args.define], stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    labels = re.findall(r"^(_Z\w+):", txt, flags=re.M)
    names = demangle(labels)
    want = [m for m in labels if args.kernel in names[m] and "k_" in names[m]]
    if not want:
        sys.exit("kernel not found; kernels: " + ", ".join(sorted(set(v.split("(")[0] for v in names.values() if "k_" in v))))
    mangled = want[0]
    start = txt.index(mangled + ":")
    end = txt.index(".Lfunc_end", start)
    body = txt[start:end].split("\n")
    meta = {}
    mm = re.search(r"\.name:\s+%s.*?(?=\n  - |\Z)" % re.escape(mangled), txt, flags=re.S)
    if mm:
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            m2 = re.search(r"\.%s:\s+(\d+)" % key, mm.group(0))
            if m2:
                meta[key] = int(m2.group(1))
    blocks, cur = [], ["<entry>", collections.Counter()]
    for line in body:
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            blocks.append(cur)
            cur = [m.group(1), collections.Counter()]
            continue
        m = re.match(r"^\s+([a-z][a-z_0-9]*)", line)
        if m:
            cur[1][m.group(1)] += 1
    blocks.append(cur)
    print(f"# {names[mangled].split('(')[0]}")
    print(f"# unit {args.unit}; resources: {meta}")
    if args.branches:
        branch_census(body)
        return
    print("# block: instructions / VALU / multiplies; share of the block's issue cycles spent on multiplies")
    for name, c in blocks:
        mads = c["v_mad_u64_u32"] + c["v_mad_i64_i32"]
        if mads < args.min_mads:
            continue
        valu = {k: v for k, v in c.items() if k.startswith("v_")}
        cyc = sum(v * COST.get(k.replace("_e32", ""), COST.get(k, DEFAULT_COST)) for k, v in valu.items())
        mad_cyc = c["v_mad_u64_u32"] * COST["v_mad_u64_u32"] + c["v_mad_i64_i32"] * COST["v_mad_i64_i32"]
        tot = sum(c.values())
        print(f"\n{name}: {tot} instructions, {sum(valu.values())} VALU, {mads} multiplies "
              f"({100.0 * mads / max(1, sum(valu.values())):.1f} % of VALU instructions, {100.0 * mad_cyc / cyc:.1f} % of VALU issue cycles)")
        for k, v in c.most_common():
            print(f"    {v:6d}  {k}")


def branch_census(body):
    """Every conditional branch, the instruction that last wrote the register it tests (scc / vcc / exec) and the
    memory-instruction mix.  A scalar compare (s_cmp_*) can only see SGPRs: loop counters, kernel arguments and
    values made wave-uniform beforehand (ballots, readlanes); those producers are listed too."""
    mem = collections.Counter()
    uniformisers = collections.Counter()
    rows = []
    block = "<entry>"
    last = {"scc": "-", "vcc": "-", "exec": "-"}
    sgpr_def = {}  # "s[4:5]" / "s7" -> the instruction that last wrote it (textually: a hint, not a dataflow analysis)

    def trace(producer, depth=2):
        """follow the SGPR operands of a mask-combining producer (s_and_b64 vcc, exec, s[4:5]; s_and_saveexec_b64 ...)
        to the instructions that last wrote them, so that the census shows the comparison behind the mask"""
        out = producer
        if depth == 0 or producer == "-":
            return out
        ops = producer.split(None, 1)
        if len(ops) < 2 or not ops[0].startswith(("s_and", "s_or", "s_xor", "s_nor", "s_nand", "s_cselect", "s_mov", "s_not")):
            return out
        srcs = [a.strip() for a in ops[1].split(",")][1:]
        for src in srcs:
            if re.fullmatch(r"s\d+|s\[\d+:\d+\]", src) and src in sgpr_def:
                out += f"   [{src} <- {trace(sgpr_def[src], depth - 1)}]"
        return out
    for line in body:
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^\s+([a-z][a-z_0-9]*)\s*(.*)", line)
        if not m:
            continue
        op, args = m.group(1), m.group(2).split(";")[0].strip()
        if op.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_", "s_load", "s_buffer_load")):
            mem[op] += 1
        if op.startswith(("v_readlane", "v_readfirstlane")) or "ballot" in op:
            uniformisers[op] += 1
        if op.startswith("s_cbranch"):
            reg = "scc" if "scc" in op else ("vcc" if "vcc" in op else ("exec" if "exec" in op else "?"))
            rows.append((block, op, args, trace(last.get(reg, "-"))))
            continue
        dst = args.split(",")[0].strip() if args else ""
        if re.fullmatch(r"s\d+|s\[\d+:\d+\]", dst) and not op.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_nop")):
            # for the instructions that write scc as a side effect the defining compare matters more than the op itself
            sgpr_def[dst] = f"{op} {args}" + (f" (scc <- {last['scc']})" if op.startswith("s_cselect") else "")
        if op.startswith("s_cmp") or op.startswith("s_bitcmp") or (op.startswith("s_") and not op.startswith(("s_mov", "s_load", "s_waitcnt", "s_nop", "s_branch", "s_barrier", "s_cbranch")) and not op.startswith("s_cmp")):
            # most SALU arithmetic writes scc as a side effect; remember the last one
            last["scc"] = f"{op} {args}"
        if op.startswith("v_cmp") and ("vcc" in args.split(",")[0] or "_e32" in op):
            last["vcc"] = f"{op} {args}"
        if "saveexec" in op or (op.startswith("s_") and args.startswith("exec")):
            last["exec"] = f"{op} {args}"
        if op.startswith("s_") and args.startswith("vcc"):
            last["vcc"] = f"{op} {args}"
    print(f"# {len(rows)} conditional branches")
    for block, op, args, producer in rows:
        print(f"  {block:14s} {op:18s} {args:12s} <- {producer}")
    print("# wave-uniformising instructions (a branch on their results would be data-dependent):",
          dict(uniformisers) if uniformisers else "none")
    print("# memory instructions:")
    for k, v in sorted(mem.items()):
        print(f"    {v:6d}  {k}")


if __name__ == "__main__":
    main()
