#!/usr/bin/env python3
"""Per-kernel resources from the code objects inside eccoxide_amd/libeccx.so.

    python tools/kernel_resources.py [path/to/libeccx.so] > profiles/r03_kernel_resources.json

rocprofv3's kernel trace reports `vgpr_count` at HALF the allocation on gfx950 (64 for a kernel whose code
object says 128), so the resource lines of the profile summaries come from here instead: the AMDGPU metadata
note of each embedded gfx950 code object (`llvm-objdump --offloading` to unbundle, `llvm-readelf --notes` to
read): allocated VGPRs / AGPRs / SGPRs, scratch (`.private_segment_fixed_size`) and static LDS
(`.group_segment_fixed_size`) per kernel, plus the waves per SIMD the register allocation permits
(MI355X_MICROARCH.md, Register files: granule 8, 512 registers per lane per SIMD).
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = {"vgpr_count": "vgpr", "agpr_count": "agpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch_bytes",
        "group_segment_fixed_size": "lds_bytes", "max_flat_workgroup_size": "max_workgroup", "vgpr_spill_count": "vgpr_spills",
        "sgpr_spill_count": "sgpr_spills"}


def waves_per_simd(vgpr, agpr):
    alloc = -(-(vgpr + agpr) // 8) * 8
    return min(8, 512 // max(alloc, 8))


def kernel_resources(lib_path):
    tmp = tempfile.mkdtemp(prefix="eccx_co_")
    try:
        work = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=False)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True,
                                   text=True).stdout
            # one "- .agpr_count:" item per kernel; its scalar fields sit at four spaces of indentation
            for item in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
                fields = dict(re.findall(r"^\s{2,4}\.([a-z_]+):\s+(\S+)\s*$", "    " + item, flags=re.M))
                name = fields.get("name")
                if not name:
                    continue
                rec = {short: int(fields[key]) for key, short in KEYS.items() if key in fields and fields[key].isdigit()}
                rec["waves_per_simd_by_registers"] = waves_per_simd(rec.get("vgpr", 0), rec.get("agpr", 0))
                out[name] = rec
        names = list(out)
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.strip().split("\n")
        return {d.replace("void ", "").split("(")[0]: out[m] for m, d in zip(names, dem)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "eccoxide_amd", "libeccx.so")
    res = kernel_resources(lib)
    json.dump({"library": os.path.relpath(lib, ROOT), "source": "AMDGPU metadata notes of the embedded gfx950 code objects",
               "kernels": dict(sorted(res.items()))}, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
