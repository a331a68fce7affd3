#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd SQLite databases written by tools/profile.sh) into a
small JSON file suitable for profiles/.

    python tools/prof_summary.py gpurun_out/prof_r01 > profiles/r01_p256r1_var.json

Kernel-trace: per (kernel, grid) calls / average / min / max duration in microseconds; the resource line of
each kernel comes from the code object's metadata (tools/kernel_resources.py) -- rocprofv3's own vgpr_count
reads half the allocation on gfx950 and is kept only as `vgpr_rocprof`.
PMC passes: per (kernel, grid) median counter value per dispatch.  Where a pass holds GRBM_GUI_ACTIVE, the
effective clock of each dispatch is GRBM_GUI_ACTIVE / 8 XCDs / the duration of the SAME dispatch
(MI355X_MICROARCH.md, DVFS give-back), and with SQ_INSTS_VALU in that pass too, cycles_per_valu_inst =
(GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024 SIMDs): the cycles one SIMD spends per vector instruction it issues.  HBM traffic follows
MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are reported in KiB-like units of the
counter expression; on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x
(doubled here and labelled), WRITE_SIZE reads exact for 16-byte stores.
"""
import glob
import json
import os
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name.split("(")[0]


SIMDS = 1024  # 256 CUs x 4
XCDS = 8


def _resources():
    """kernel name -> resources from the code objects of the library that ran (None if it cannot be read)"""
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kernel_resources as KR

        lib = os.environ.get("ECCX_LIB_PATH") or os.path.join(KR.ROOT, "eccoxide_amd", "libeccx.so")
        return KR.kernel_resources(lib)
    except Exception as e:  # noqa: BLE001 -- a summary without resource lines beats no summary
        sys.stderr.write(f"kernel resources unavailable: {e}\n")
        return None


def trace(db_path, resources=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, grid_x, workgroup_x, duration, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, lds_size from kernels").fetchall()
    agg = {}
    for name, grid, wg, dur, vgpr, agpr, sgpr, scratch, lds in rows:
        k = (short(name), grid, wg)
        a = agg.setdefault(k, {"calls": 0, "sum": 0, "min": 1e30, "max": 0, "durs": [], "vgpr": vgpr, "agpr": agpr, "sgpr": sgpr, "scratch": scratch, "lds": lds})
        a["calls"] += 1
        a["durs"].append(dur)
        a["sum"] += dur
        a["min"] = min(a["min"], dur)
        a["max"] = max(a["max"], dur)
    out = []
    for (name, grid, wg), a in sorted(agg.items(), key=lambda kv: -kv[1]["sum"]):
        rec = {"kernel": name, "grid_threads": grid, "workgroup": wg, "calls": a["calls"],
               "avg_us": a["sum"] / a["calls"] / 1e3, "median_us": sorted(a["durs"])[len(a["durs"]) // 2] / 1e3,
               "min_us": a["min"] / 1e3, "max_us": a["max"] / 1e3, "total_us": a["sum"] / 1e3,
               "vgpr_rocprof": a["vgpr"], "scratch_bytes_rocprof": a["scratch"], "lds_bytes_rocprof": a["lds"]}
        if resources and name in resources:
            rec["resources"] = resources[name]  # code-object metadata: allocated registers, scratch, LDS
        out.append(rec)
    return out


def pmc(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
    # one row per (dispatch, counter instance): sum the instances of a dispatch first, then take the
    # MEDIAN over the dispatches of a (kernel, grid) -- the timed launches of a bench run are identical,
    # while a one-off launch of the same kernel with another batch (the comb-table build runs the
    # variable-base kernel over windows x 65536 units) would skew an average
    return _pmc_median(db)


def _pmc_median(db):
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)").fetchall()]
    key = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "rowid")
    dur = "duration" if "duration" in cols else "0"
    rows = db.execute(f"select kernel_name, grid_size, counter_name, value, {key}, {dur} from counters_collection").fetchall()
    per, durs = {}, {}
    for name, grid, counter, value, disp, d in rows:
        per.setdefault((short(name), grid, counter), {}).setdefault(disp, 0.0)
        per[(short(name), grid, counter)][disp] += value
        durs[(short(name), grid, disp)] = d
    out = {}
    for (name, grid, counter), by_disp in per.items():
        vals = sorted(by_disp.values())
        out.setdefault(f"{name} grid={grid}", {})[counter] = {
            "dispatches": len(vals), "avg_per_dispatch": vals[len(vals) // 2], "mean_per_dispatch": sum(vals) / len(vals),
            "min_per_dispatch": vals[0], "max_per_dispatch": vals[-1], "statistic": "median over dispatches"}
    # effective clock and cycles per vector instruction, dispatch by dispatch, from the counters of ONE pass
    for (name, grid, counter), by_disp in per.items():
        if counter != "GRBM_GUI_ACTIVE":
            continue
        clocks, cpi, dd = [], [], []
        valu = per.get((name, grid, "SQ_INSTS_VALU"), {})
        for disp, gui in by_disp.items():
            d_ns = durs.get((name, grid, disp)) or 0
            if d_ns > 0:
                clocks.append(gui / XCDS / d_ns)  # cycles per ns = GHz
                dd.append(d_ns / 1e3)
            if valu.get(disp):
                cpi.append((gui / XCDS) / (valu[disp] / SIMDS))
        rec = out[f"{name} grid={grid}"]
        if clocks:
            clocks.sort(); dd.sort()
            rec["effective_clock_ghz"] = {"median": clocks[len(clocks) // 2], "min": clocks[0], "max": clocks[-1], "dispatches": len(clocks),
                                          "how": "GRBM_GUI_ACTIVE / 8 / duration of the same dispatch"}
            rec["duration_us_in_this_pass"] = {"median": dd[len(dd) // 2], "min": dd[0], "max": dd[-1]}
        if cpi:
            cpi.sort()
            rec["cycles_per_valu_inst"] = {"median": cpi[len(cpi) // 2], "min": cpi[0], "max": cpi[-1],
                                           "how": "(GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024 SIMDs), per dispatch"}
    return out


def main():
    root = sys.argv[1]
    res = {"source": os.path.basename(os.path.normpath(root)), "kernel_trace": None, "pmc": {}}
    resources = _resources()
    for f in glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True):
        res["kernel_trace"] = [r for r in trace(f, resources) if r["total_us"] > 50]
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            p = pmc(f)
            for k, v in p.items():
                if "eccx::" in k:
                    res["pmc"].setdefault(k, {}).update(v)
    # derived HBM traffic for the engine's kernels
    for k, v in res["pmc"].items():
        if "FETCH_SIZE" in v:
            kb = v["FETCH_SIZE"]["avg_per_dispatch"]
            v["hbm_read_bytes_per_dispatch_raw"] = kb * 1024
            v["hbm_read_bytes_per_dispatch_x2_gfx950_correction"] = kb * 1024 * 2
        if "WRITE_SIZE" in v:
            v["hbm_write_bytes_per_dispatch"] = v["WRITE_SIZE"]["avg_per_dispatch"] * 1024
    json.dump(res, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
