"""bench.py's static side, without a GPU: every workload's `roofline.traffic` resolves from a committed
rocprofv3 summary (no typed-in constants), the kernels it names exist in that summary, and the
multiplier-count model is self-consistent.  CPU only."""
import importlib.util
import json
import os

import pytest

from tests.oracle_lib import ROOT

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

PROFILED = ["p256r1_var_2^20", "p384r1_var_2^19", "p521r1_var_2^19", "bls12_381_g1_var_2^20", "ed25519_var_2^20",
            "p256r1_verify_2^20", "ed25519_base_2^20", "p256r1_base_2^20", "x25519_2^20"]


@pytest.mark.parametrize("workload", PROFILED)
def test_traffic_comes_from_the_committed_profile(workload):
    curve, op, n, alg_bytes, mult = bench.WORKLOADS[workload]
    t = bench.measured_traffic(workload, curve, op, "default")
    assert t is not None, f"no usable profile for {workload}"
    assert os.path.exists(os.path.join(ROOT, t["source"])) and any(t["source"].startswith("profiles/" + r) for r in bench.PROFILE_ROUNDS)
    assert t["bytes"] == 2 * t["fetch_raw"] + t["write"]          # the gfx950 FETCH_SIZE correction, once
    assert t["bytes"] > alg_bytes * n                              # counters see at least the algorithmic bytes
    prof = json.load(open(os.path.join(ROOT, t["source"])))
    names = {k["kernel"] for k in prof["kernel_trace"]}
    for key in t["kernels"]:
        kernel = key.split(" grid=")[0]
        assert kernel in names, f"{kernel} has counters but no kernel-trace line"
        row = next(k for k in prof["kernel_trace"] if k["kernel"] == kernel)
        assert {"median_us", "calls"} <= set(row)
        if t["source"].startswith("profiles/r03"):
            # the resource line comes from the code object's metadata (rocprofv3's vgpr_count reads half on gfx950)
            assert {"vgpr", "sgpr", "scratch_bytes", "lds_bytes", "waves_per_simd_by_registers"} <= set(row["resources"])
            assert row["resources"]["vgpr"] in (2 * row["vgpr_rocprof"], row["vgpr_rocprof"]) or row["resources"]["vgpr"] >= row["vgpr_rocprof"]
        else:
            assert {"vgpr", "scratch_bytes", "lds_bytes"} <= set(row)


@pytest.mark.parametrize("workload,variant", [("bls12_381_g1_var_2^20", "glv"), ("ed25519_base_2^20", "lds")])
def test_variant_profiles_resolve(workload, variant):
    curve, op, n, alg_bytes, mult = bench.WORKLOADS[workload]
    t = bench.measured_traffic(workload, curve, op, variant)
    assert t is not None and variant in t["source"]


@pytest.mark.parametrize("workload", PROFILED)
def test_clock_comes_from_the_committed_profile(workload):
    """The VALU roofline is priced at the clock the chip HELD (GRBM_GUI_ACTIVE / 8 / duration of the same dispatch),
    read from the workload's rocprofv3 summary -- not at the nominal 2.4 GHz."""
    curve, op, n, alg_bytes, mult = bench.WORKLOADS[workload]
    t = bench.measured_traffic(workload, curve, op, "default")
    c = bench.measured_clock(workload, curve, op, "default")
    if t["source"].startswith("profiles/r03"):
        assert c is not None and 1.6e9 < c["hz"] < 2.45e9 and c["source"] == t["source"]
        assert c["cycles_per_valu_inst"] is None or 2.0 < c["cycles_per_valu_inst"] < 12.0
    else:
        assert c is None  # round 2's summaries carry no per-dispatch clock: bench.py then says "nominal"


def test_multi_gpu_launch_happens_before_any_gpu_call():
    """`python bench.py --gpus N` without WORLD_SIZE starts torch.distributed.run as a CHILD and exits with its code; on
    this pool a process that has initialised the GPU must not spawn-and-replace, and the ranks must not inherit an
    initialised runtime: the spawn has to stay in front of every torch / engine import of main()."""
    import ast

    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    spawn_at = gpu_at = None
    for i, stmt in enumerate(main.body):
        src = ast.unparse(stmt)
        if spawn_at is None and "torch.distributed.run" in src and "subprocess.call" in src and "sys.exit" in src:
            spawn_at = i
        touches_gpu = any(isinstance(n, (ast.Import, ast.ImportFrom)) and any(
            (a.name or "").split(".")[0] in ("torch", "eccoxide_amd") for a in n.names) or
            (isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] in ("torch", "eccoxide_amd", "tests"))
            for n in ast.walk(stmt))
        if gpu_at is None and touches_gpu:
            gpu_at = i
    assert spawn_at is not None and gpu_at is not None and spawn_at < gpu_at
    # and nothing at module level imports them either
    for n in tree.body:
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            names = [a.name.split(".")[0] for a in n.names] + [((n.module or "") if isinstance(n, ast.ImportFrom) else "").split(".")[0]]
            assert not ({"torch", "eccoxide_amd"} & set(names))


def test_unprofiled_workload_reports_null_traffic():
    curve, op, n, alg_bytes, mult = bench.WORKLOADS["p384r1_base_2^19"]
    assert bench.measured_traffic("p384r1_base_2^19", curve, op, "default") is None


def test_multiplier_model():
    m = bench.WORKLOADS["p256r1_var_2^20"][4]["mad"]
    # 9-limb P-256: product 81 + 36, square 45 + 36; doubling 4 + 4, mixed addition 8 + 3, co-Z addition 5 + 2
    # (the denominator tracked), rescaling 4 + 1 per entry, 2P from the affine P 2 + 4
    dbl, madd, zaddu, scale = 4 * 117 + 4 * 81, 8 * 117 + 3 * 81, 5 * 117 + 2 * 81, 4 * 117 + 81
    assert dbl == 792 and madd == 1179 and zaddu == 747
    ladder = 51 * 5 * dbl + 51 * madd + (2 * 117 + 4 * 81) + 14 * zaddu + 15 * scale
    inversion = 20 * (8 * 9 + 2 * 7)                    # 20 batches of 30 division steps (delta = 1/2 variant)
    assert 0 < m - ladder - inversion < 3500            # the rest: conversions, entry 16 and the normalisation's products
    # the generic ladder (table of Jacobian entries), as the fused verification kernel still runs it
    g = bench._var_generic(9, 4, 32)["mad"]
    assert 0 < g - ((8 + 51 * 5) * dbl + (7 + 51) * (11 * 117 + 3 * 81) + 15 * (117 + 81)) < 3000
    assert m < g
    glv = bench.VARIANT_MULT[("bls12_381_g1_var_2^20", "glv")]["mad"]
    assert glv < 0.7 * bench.WORKLOADS["bls12_381_g1_var_2^20"][4]["mad"]


def test_usable_cores_is_sane():
    n = bench._usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
