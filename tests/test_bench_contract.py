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
    assert os.path.exists(os.path.join(ROOT, t["source"])) and t["source"].startswith("profiles/" + bench.PROFILE_ROUND)
    assert t["bytes"] == 2 * t["fetch_raw"] + t["write"]          # the gfx950 FETCH_SIZE correction, once
    assert t["bytes"] > alg_bytes * n                              # counters see at least the algorithmic bytes
    prof = json.load(open(os.path.join(ROOT, t["source"])))
    names = {k["kernel"] for k in prof["kernel_trace"]}
    for key in t["kernels"]:
        kernel = key.split(" grid=")[0]
        assert kernel in names, f"{kernel} has counters but no kernel-trace line"
        row = next(k for k in prof["kernel_trace"] if k["kernel"] == kernel)
        assert {"vgpr", "scratch_bytes", "lds_bytes", "median_us", "calls"} <= set(row)   # the resource line


@pytest.mark.parametrize("workload,variant", [("bls12_381_g1_var_2^20", "glv"), ("ed25519_base_2^20", "lds")])
def test_variant_profiles_resolve(workload, variant):
    curve, op, n, alg_bytes, mult = bench.WORKLOADS[workload]
    t = bench.measured_traffic(workload, curve, op, variant)
    assert t is not None and variant in t["source"]


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
