"""ctypes wrapper of the CPU oracle (oracle/liboracle.so) for tests, smoke() and the
bench's cpu_baseline leg.  Never imported by the product package."""
from __future__ import annotations

import ctypes
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CURVES = ["p256r1", "p384r1", "p521r1", "bls12_381_g1", "ed25519"]
CURVE_ID = {n: i for i, n in enumerate(CURVES)}


def golden(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.eccx_oracle_scalarmul_var.restype = ctypes.c_int
        lib.eccx_oracle_scalarmul_var.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.eccx_oracle_scalarmul_base.restype = ctypes.c_int
        lib.eccx_oracle_scalarmul_base.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.eccx_oracle_comb_table.restype = ctypes.c_int
        lib.eccx_oracle_comb_table.argtypes = [ctypes.c_int, ctypes.c_void_p]

    def fb(self, curve) -> int:
        return self.lib.eccx_oracle_field_bytes(CURVE_ID[curve])

    def sb(self, curve) -> int:
        return self.lib.eccx_oracle_scalar_bytes(CURVE_ID[curve])

    def _pw(self, curve) -> int:
        return (4 if curve == "ed25519" else 3) * self.fb(curve)

    def var(self, curve, scalars: bytes, points: bytes, threads: int = 8):
        """-> (affine bytes, flags, projective bytes)"""
        n = len(scalars) // self.sb(curve)
        assert len(points) == n * 2 * self.fb(curve)
        out = ctypes.create_string_buffer(max(1, n * 2 * self.fb(curve)))
        inf = ctypes.create_string_buffer(max(1, n))
        proj = ctypes.create_string_buffer(max(1, n * self._pw(curve)))
        if n:
            rc = self.lib.eccx_oracle_scalarmul_var(CURVE_ID[curve], n, scalars, points, out, inf, proj, threads)
            assert rc == 0
        return out.raw[: n * 2 * self.fb(curve)], inf.raw[:n], proj.raw[: n * self._pw(curve)]

    def base(self, curve, scalars: bytes, threads: int = 8):
        n = len(scalars) // self.sb(curve)
        out = ctypes.create_string_buffer(max(1, n * 2 * self.fb(curve)))
        inf = ctypes.create_string_buffer(max(1, n))
        proj = ctypes.create_string_buffer(max(1, n * self._pw(curve)))
        if n:
            rc = self.lib.eccx_oracle_scalarmul_base(CURVE_ID[curve], n, scalars, out, inf, proj, threads)
            assert rc == 0
        return out.raw[: n * 2 * self.fb(curve)], inf.raw[:n], proj.raw[: n * self._pw(curve)]

    def x25519(self, scalars: bytes, u=None, rfc: bool = True, threads: int = 8):
        """-> (n x 32 little-endian u-coordinates, flags)"""
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(1, n * 32))
        fl = ctypes.create_string_buffer(max(1, n))
        f = self.lib.eccx_oracle_x25519
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_int, ctypes.c_int]
        if n:
            assert f(n, scalars, u, out, fl, 1 if rfc else 0, threads) == 0
        return out.raw[: n * 32], fl.raw[:n]

    def comb_table(self, curve) -> bytes:
        size = 2 * self.sb(curve) * 15 * 2 * self.fb(curve)
        out = ctypes.create_string_buffer(size)
        assert self.lib.eccx_oracle_comb_table(CURVE_ID[curve], out) == 0
        return out.raw


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        # ECCX_ORACLE_SAN=1: the AddressSanitizer + UBSan build (tools/run_sanitizers.sh preloads the runtimes)
        san = os.environ.get("ECCX_ORACLE_SAN", "0") == "1"
        so = os.path.join(ORACLE_DIR, "liboracle_san.so" if san else "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["SAN=1"] if san else []))
        _cached = Oracle(ctypes.CDLL(so))
    return _cached
