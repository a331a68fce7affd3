"""Thin Python host layer over the C ABI: a context per GPU, host-bytes and
device-tensor entry points.  PyTorch is used only as plumbing (device memory, streams).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

from . import _lib

# curve ids (include/eccx.h: eccx_curve)
P256R1, P384R1, P521R1, BLS12_381_G1, ED25519 = 0, 1, 2, 3, 4
CURVE_IDS = {"p256r1": P256R1, "p384r1": P384R1, "p521r1": P521R1, "bls12_381_g1": BLS12_381_G1, "ed25519": ED25519}
CURVE_NAMES = {v: k for k, v in CURVE_IDS.items()}

VALIDATE_POINTS = 1 << 0
MIRROR_REFERENCE = 1 << 1
TABLE_IN_LDS = 1 << 2
TABLE_IN_L2 = 1 << 3
X25519_RAW_LADDER = 1 << 4
SUBTRACT = 1 << 5
CHECK_SUBGROUP = 1 << 6
UNCOMPRESSED = 1 << 7
CT_SCAN = 1 << 8
ASSUME_SUBGROUP = 1 << 9
CT_GATHER = 1 << 10
OUT_X_ONLY = 1 << 11
PREP_VAR, PREP_BASE, PREP_BASE_LDS, PREP_MIRROR, PREP_CT, PREP_CT_GATHER, PREP_HOST = 1, 2, 4, 8, 16, 32, 64
FLAG_FINITE, FLAG_INFINITY, FLAG_REJECTED = 0, 1, 2


class EccxError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        lib = _lib.load()
        msg = lib.eccx_strerror(code).decode()
        super().__init__(f"eccx error {code} ({msg}){': ' + detail if detail else ''}")
        self.code = code


def curve_id(curve) -> int:
    if isinstance(curve, str):
        return CURVE_IDS[curve]
    return int(curve)


def field_bytes(curve) -> int:
    r = _lib.load().eccx_field_bytes(curve_id(curve))
    if r < 0:
        raise EccxError(r)
    return r


def scalar_bytes(curve) -> int:
    r = _lib.load().eccx_scalar_bytes(curve_id(curve))
    if r < 0:
        raise EccxError(r)
    return r


def _proj_width(cid: int) -> int:
    return (4 if cid == ED25519 else 3) * field_bytes(cid)


class Engine:
    """One engine context bound to one GPU (eccx_init / eccx_shutdown)."""

    def __init__(self, device: int = 0, secret_scalars: bool = False):
        """secret_scalars: what `ct_scan=None` means in the scalar-multiplication calls of this engine -- True selects the
        secret-scalar kernels (ECCX_CT_SCAN: every table entry read at every lookup, no branch on scalar-derived data;
        include/eccx.h "SIDE CHANNELS") for key generation, signing and ECDH; False (verification, public keys, tests) the
        faster public-scalar kernels, whose lookups and branches follow the digits.  The C++ and Rust wrappers take the
        same choice with no default (eccx::Secrecy, eccoxide_gpu::Secrecy)."""
        self.secret_scalars = bool(secret_scalars)
        self._lib = _lib.load()
        self._ctx = ctypes.c_void_p()
        rc = self._lib.eccx_init(int(device), ctypes.byref(self._ctx))
        if rc != 0:
            raise EccxError(rc, f"eccx_init(device={device}) failed: no usable HIP device?")
        self.device = int(device)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.eccx_shutdown(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _secret(self, ct_scan: Optional[bool]) -> bool:
        return self.secret_scalars if ct_scan is None else bool(ct_scan)

    def _check(self, rc: int):
        if rc != 0:
            raise EccxError(rc, self._lib.eccx_last_error(self._ctx).decode())

    def _tensors(self, n: int, *specs):
        """Validate the device tensors of a `_t` call before their raw addresses reach a kernel:
        each spec is (name, tensor or None, bytes per unit).  Every tensor must be a contiguous
        torch.uint8 CUDA tensor ON THIS ENGINE'S GPU holding exactly n units -- a short or foreign
        buffer would otherwise be an out-of-bounds or cross-device access inside the kernel."""
        import torch

        for name, t, width in specs:
            if t is None:
                continue
            if not (getattr(t, "is_cuda", False) and t.dtype == torch.uint8 and t.is_contiguous()):
                raise ValueError(f"{name}: tensors must be contiguous torch.uint8 CUDA tensors")
            if t.device.index != self.device:
                raise ValueError(f"{name}: tensor lives on cuda:{t.device.index}, this engine is bound to cuda:{self.device}")
            if t.numel() != n * width:
                raise ValueError(f"{name}: expected {n} x {width} bytes, got {t.numel()}")

    @staticmethod
    def _units(t, width: int, name: str) -> int:
        if t.numel() % width:
            raise ValueError(f"{name}: {t.numel()} bytes is not a multiple of the {width}-byte unit")
        return t.numel() // width

    # ---- one-time costs ------------------------------------------------------
    def prepare(self, curve, *, base: bool = True, base_lds: bool = False, ct: bool = False, ct_gather: bool = False):
        """eccx_prepare: build the fixed-base tables of `curve` now (blocking); ct: the signed-window table
        of the secret-scalar (ECCX_CT_SCAN) fixed-base kernel."""
        self._check(self._lib.eccx_prepare(self._ctx, curve_id(curve),
                                           (PREP_BASE if base else 0) | (PREP_BASE_LDS if base_lds else 0)
                                           | (PREP_CT if ct else 0) | (PREP_CT_GATHER if ct_gather else 0)))

    def reserve(self, curve, max_n: int, *, var: bool = True, mirror: bool = False, ct: bool = False, host: bool = False):
        """eccx_reserve: size the scratch slab and row buffer for batches of up to max_n units; ct: for the
        secret-scalar (ECCX_CT_SCAN) variable-base ladder."""
        self._check(self._lib.eccx_reserve(self._ctx, curve_id(curve), int(max_n),
                                           (PREP_VAR if var else 0) | (PREP_MIRROR if mirror else 0)
                                           | (PREP_CT if ct else 0) | (PREP_HOST if host else 0)))

    def device_bytes(self) -> int:
        return int(self._lib.eccx_device_bytes(self._ctx))

    # ---- host buffers ------------------------------------------------------
    def scalarmul_var(self, curve, scalars: bytes, points: bytes, *, validate: bool = False,
                      want_proj: bool = False, mirror: bool = False, ct_scan: Optional[bool] = None,
                      assume_subgroup: bool = False):
        """out[i] = scalars[i] * points[i]; returns (affine bytes, flags[, proj bytes]).
        mirror=True (implied by want_proj) runs the reference-mirroring kernels; ct_scan=True the secret-scalar
        ladder (every table row read at every lookup, selects only; None: the engine's secret_scalars); assume_subgroup=True (bls12_381_g1) the
        endomorphism ladder for bases known to be in G1."""
        cid = curve_id(curve)
        sb, fb = scalar_bytes(cid), field_bytes(cid)
        if len(scalars) % sb:
            raise ValueError("scalars length is not a multiple of the scalar size")
        n = len(scalars) // sb
        if len(points) != n * 2 * fb:
            raise ValueError("points length does not match the number of scalars")
        out = ctypes.create_string_buffer(max(1, n * 2 * fb))
        flags = ctypes.create_string_buffer(max(1, n))
        proj = ctypes.create_string_buffer(max(1, n * _proj_width(cid))) if want_proj else None
        rc = self._lib.eccx_scalarmul_var(self._ctx, cid, n, scalars, points, out, flags, proj,
                                          (VALIDATE_POINTS if validate else 0) | (MIRROR_REFERENCE if mirror else 0)
                                          | (CT_SCAN if self._secret(ct_scan) else 0) | (ASSUME_SUBGROUP if assume_subgroup else 0))
        self._check(rc)
        res = (out.raw[: n * 2 * fb], flags.raw[:n])
        return res + (proj.raw[: n * _proj_width(cid)],) if want_proj else res

    def scalarmul_base(self, curve, scalars: bytes, *, want_proj: bool = False, mirror: bool = False,
                       ct_scan: Optional[bool] = None, ct_gather: bool = False):
        """out[i] = scalars[i] * G via the fixed-base comb table.
        mirror=True (implied by want_proj) runs the reference-mirroring kernels; ct_scan=True the secret-scalar comb
        (signed windows, every entry of a window read by every lane; None: the engine's secret_scalars); ct_gather=True its
        cross-lane lookup (ECCX_CT_GATHER, opt-in)."""
        cid = curve_id(curve)
        sb, fb = scalar_bytes(cid), field_bytes(cid)
        if len(scalars) % sb:
            raise ValueError("scalars length is not a multiple of the scalar size")
        n = len(scalars) // sb
        out = ctypes.create_string_buffer(max(1, n * 2 * fb))
        flags = ctypes.create_string_buffer(max(1, n))
        proj = ctypes.create_string_buffer(max(1, n * _proj_width(cid))) if want_proj else None
        rc = self._lib.eccx_scalarmul_base(self._ctx, cid, n, scalars, out, flags, proj,
                                           (MIRROR_REFERENCE if mirror else 0) | (CT_SCAN if self._secret(ct_scan) or ct_gather else 0)
                                           | (CT_GATHER if ct_gather else 0))
        self._check(rc)
        res = (out.raw[: n * 2 * fb], flags.raw[:n])
        return res + (proj.raw[: n * _proj_width(cid)],) if want_proj else res

    def point_add(self, curve, a: bytes, b: bytes, *, a_inf: Optional[bytes] = None, b_inf: Optional[bytes] = None,
                  subtract: bool = False, mirror: bool = False):
        """Batched group law out[i] = a[i] + b[i] (a[i] - b[i] with subtract=True) on affine points;
        a_inf / b_inf flag operands that are the point at infinity.  Returns (affine bytes, flags).
        mirror=True runs the saturated-limb kernels instead of the default unsaturated ones."""
        cid = curve_id(curve)
        fb = field_bytes(cid)
        if len(a) != len(b) or len(a) % (2 * fb):
            raise ValueError("a and b must both be n x 2FB bytes")
        n = len(a) // (2 * fb)
        out = ctypes.create_string_buffer(max(1, n * 2 * fb))
        flags = ctypes.create_string_buffer(max(1, n))
        rc = self._lib.eccx_point_add(self._ctx, cid, n, a, a_inf, b, b_inf, out, flags,
                                      (SUBTRACT if subtract else 0) | (MIRROR_REFERENCE if mirror else 0))
        self._check(rc)
        return out.raw[: n * 2 * fb], flags.raw[:n]

    def double_scalarmul(self, curve, u1: bytes, u2: bytes, q: bytes, *, subtract: bool = False,
                         validate: bool = False, x_only: bool = False):
        """out[i] = u1[i]*G + u2[i]*q[i] (minus with subtract=True): the signature-verification
        shape (ECDSA u1*G + u2*Q, Ed25519 [s]B - [k]A).  Returns (affine bytes, flags); x_only=True: the
        x-coordinates alone, FB bytes per unit (what ECDSA verification reads)."""
        cid = curve_id(curve)
        sb, fb = scalar_bytes(cid), field_bytes(cid)
        if len(u1) != len(u2) or len(u1) % sb or len(q) != (len(u1) // sb) * 2 * fb:
            raise ValueError("u1, u2 must be n x SB bytes and q n x 2FB bytes")
        n = len(u1) // sb
        width = fb if x_only else 2 * fb
        out = ctypes.create_string_buffer(max(1, n * width))
        flags = ctypes.create_string_buffer(max(1, n))
        rc = self._lib.eccx_double_scalarmul(self._ctx, cid, n, u1, u2, q, out, flags,
                                             (SUBTRACT if subtract else 0) | (VALIDATE_POINTS if validate else 0)
                                             | (OUT_X_ONLY if x_only else 0))
        self._check(rc)
        return out.raw[: n * width], flags.raw[:n]

    def compressed_bytes(self, curve) -> int:
        """Bytes per compressed point: FB + 1 (SEC1), 48 (zcash G1), 32 (RFC 8032)."""
        return self._lib.eccx_compressed_bytes(curve_id(curve))

    def point_decompress(self, curve, enc: bytes, *, check_subgroup: bool = False, uncompressed: bool = False):
        """Compressed encodings -> (n x 2FB affine x||y, flags): 0 point, 1 infinity encoding, 2 rejected.
        SEC1 for the sec2 curves, zcash for bls12_381_g1 (check_subgroup=True: from_compressed,
        else from_compressed_oncurve_only), RFC 8032 for ed25519."""
        cid = curve_id(curve)
        fb = field_bytes(cid)
        eb = 2 * fb if uncompressed else self.compressed_bytes(cid)
        if len(enc) % eb:
            raise ValueError(f"enc must be n x {eb} bytes")
        n = len(enc) // eb
        out = ctypes.create_string_buffer(max(1, n * 2 * fb))
        flags = ctypes.create_string_buffer(max(1, n))
        self._check(self._lib.eccx_point_decompress(self._ctx, cid, n, enc, out, flags,
                                                    (CHECK_SUBGROUP if check_subgroup else 0) | (UNCOMPRESSED if uncompressed else 0)))
        return out.raw[: n * 2 * fb], flags.raw[:n]

    def point_compress(self, curve, xy: bytes, inf: Optional[bytes] = None, *, uncompressed: bool = False) -> bytes:
        """Affine x||y records (+ optional infinity flags) -> compressed encodings (uncompressed=True:
        the 96-byte zcash flavour of bls12_381_g1)."""
        cid = curve_id(curve)
        fb = field_bytes(cid)
        eb = 2 * fb if uncompressed else self.compressed_bytes(cid)
        if len(xy) % (2 * fb) or (inf is not None and len(inf) != len(xy) // (2 * fb)):
            raise ValueError("xy must be n x 2FB bytes and inf n bytes")
        n = len(xy) // (2 * fb)
        out = ctypes.create_string_buffer(max(1, n * eb))
        self._check(self._lib.eccx_point_compress(self._ctx, cid, n, xy, inf, out, UNCOMPRESSED if uncompressed else 0))
        return out.raw[: n * eb]

    def point_decompress_t(self, curve, enc, out=None, flags=None, *, check_subgroup: bool = False,
                           uncompressed: bool = False, stream: Optional[int] = None):
        """Device-tensor form of point_decompress (torch.uint8 CUDA tensors)."""
        import torch

        cid = curve_id(curve)
        fb = field_bytes(cid)
        eb = 2 * fb if uncompressed else self.compressed_bytes(cid)
        n = self._units(enc, eb, "enc")
        if out is None:
            out = torch.empty((n, 2 * fb), dtype=torch.uint8, device=enc.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=enc.device)
        self._tensors(n, ("enc", enc, eb), ("out", out, 2 * fb), ("flags", flags, 1))
        if stream is None:
            stream = torch.cuda.current_stream(enc.device).cuda_stream
        self._check(self._lib.eccx_point_decompress_dev(self._ctx, cid, n, enc.data_ptr(), out.data_ptr(), flags.data_ptr(),
                                                        (CHECK_SUBGROUP if check_subgroup else 0)
                                                        | (UNCOMPRESSED if uncompressed else 0), stream))
        return out, flags

    def point_compress_t(self, curve, xy, inf=None, out=None, *, uncompressed: bool = False, stream: Optional[int] = None):
        """Device-tensor form of point_compress (torch.uint8 CUDA tensors)."""
        import torch

        cid = curve_id(curve)
        fb = field_bytes(cid)
        eb = 2 * fb if uncompressed else self.compressed_bytes(cid)
        n = self._units(xy, 2 * fb, "xy")
        if out is None:
            out = torch.empty((n, eb), dtype=torch.uint8, device=xy.device)
        self._tensors(n, ("xy", xy, 2 * fb), ("inf", inf, 1), ("out", out, eb))
        if stream is None:
            stream = torch.cuda.current_stream(xy.device).cuda_stream
        self._check(self._lib.eccx_point_compress_dev(self._ctx, cid, n, xy.data_ptr(),
                                                      inf.data_ptr() if inf is not None else None, out.data_ptr(),
                                                      UNCOMPRESSED if uncompressed else 0, stream))
        return out

    def x25519(self, scalars: bytes, u: Optional[bytes] = None, *, raw_ladder: bool = False):
        """X25519 over a batch: returns (n x 32 little-endian u-coordinates, flags).
        Default: RFC 7748 semantics (protocol::x25519::x25519): little-endian scalars, clamped;
        raw_ladder=True: MontgomeryPoint::scale_bytes, big-endian scalars used as given.
        u=None multiplies the base point u = 9."""
        if len(scalars) % 32 or (u is not None and len(u) != len(scalars)):
            raise ValueError("scalars / u must be n x 32 bytes")
        n = len(scalars) // 32
        out = ctypes.create_string_buffer(max(1, n * 32))
        flags = ctypes.create_string_buffer(max(1, n))
        rc = self._lib.eccx_x25519(self._ctx, n, scalars, u, out, flags, X25519_RAW_LADDER if raw_ladder else 0)
        self._check(rc)
        return out.raw[: n * 32], flags.raw[:n]

    def point_add_t(self, curve, a, b, out=None, flags=None, *, a_inf=None, b_inf=None, subtract: bool = False,
                    mirror: bool = False, stream: Optional[int] = None):
        """Device-tensor form of point_add (torch.uint8 CUDA tensors): out[i] = a[i] +- b[i]."""
        import torch

        cid = curve_id(curve)
        fb = field_bytes(cid)
        n = self._units(a, 2 * fb, "a")
        if out is None:
            out = torch.empty((n, 2 * fb), dtype=torch.uint8, device=a.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=a.device)
        self._tensors(n, ("a", a, 2 * fb), ("b", b, 2 * fb), ("a_inf", a_inf, 1), ("b_inf", b_inf, 1),
                      ("out", out, 2 * fb), ("flags", flags, 1))
        if stream is None:
            stream = torch.cuda.current_stream(a.device).cuda_stream
        rc = self._lib.eccx_point_add_dev(self._ctx, cid, n, a.data_ptr(), a_inf.data_ptr() if a_inf is not None else None,
                                          b.data_ptr(), b_inf.data_ptr() if b_inf is not None else None,
                                          out.data_ptr(), flags.data_ptr(),
                                          (SUBTRACT if subtract else 0) | (MIRROR_REFERENCE if mirror else 0), stream)
        self._check(rc)
        return out, flags

    def double_scalarmul_t(self, curve, u1, u2, q, out=None, flags=None, *, subtract: bool = False,
                           validate: bool = False, x_only: bool = False, stream: Optional[int] = None):
        """Device-tensor form of double_scalarmul (torch.uint8 CUDA tensors): out[i] = u1[i]*G +- u2[i]*q[i]."""
        import torch

        cid = curve_id(curve)
        fb, sb = field_bytes(cid), scalar_bytes(cid)
        n = self._units(u1, sb, "u1")
        width = fb if x_only else 2 * fb
        if out is None:
            out = torch.empty((n, width), dtype=torch.uint8, device=u1.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=u1.device)
        self._tensors(n, ("u1", u1, sb), ("u2", u2, sb), ("q", q, 2 * fb), ("out", out, width), ("flags", flags, 1))
        if stream is None:
            stream = torch.cuda.current_stream(u1.device).cuda_stream
        rc = self._lib.eccx_double_scalarmul_dev(self._ctx, cid, n, u1.data_ptr(), u2.data_ptr(), q.data_ptr(),
                                                 out.data_ptr(), flags.data_ptr(),
                                                 (SUBTRACT if subtract else 0) | (VALIDATE_POINTS if validate else 0)
                                                 | (OUT_X_ONLY if x_only else 0), stream)
        self._check(rc)
        return out, flags

    def x25519_t(self, scalars, u=None, out=None, flags=None, *, raw_ladder: bool = False,
                 stream: Optional[int] = None):
        """Device-tensor form of x25519 (torch.uint8 CUDA tensors, n x 32)."""
        import torch

        n = self._units(scalars, 32, "scalars")
        if out is None:
            out = torch.empty((n, 32), dtype=torch.uint8, device=scalars.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=scalars.device)
        self._tensors(n, ("scalars", scalars, 32), ("u", u, 32), ("out", out, 32), ("flags", flags, 1))
        if stream is None:
            stream = torch.cuda.current_stream(scalars.device).cuda_stream
        rc = self._lib.eccx_x25519_dev(self._ctx, n, scalars.data_ptr(), u.data_ptr() if u is not None else None,
                                       out.data_ptr(), flags.data_ptr(), X25519_RAW_LADDER if raw_ladder else 0, stream)
        self._check(rc)
        return out, flags

    def comb_table(self, curve) -> bytes:
        """The fixed-base table in the reference's on-disk layout (NW x 15 x (x||y))."""
        cid = curve_id(curve)
        size = 2 * scalar_bytes(cid) * 15 * 2 * field_bytes(cid)
        out = ctypes.create_string_buffer(size)
        self._check(self._lib.eccx_comb_table(self._ctx, cid, out))
        return out.raw

    # ---- device tensors (torch.uint8, resident on this engine's GPU) -----------
    def scalarmul_var_t(self, curve, scalars, points, out=None, flags=None, proj=None, *,
                        validate: bool = False, mirror: bool = False, ct_scan: Optional[bool] = None,
                        assume_subgroup: bool = False, stream: Optional[int] = None):
        """Device-resident variant: tensors are torch.uint8 CUDA tensors; the launch is
        enqueued on `stream` (raw hipStream_t handle; default: torch's current stream)."""
        import torch

        cid = curve_id(curve)
        sb, fb = scalar_bytes(cid), field_bytes(cid)
        n = self._units(scalars, sb, "scalars")
        if out is None:
            out = torch.empty((n, 2 * fb), dtype=torch.uint8, device=scalars.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=scalars.device)
        self._tensors(n, ("scalars", scalars, sb), ("points", points, 2 * fb), ("out", out, 2 * fb), ("flags", flags, 1),
                      ("proj", proj, _proj_width(cid)))
        if points is None:
            raise ValueError("points: required")
        if stream is None:
            stream = torch.cuda.current_stream(scalars.device).cuda_stream
        rc = self._lib.eccx_scalarmul_var_dev(self._ctx, cid, n, scalars.data_ptr(), points.data_ptr(),
                                              out.data_ptr(), flags.data_ptr(),
                                              proj.data_ptr() if proj is not None else None,
                                              (VALIDATE_POINTS if validate else 0) | (MIRROR_REFERENCE if mirror else 0)
                                              | (CT_SCAN if self._secret(ct_scan) else 0) | (ASSUME_SUBGROUP if assume_subgroup else 0),
                                              stream)
        self._check(rc)
        return out, flags

    def scalarmul_base_t(self, curve, scalars, out=None, flags=None, proj=None, *, stream: Optional[int] = None,
                         table_in_lds: Optional[bool] = None, mirror: bool = False, ct_scan: Optional[bool] = None,
                         ct_gather: bool = False):
        import torch

        cid = curve_id(curve)
        sb, fb = scalar_bytes(cid), field_bytes(cid)
        n = self._units(scalars, sb, "scalars")
        if out is None:
            out = torch.empty((n, 2 * fb), dtype=torch.uint8, device=scalars.device)
        if flags is None:
            flags = torch.empty((n,), dtype=torch.uint8, device=scalars.device)
        self._tensors(n, ("scalars", scalars, sb), ("out", out, 2 * fb), ("flags", flags, 1),
                      ("proj", proj, _proj_width(cid)))
        if stream is None:
            stream = torch.cuda.current_stream(scalars.device).cuda_stream
        rc = self._lib.eccx_scalarmul_base_dev(self._ctx, cid, n, scalars.data_ptr(), out.data_ptr(),
                                               flags.data_ptr(), proj.data_ptr() if proj is not None else None,
                                               (0 if table_in_lds is None else (TABLE_IN_LDS if table_in_lds else TABLE_IN_L2))
                                               | (MIRROR_REFERENCE if mirror else 0) | (CT_SCAN if self._secret(ct_scan) or ct_gather else 0)
                                               | (CT_GATHER if ct_gather else 0),
                                               stream)
        self._check(rc)
        return out, flags
