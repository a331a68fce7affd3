//! `extern "C"` declarations of `include/eccx.h`, one to one.
//!
//! `tools/check_rust_ffi.py` (run by the CPU test suite) parses the header and this file and fails
//! on any difference in names, arity, argument or return types, or constant values: this file
//! cannot drift from the header silently even where no Rust toolchain is at hand.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_void};

/// Opaque engine context (`struct eccx_ctx`).
#[repr(C)]
pub struct eccx_ctx {
    _private: [u8; 0],
}

// eccx_curve
pub const ECCX_P256R1: c_int = 0;
pub const ECCX_P384R1: c_int = 1;
pub const ECCX_P521R1: c_int = 2;
pub const ECCX_BLS12_381_G1: c_int = 3;
pub const ECCX_ED25519: c_int = 4;

// status codes
pub const ECCX_OK: c_int = 0;
pub const ECCX_ERR_CURVE: c_int = -1;
pub const ECCX_ERR_ARG: c_int = -2;
pub const ECCX_ERR_HIP: c_int = -3;
pub const ECCX_ERR_NOMEM: c_int = -4;

// option bits
pub const ECCX_VALIDATE_POINTS: u32 = 1 << 0;
pub const ECCX_MIRROR_REFERENCE: u32 = 1 << 1;
pub const ECCX_TABLE_IN_LDS: u32 = 1 << 2;
pub const ECCX_TABLE_IN_L2: u32 = 1 << 3;
pub const ECCX_X25519_RAW_LADDER: u32 = 1 << 4;
pub const ECCX_SUBTRACT: u32 = 1 << 5;
pub const ECCX_CHECK_SUBGROUP: u32 = 1 << 6;
pub const ECCX_UNCOMPRESSED: u32 = 1 << 7;
pub const ECCX_CT_SCAN: u32 = 1 << 8;
pub const ECCX_ASSUME_SUBGROUP: u32 = 1 << 9;
pub const ECCX_CT_GATHER: u32 = 1 << 10;
pub const ECCX_OUT_X_ONLY: u32 = 1 << 11;

// eccx_prepare / eccx_reserve
pub const ECCX_PREP_VAR: u32 = 1 << 0;
pub const ECCX_PREP_BASE: u32 = 1 << 1;
pub const ECCX_PREP_BASE_LDS: u32 = 1 << 2;
pub const ECCX_PREP_MIRROR: u32 = 1 << 3;
pub const ECCX_PREP_HOST: u32 = 1 << 6;
pub const ECCX_PREP_CT_GATHER: u32 = 1 << 5;
pub const ECCX_PREP_HOST: u32 = 1 << 6;
pub const ECCX_PREP_CT_GATHER: u32 = 1 << 5;
pub const ECCX_PREP_CT: u32 = 1 << 4; // ECCX_CT_SCAN: the secret-scalar fixed-base table / variable-base slabs

// per-unit flags
pub const ECCX_FLAG_FINITE: u8 = 0;
pub const ECCX_FLAG_INFINITY: u8 = 1;
pub const ECCX_FLAG_REJECTED: u8 = 2;

#[link(name = "eccx")]
extern "C" {
    pub fn eccx_field_bytes(curve: c_int) -> c_int;
    pub fn eccx_scalar_bytes(curve: c_int) -> c_int;
    pub fn eccx_init(device: c_int, out_ctx: *mut *mut eccx_ctx) -> c_int;
    pub fn eccx_shutdown(ctx: *mut eccx_ctx);
    pub fn eccx_last_error(ctx: *const eccx_ctx) -> *const c_char;
    pub fn eccx_strerror(code: c_int) -> *const c_char;
    pub fn eccx_prepare(ctx: *mut eccx_ctx, curve: c_int, what: u32) -> c_int;
    pub fn eccx_reserve(ctx: *mut eccx_ctx, curve: c_int, max_n: usize, what: u32) -> c_int;
    pub fn eccx_device_bytes(ctx: *const eccx_ctx) -> usize;

    // out[i] = scalars[i] * points[i]          impl Mul<&Scalar> for &Point
    pub fn eccx_scalarmul_var(ctx: *mut eccx_ctx, curve: c_int, n: usize, scalars: *const u8, points: *const u8,
                              out: *mut u8, flags: *mut u8, proj: *mut u8, opts: u32) -> c_int;
    // out[i] = scalars[i] * G                  Point::mul_base
    pub fn eccx_scalarmul_base(ctx: *mut eccx_ctx, curve: c_int, n: usize, scalars: *const u8, out: *mut u8,
                               flags: *mut u8, proj: *mut u8, opts: u32) -> c_int;
    pub fn eccx_scalarmul_var_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_scalars: *const c_void,
                                  d_points: *const c_void, d_out: *mut c_void, d_flags: *mut c_void,
                                  d_proj: *mut c_void, opts: u32, stream: *mut c_void) -> c_int;
    pub fn eccx_scalarmul_base_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_scalars: *const c_void,
                                   d_out: *mut c_void, d_flags: *mut c_void, d_proj: *mut c_void, opts: u32,
                                   stream: *mut c_void) -> c_int;

    // out[i] = a[i] + b[i] (ECCX_SUBTRACT: a - b)        impl Add / Sub / Neg, CurveGroup::double
    pub fn eccx_point_add(ctx: *mut eccx_ctx, curve: c_int, n: usize, a: *const u8, a_inf: *const u8, b: *const u8,
                          b_inf: *const u8, out: *mut u8, flags: *mut u8, opts: u32) -> c_int;
    pub fn eccx_point_add_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_a: *const c_void, d_a_inf: *const c_void,
                              d_b: *const c_void, d_b_inf: *const c_void, d_out: *mut c_void, d_flags: *mut c_void,
                              opts: u32, stream: *mut c_void) -> c_int;

    // out[i] = u1[i]*G + u2[i]*q[i]                        ecdsa verify / ed25519 verify shape
    pub fn eccx_double_scalarmul(ctx: *mut eccx_ctx, curve: c_int, n: usize, u1: *const u8, u2: *const u8,
                                 q: *const u8, out: *mut u8, flags: *mut u8, opts: u32) -> c_int;
    pub fn eccx_double_scalarmul_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_u1: *const c_void,
                                     d_u2: *const c_void, d_q: *const c_void, d_out: *mut c_void,
                                     d_flags: *mut c_void, opts: u32, stream: *mut c_void) -> c_int;

    // X25519 over a batch                                    protocol::x25519::x25519
    pub fn eccx_x25519(ctx: *mut eccx_ctx, n: usize, scalars: *const u8, u: *const u8, out: *mut u8, flags: *mut u8,
                       opts: u32) -> c_int;
    pub fn eccx_x25519_dev(ctx: *mut eccx_ctx, n: usize, d_scalars: *const c_void, d_u: *const c_void,
                           d_out: *mut c_void, d_flags: *mut c_void, opts: u32, stream: *mut c_void) -> c_int;

    // wire formats: SEC1 compressed (sec2), zcash (G1), RFC 8032 (ed25519)
    pub fn eccx_compressed_bytes(curve: c_int) -> c_int;
    pub fn eccx_point_decompress(ctx: *mut eccx_ctx, curve: c_int, n: usize, enc: *const u8, out: *mut u8,
                                 flags: *mut u8, opts: u32) -> c_int;
    pub fn eccx_point_compress(ctx: *mut eccx_ctx, curve: c_int, n: usize, xy: *const u8, inf: *const u8,
                               out: *mut u8, opts: u32) -> c_int;
    pub fn eccx_point_decompress_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_enc: *const c_void,
                                     d_out: *mut c_void, d_flags: *mut c_void, opts: u32, stream: *mut c_void) -> c_int;
    pub fn eccx_point_compress_dev(ctx: *mut eccx_ctx, curve: c_int, n: usize, d_xy: *const c_void,
                                   d_inf: *const c_void, d_out: *mut c_void, opts: u32, stream: *mut c_void) -> c_int;

    pub fn eccx_comb_table(ctx: *mut eccx_ctx, curve: c_int, out: *mut u8) -> c_int;

    pub fn eccx_scalarmul_var_sharded(ctxs: *mut *mut eccx_ctx, nctx: c_int, curve: c_int, n: usize,
                                      scalars: *const u8, points: *const u8, out: *mut u8, flags: *mut u8,
                                      opts: u32) -> c_int;
    pub fn eccx_scalarmul_base_sharded(ctxs: *mut *mut eccx_ctx, nctx: c_int, curve: c_int, n: usize,
                                       scalars: *const u8, out: *mut u8, flags: *mut u8, opts: u32) -> c_int;
}
