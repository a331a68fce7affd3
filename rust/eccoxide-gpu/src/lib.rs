//! Batched scalar multiplication for eccoxide's curve types on an AMD MI355X.
//!
//! eccoxide computes one `&Point * &Scalar` / `Point::mul_base` at a time on the CPU
//! (`src/curve/fiat/curve_macros.rs:321-343`, `:55-63`, `src/curve/curve25519.rs:746-762`, `:840-851`;
//! the `CurveGroup` trait, `src/curve/group.rs:28-70`).  This crate keeps those types -- `Scalar`,
//! `PointAffine`, `FieldElement`, `curve25519::Point` -- and adds the batch forms, computed by
//! `libeccx.so` (`include/eccx.h`): every function here marshals the crate's values into the byte
//! layout the reference's own `to_bytes` produces, makes ONE call into the library for the whole
//! batch, and parses the canonical bytes back through the crate's checked constructors.  Results are
//! bit-identical to the CPU path (affine coordinates; `None` for the point at infinity).
//!
//! ```ignore
//! let gpu = eccoxide_gpu::GpuContext::new(0)?;
//! let out = eccoxide_gpu::p256r1::mul_batch(&gpu, &points, &scalars, Secrecy::Secret)?;
//! assert_eq!(out[0].clone().point(), (&Point::from_affine(&points[0]) * &scalars[0]).to_affine());
//! ```
//!
//! Side channels: every scalar-multiplication function takes a [`Secrecy`] and there is no default --
//! the reference's `*` and `mul_base` are constant-time, so a drop-in caller has to say which it needs.
//! [`Secrecy::Secret`] (key generation, signing, ECDH) selects `ECCX_CT_SCAN`: no memory address and no
//! branch depends on a scalar digit; [`Secrecy::Public`] (signature verification, public-key checks)
//! the faster kernels that index their tables by scalar digits.  See `include/eccx.h`.
//!
//! STATUS: this crate has never been compiled -- the build image has no Rust toolchain.  `ffi.rs` is
//! checked against `include/eccx.h` mechanically (`tools/check_rust_ffi.py`), the wrappers only by eye
//! and by that script's lints (macro fragments, the eccoxide items they name).

pub mod ffi;

mod weierstrass;
pub use weierstrass::{p256r1, p384r1, p521r1};
pub mod bls12_381_g1;
pub mod ed25519;
pub mod x25519;

use core::ffi::{c_int, CStr};
use std::fmt;

/// Error of a library call: the `ECCX_ERR_*` code and the library's message.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for GpuError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "eccx error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for GpuError {}

/// Whether the scalars of a call are secret.  Deliberately without a `Default`.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Secrecy {
    /// Public scalars (signature verification, public-key validation): the fast kernels, whose table
    /// lookups are indexed by scalar digits.
    Public,
    /// Secret scalars: `ECCX_CT_SCAN` -- every table entry read at every lookup, selects instead of
    /// branches (what the reference's own `select_from_table` does).
    Secret,
    /// Secret scalars, fixed base only: `ECCX_CT_SCAN | ECCX_CT_GATHER` -- the lookup is a cross-lane
    /// register gather inside the wavefront (no memory address depends on a digit; its timing was measured
    /// uniform, which is not an architectural guarantee).  About 1.4x faster than `Secret`.
    SecretLaneGather,
}

impl Secrecy {
    pub(crate) fn opts(self) -> u32 {
        match self {
            Secrecy::Public => 0,
            Secrecy::Secret => ffi::ECCX_CT_SCAN,
            Secrecy::SecretLaneGather => ffi::ECCX_CT_SCAN | ffi::ECCX_CT_GATHER,
        }
    }
    /// For the entry points that have no gather form (variable base): plain `ECCX_CT_SCAN`.
    pub(crate) fn opts_var(self) -> u32 {
        match self {
            Secrecy::Public => 0,
            _ => ffi::ECCX_CT_SCAN,
        }
    }
}

/// One unit of a batch result: the library's three flag values kept apart (`ECCX_FLAG_*`).
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Unit<P> {
    /// A finite point.
    Point(P),
    /// The point at infinity / neutral element (the reference's `to_affine() == None`).
    Infinity,
    /// The input of this unit was rejected (not canonical, not on the curve, not in the subgroup), or the
    /// library's bytes did not parse back through the crate's checked constructors.
    Rejected,
}

impl<P> Unit<P> {
    /// `Some` for a finite point, `None` for infinity AND for a rejected unit (the reference's `Option`).
    pub fn point(self) -> Option<P> {
        match self {
            Unit::Point(p) => Some(p),
            _ => None,
        }
    }
    pub fn is_rejected(&self) -> bool {
        matches!(self, Unit::Rejected)
    }
    pub(crate) fn from_flag(flag: u8, parse: impl FnOnce() -> Option<P>) -> Self {
        match flag {
            ffi::ECCX_FLAG_FINITE => parse().map_or(Unit::Rejected, Unit::Point),
            ffi::ECCX_FLAG_INFINITY => Unit::Infinity,
            _ => Unit::Rejected,
        }
    }
}

/// One engine context bound to one GPU (`eccx_init` / `eccx_shutdown`).
///
/// Calls on one context must not overlap in time: the batch functions take `&GpuContext` and the
/// context is `Send` but not `Sync`; use one context per host thread (and per GPU).
pub struct GpuContext {
    raw: *mut ffi::eccx_ctx,
    device: i32,
}

// The context owns device memory and streams only; moving it to another thread is fine.
unsafe impl Send for GpuContext {}

impl GpuContext {
    /// Create a context on HIP device `device`.
    pub fn new(device: i32) -> Result<Self, GpuError> {
        let mut raw: *mut ffi::eccx_ctx = core::ptr::null_mut();
        let rc = unsafe { ffi::eccx_init(device as c_int, &mut raw) };
        if rc != ffi::ECCX_OK {
            return Err(GpuError { code: rc, message: strerror(rc) });
        }
        Ok(GpuContext { raw, device })
    }

    pub fn device(&self) -> i32 {
        self.device
    }

    pub(crate) fn raw(&self) -> *mut ffi::eccx_ctx {
        self.raw
    }

    /// Build the fixed-base tables of `curve` now instead of at the first `mul_base` (`eccx_prepare`).
    pub fn prepare_base(&self, curve: Curve) -> Result<(), GpuError> {
        self.check(unsafe { ffi::eccx_prepare(self.raw, curve.id(), ffi::ECCX_PREP_BASE) })
    }

    /// Size the scratch buffers for batches of up to `max_n` units of `curve` (`eccx_reserve`).
    pub fn reserve(&self, curve: Curve, max_n: usize) -> Result<(), GpuError> {
        // the batch functions of this crate call the host-buffer entry points: their device-side copies too
        self.check(unsafe {
            ffi::eccx_reserve(self.raw, curve.id(), max_n, ffi::ECCX_PREP_VAR | ffi::ECCX_PREP_CT | ffi::ECCX_PREP_HOST)
        })
    }

    /// Device memory the context owns.
    pub fn device_bytes(&self) -> usize {
        unsafe { ffi::eccx_device_bytes(self.raw) }
    }

    pub(crate) fn check(&self, rc: c_int) -> Result<(), GpuError> {
        if rc == ffi::ECCX_OK {
            return Ok(());
        }
        let detail = unsafe { CStr::from_ptr(ffi::eccx_last_error(self.raw)) }.to_string_lossy().into_owned();
        let message = if detail.is_empty() { strerror(rc) } else { format!("{}: {}", strerror(rc), detail) };
        Err(GpuError { code: rc, message })
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { ffi::eccx_shutdown(self.raw) };
    }
}

fn strerror(rc: c_int) -> String {
    unsafe { CStr::from_ptr(ffi::eccx_strerror(rc)) }.to_string_lossy().into_owned()
}

/// The curves the engine implements (`eccx_curve`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Curve {
    P256r1,
    P384r1,
    P521r1,
    Bls12381G1,
    Ed25519,
}

impl Curve {
    pub fn id(self) -> c_int {
        match self {
            Curve::P256r1 => ffi::ECCX_P256R1,
            Curve::P384r1 => ffi::ECCX_P384R1,
            Curve::P521r1 => ffi::ECCX_P521R1,
            Curve::Bls12381G1 => ffi::ECCX_BLS12_381_G1,
            Curve::Ed25519 => ffi::ECCX_ED25519,
        }
    }
    /// Field bytes FB and scalar bytes SB, as the library reports them.
    pub fn sizes(self) -> (usize, usize) {
        unsafe { (ffi::eccx_field_bytes(self.id()) as usize, ffi::eccx_scalar_bytes(self.id()) as usize) }
    }
}

/// Run a batch over several contexts (one per GPU): contiguous shards, results in order
/// (`eccx_scalarmul_var_sharded`).  `scalars` is n x SB bytes, `points` n x 2FB; returns
/// (n x 2FB affine bytes, n flag bytes).
pub fn scalarmul_var_sharded_bytes(ctxs: &[&GpuContext], curve: Curve, scalars: &[u8], points: &[u8])
                                   -> Result<(Vec<u8>, Vec<u8>), GpuError> {
    let (fb, sb) = curve.sizes();
    assert!(!ctxs.is_empty() && scalars.len() % sb == 0);
    let n = scalars.len() / sb;
    assert_eq!(points.len(), n * 2 * fb);
    let mut raw: Vec<*mut ffi::eccx_ctx> = ctxs.iter().map(|c| c.raw()).collect();
    let (mut out, mut flags) = (vec![0u8; n * 2 * fb], vec![0u8; n]);
    let rc = unsafe {
        ffi::eccx_scalarmul_var_sharded(raw.as_mut_ptr(), raw.len() as c_int, curve.id(), n, scalars.as_ptr(),
                                        points.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(), 0)
    };
    ctxs[0].check(rc)?;
    Ok((out, flags))
}

/// Fixed-base counterpart of [`scalarmul_var_sharded_bytes`] (`eccx_scalarmul_base_sharded`).
pub fn scalarmul_base_sharded_bytes(ctxs: &[&GpuContext], curve: Curve, scalars: &[u8])
                                    -> Result<(Vec<u8>, Vec<u8>), GpuError> {
    let (fb, sb) = curve.sizes();
    assert!(!ctxs.is_empty() && scalars.len() % sb == 0);
    let n = scalars.len() / sb;
    let mut raw: Vec<*mut ffi::eccx_ctx> = ctxs.iter().map(|c| c.raw()).collect();
    let (mut out, mut flags) = (vec![0u8; n * 2 * fb], vec![0u8; n]);
    let rc = unsafe {
        ffi::eccx_scalarmul_base_sharded(raw.as_mut_ptr(), raw.len() as c_int, curve.id(), n, scalars.as_ptr(),
                                         out.as_mut_ptr(), flags.as_mut_ptr(), 0)
    };
    ctxs[0].check(rc)?;
    Ok((out, flags))
}
