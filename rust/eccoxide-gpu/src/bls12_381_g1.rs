//! BLS12-381 G1 (`src/curve/bls12_381/g1.rs`, a = 0 Weierstrass over Fp): batch scalar
//! multiplication, the zcash encodings (`src/curve/bls12_381/serialize.rs:253-383`) and the
//! subgroup membership test (`g1.rs:90-109`).
use eccoxide::curve::bls12_381::g1::PointAffine;
use eccoxide::curve::bls12_381::{Fp, Scalar};

use crate::{ffi, GpuContext, GpuError, Secrecy, Unit};

const FB: usize = 48;
const SB: usize = 32;
const ID: core::ffi::c_int = ffi::ECCX_BLS12_381_G1;

fn push_point(buf: &mut Vec<u8>, p: &PointAffine) {
    let (x, y) = p.to_coordinate();
    buf.extend_from_slice(&x.to_bytes());
    buf.extend_from_slice(&y.to_bytes());
}

fn parse_points(out: &[u8], flags: &[u8]) -> Vec<Unit<PointAffine>> {
    flags
        .iter()
        .enumerate()
        .map(|(i, &f)| {
            Unit::from_flag(f, || {
                let rec = &out[i * 2 * FB..(i + 1) * 2 * FB];
                let x = Fp::from_bytes(rec[..FB].try_into().unwrap())?;
                let y = Fp::from_bytes(rec[FB..].try_into().unwrap())?;
                PointAffine::from_coordinate(&x, &y)
            })
        })
        .collect()
}

/// Where the base points of a variable-base call are known to lie.
#[derive(Debug, Clone, Copy, PartialEq, Eq, Default)]
pub enum Bases {
    /// Any point of the curve (what the reference's `*` accepts).
    #[default]
    AnyCurvePoint,
    /// Every point is in the prime-order subgroup G1 (decoded by `from_compressed`, a hash-to-curve output, or a
    /// multiple of the generator): `ECCX_ASSUME_SUBGROUP`.  Public scalars: the endomorphism ladder, about 1.4x
    /// faster.  Secret scalars (`sk * H(m)`): the same ladder in secret-scalar form (every table row read, selects
    /// only), about 1.6x faster than for arbitrary curve points.
    InSubgroup,
}

/// `out[i] = (&Point::from_affine(&points[i]) * &scalars[i]).to_affine()` (g1.rs:38-41 ->
/// curve_macros.rs:103-105, :321-327).
pub fn mul_batch(ctx: &GpuContext, points: &[PointAffine], scalars: &[Scalar], bases: Bases, secrecy: Secrecy)
                 -> Result<Vec<Unit<PointAffine>>, GpuError> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let (mut k, mut xy) = (Vec::with_capacity(n * SB), Vec::with_capacity(n * 2 * FB));
    for (p, s) in points.iter().zip(scalars) {
        k.extend_from_slice(&s.to_bytes());
        push_point(&mut xy, p);
    }
    let mut opts = secrecy.opts_var();
    if bases == Bases::InSubgroup {
        opts |= ffi::ECCX_ASSUME_SUBGROUP;
    }
    let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_scalarmul_var(ctx.raw(), ID, n, k.as_ptr(), xy.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                core::ptr::null_mut(), opts)
    })?;
    Ok(parse_points(&out, &flags))
}

/// `out[i] = Point::mul_base(&scalars[i]).to_affine()` (curve_macros.rs:111-119).
pub fn mul_base_batch(ctx: &GpuContext, scalars: &[Scalar], secrecy: Secrecy) -> Result<Vec<Unit<PointAffine>>, GpuError> {
    let n = scalars.len();
    let mut k = Vec::with_capacity(n * SB);
    for s in scalars {
        k.extend_from_slice(&s.to_bytes());
    }
    let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_scalarmul_base(ctx.raw(), ID, n, k.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                 core::ptr::null_mut(), secrecy.opts())
    })?;
    Ok(parse_points(&out, &flags))
}

/// `PointAffine::from_compressed` (`check_subgroup`, serialize.rs:299-313) or
/// `from_compressed_oncurve_only` (:321-335) over a batch of 48-byte encodings.  The encoding of the
/// point at infinity and every rejected encoding give `None`; `statuses` tells them apart
/// (`ECCX_FLAG_INFINITY` / `ECCX_FLAG_REJECTED`).
pub fn from_compressed_batch(ctx: &GpuContext, encodings: &[[u8; 48]], check_subgroup: bool)
                             -> Result<(Vec<Unit<PointAffine>>, Vec<u8>), GpuError> {
    let n = encodings.len();
    let enc: Vec<u8> = encodings.iter().flatten().copied().collect();
    let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_point_decompress(ctx.raw(), ID, n, enc.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                   if check_subgroup { ffi::ECCX_CHECK_SUBGROUP } else { 0 })
    })?;
    Ok((parse_points(&out, &flags), flags))
}

/// `points[i].to_compressed()` (serialize.rs:253-262).
pub fn to_compressed_batch(ctx: &GpuContext, points: &[PointAffine]) -> Result<Vec<[u8; 48]>, GpuError> {
    let n = points.len();
    let mut xy = Vec::with_capacity(n * 2 * FB);
    for p in points {
        push_point(&mut xy, p);
    }
    let mut out = vec![0u8; n * 48];
    ctx.check(unsafe { ffi::eccx_point_compress(ctx.raw(), ID, n, xy.as_ptr(), core::ptr::null(), out.as_mut_ptr(), 0) })?;
    Ok(out.chunks_exact(48).map(|c| c.try_into().unwrap()).collect())
}

/// `points[i].is_in_subgroup()` (g1.rs:105-109) over a batch: decode of the uncompressed form under
/// `ECCX_CHECK_SUBGROUP`, i.e. the endomorphism test on the device.
pub fn is_in_subgroup_batch(ctx: &GpuContext, points: &[PointAffine]) -> Result<Vec<bool>, GpuError> {
    let n = points.len();
    let mut enc = Vec::with_capacity(n * 2 * FB);
    for p in points {
        push_point(&mut enc, p); // x||y with clear flag bits = to_uncompressed of a finite point
    }
    let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_point_decompress(ctx.raw(), ID, n, enc.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                   ffi::ECCX_UNCOMPRESSED | ffi::ECCX_CHECK_SUBGROUP)
    })?;
    Ok(flags.iter().map(|&f| f == ffi::ECCX_FLAG_FINITE).collect())
}
