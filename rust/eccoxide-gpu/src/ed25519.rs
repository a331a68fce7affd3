//! edwards25519 (`src/curve/curve25519.rs`): `Point::scale`, `Point::mul_base`, the Ed25519 verification
//! shape and the RFC 8032 point encoding over batches.
use eccoxide::curve::curve25519::{FieldElement, Point, Scalar};

use crate::{ffi, GpuContext, GpuError, Secrecy, Unit};

const ID: core::ffi::c_int = ffi::ECCX_ED25519;

fn push_point(buf: &mut Vec<u8>, p: &Point) {
    let (x, y) = p.to_affine(); // curve25519.rs:663-666
    buf.extend_from_slice(&x.to_bytes()); // little-endian (curve25519.rs:138)
    buf.extend_from_slice(&y.to_bytes());
}

/// Flag 1 marks the neutral element, which HAS affine coordinates (0, 1) on this curve: it comes back as
/// `Unit::Point(identity)`, never as `Unit::Infinity`; only a rejected input gives `Unit::Rejected`.
fn parse_points(out: &[u8], flags: &[u8]) -> Vec<Unit<Point>> {
    flags
        .iter()
        .enumerate()
        .map(|(i, &f)| {
            let as_finite = if f == ffi::ECCX_FLAG_INFINITY { ffi::ECCX_FLAG_FINITE } else { f };
            Unit::from_flag(as_finite, || {
                let rec = &out[i * 64..(i + 1) * 64];
                let x = FieldElement::from_bytes(rec[..32].try_into().unwrap())?;
                let y = FieldElement::from_bytes(rec[32..].try_into().unwrap())?;
                Point::from_coordinate(&x, &y)
            })
        })
        .collect()
}

/// `out[i] = points[i].scale(&scalars[i])` (curve25519.rs:746-762).  The library takes the scalar as
/// the big-endian string the reference's loop indexes (`Scalar::to_bytes_be`).
pub fn scale_batch(ctx: &GpuContext, points: &[Point], scalars: &[Scalar], secrecy: Secrecy)
                   -> Result<Vec<Unit<Point>>, GpuError> {
    assert_eq!(points.len(), scalars.len());
    let n = points.len();
    let (mut k, mut xy) = (Vec::with_capacity(n * 32), Vec::with_capacity(n * 64));
    for (p, s) in points.iter().zip(scalars) {
        k.extend_from_slice(&s.to_bytes_be());
        push_point(&mut xy, p);
    }
    let (mut out, mut flags) = (vec![0u8; n * 64], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_scalarmul_var(ctx.raw(), ID, n, k.as_ptr(), xy.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                core::ptr::null_mut(), secrecy.opts_var())
    })?;
    Ok(parse_points(&out, &flags))
}

/// `out[i] = Point::mul_base(&scalars[i])` (curve25519.rs:840-851): Ed25519 key generation / signing.
pub fn mul_base_batch(ctx: &GpuContext, scalars: &[Scalar], secrecy: Secrecy) -> Result<Vec<Unit<Point>>, GpuError> {
    let n = scalars.len();
    let mut k = Vec::with_capacity(n * 32);
    for s in scalars {
        k.extend_from_slice(&s.to_bytes_be());
    }
    let (mut out, mut flags) = (vec![0u8; n * 64], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_scalarmul_base(ctx.raw(), ID, n, k.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                 core::ptr::null_mut(), secrecy.opts())
    })?;
    Ok(parse_points(&out, &flags))
}

/// `[s]B - [k]A` for a batch: what `Point::double_scalar_mul_base_vartime` computes in Ed25519
/// verification (curve25519.rs:1157-1183, src/protocol/ed25519.rs:145).
pub fn verify_points(ctx: &GpuContext, s: &[Scalar], k: &[Scalar], a: &[Point]) -> Result<Vec<Unit<Point>>, GpuError> {
    assert!(s.len() == k.len() && s.len() == a.len());
    let n = a.len();
    let (mut u1, mut u2, mut xy) = (Vec::with_capacity(n * 32), Vec::with_capacity(n * 32), Vec::with_capacity(n * 64));
    for i in 0..n {
        u1.extend_from_slice(&s[i].to_bytes_be());
        u2.extend_from_slice(&k[i].to_bytes_be());
        push_point(&mut xy, &a[i]);
    }
    let (mut out, mut flags) = (vec![0u8; n * 64], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_double_scalarmul(ctx.raw(), ID, n, u1.as_ptr(), u2.as_ptr(), xy.as_ptr(), out.as_mut_ptr(),
                                   flags.as_mut_ptr(), ffi::ECCX_SUBTRACT)
    })?;
    Ok(parse_points(&out, &flags))
}

/// RFC 8032 `decode_point` (src/protocol/ed25519.rs:38-59) over a batch of 32-byte encodings.
pub fn decode_points(ctx: &GpuContext, encodings: &[[u8; 32]]) -> Result<Vec<Unit<Point>>, GpuError> {
    let n = encodings.len();
    let enc: Vec<u8> = encodings.iter().flatten().copied().collect();
    let (mut out, mut flags) = (vec![0u8; n * 64], vec![0u8; n]);
    ctx.check(unsafe { ffi::eccx_point_decompress(ctx.raw(), ID, n, enc.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(), 0) })?;
    Ok(parse_points(&out, &flags))
}

/// RFC 8032 `encode_point` (src/protocol/ed25519.rs:27-36).
pub fn encode_points(ctx: &GpuContext, points: &[Point]) -> Result<Vec<[u8; 32]>, GpuError> {
    let n = points.len();
    let mut xy = Vec::with_capacity(n * 64);
    for p in points {
        push_point(&mut xy, p);
    }
    let mut out = vec![0u8; n * 32];
    ctx.check(unsafe { ffi::eccx_point_compress(ctx.raw(), ID, n, xy.as_ptr(), core::ptr::null(), out.as_mut_ptr(), 0) })?;
    Ok(out.chunks_exact(32).map(|c| c.try_into().unwrap()).collect())
}
