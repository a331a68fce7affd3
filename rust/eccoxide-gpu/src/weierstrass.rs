//! Batch forms of the sec2 curves' `Point` / `PointAffine` operations
//! (`fiat_define_weierstrass_points!`, `src/curve/fiat/curve_macros.rs:157-411`).

/// Generates one module per curve.  `$seg::...` is the curve's module in eccoxide (matched as identifiers:
/// a `path` fragment cannot be extended with `::{...}` in a `use`), `$id` its `eccx_curve`, `$fb` / `$sb`
/// its field and scalar sizes in bytes.
macro_rules! gpu_weierstrass_curve {
    ($modname:ident, $($seg:ident)::+, $id:expr, $fb:expr, $sb:expr) => {
        pub mod $modname {
            use $($seg)::+::{FieldElement, PointAffine, Scalar};
            use eccoxide::curve::Sign;

            use crate::{ffi, GpuContext, GpuError, Secrecy, Unit};

            const FB: usize = $fb;
            const SB: usize = $sb;

            fn push_point(buf: &mut Vec<u8>, p: &PointAffine) {
                let (x, y) = p.to_coordinate();
                buf.extend_from_slice(&x.to_bytes()); // big-endian (field_macros.rs:6-29)
                buf.extend_from_slice(&y.to_bytes());
            }

            fn parse_points(out: &[u8], flags: &[u8]) -> Vec<Unit<PointAffine>> {
                flags
                    .iter()
                    .enumerate()
                    .map(|(i, &f)| {
                        Unit::from_flag(f, || {
                            let rec = &out[i * 2 * FB..(i + 1) * 2 * FB];
                            let x = FieldElement::from_bytes(rec[..FB].try_into().unwrap())?;
                            let y = FieldElement::from_bytes(rec[FB..].try_into().unwrap())?;
                            PointAffine::from_coordinate(&x, &y)
                        })
                    })
                    .collect()
            }

            /// `out[i] = (&Point::from_affine(&points[i]) * &scalars[i]).to_affine()`
            /// (`impl Mul<&Scalar> for &Point`, curve_macros.rs:321-327).
            pub fn mul_batch(ctx: &GpuContext, points: &[PointAffine], scalars: &[Scalar], secrecy: Secrecy)
                             -> Result<Vec<Unit<PointAffine>>, GpuError> {
                assert_eq!(points.len(), scalars.len());
                let n = points.len();
                let mut k = Vec::with_capacity(n * SB);
                let mut xy = Vec::with_capacity(n * 2 * FB);
                for (p, s) in points.iter().zip(scalars) {
                    k.extend_from_slice(&s.to_bytes()); // what Point::scale hands the ladder (curve_macros.rs:47-49)
                    push_point(&mut xy, p);
                }
                let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
                ctx.check(unsafe {
                    ffi::eccx_scalarmul_var(ctx.raw(), $id, n, k.as_ptr(), xy.as_ptr(), out.as_mut_ptr(),
                                            flags.as_mut_ptr(), core::ptr::null_mut(), secrecy.opts_var())
                })?;
                Ok(parse_points(&out, &flags))
            }

            /// `out[i] = Point::mul_base(&scalars[i]).to_affine()` (curve_macros.rs:55-63).
            pub fn mul_base_batch(ctx: &GpuContext, scalars: &[Scalar], secrecy: Secrecy)
                                  -> Result<Vec<Unit<PointAffine>>, GpuError> {
                let n = scalars.len();
                let mut k = Vec::with_capacity(n * SB);
                for s in scalars {
                    k.extend_from_slice(&s.to_bytes());
                }
                let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
                ctx.check(unsafe {
                    ffi::eccx_scalarmul_base(ctx.raw(), $id, n, k.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(),
                                             core::ptr::null_mut(), secrecy.opts())
                })?;
                Ok(parse_points(&out, &flags))
            }

            /// ECDSA verification's point `u1*G + u2*Q` for a batch (src/protocol/ecdsa.rs:215), one
            /// fused kernel; public data only.
            pub fn verify_points(ctx: &GpuContext, u1: &[Scalar], u2: &[Scalar], q: &[PointAffine])
                                 -> Result<Vec<Unit<PointAffine>>, GpuError> {
                assert!(u1.len() == u2.len() && u1.len() == q.len());
                let n = q.len();
                let (mut k1, mut k2) = (Vec::with_capacity(n * SB), Vec::with_capacity(n * SB));
                let mut xy = Vec::with_capacity(n * 2 * FB);
                for i in 0..n {
                    k1.extend_from_slice(&u1[i].to_bytes());
                    k2.extend_from_slice(&u2[i].to_bytes());
                    push_point(&mut xy, &q[i]);
                }
                let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
                ctx.check(unsafe {
                    ffi::eccx_double_scalarmul(ctx.raw(), $id, n, k1.as_ptr(), k2.as_ptr(), xy.as_ptr(),
                                               out.as_mut_ptr(), flags.as_mut_ptr(), 0)
                })?;
                Ok(parse_points(&out, &flags))
            }

            /// `out[i] = a[i] + b[i]`, or `a[i] - b[i]` with `subtract` (impl Add / Sub for Point,
            /// curve_macros.rs:349-411); `None` operands are the point at infinity.
            pub fn add_batch(ctx: &GpuContext, a: &[Option<PointAffine>], b: &[Option<PointAffine>], subtract: bool)
                             -> Result<Vec<Unit<PointAffine>>, GpuError> {
                assert_eq!(a.len(), b.len());
                let n = a.len();
                let marshal = |ps: &[Option<PointAffine>]| {
                    let (mut xy, mut inf) = (Vec::with_capacity(n * 2 * FB), Vec::with_capacity(n));
                    for p in ps {
                        match p {
                            Some(p) => {
                                push_point(&mut xy, p);
                                inf.push(0u8);
                            }
                            None => {
                                xy.extend(core::iter::repeat(0u8).take(2 * FB));
                                inf.push(1u8);
                            }
                        }
                    }
                    (xy, inf)
                };
                let ((ax, ai), (bx, bi)) = (marshal(a), marshal(b));
                let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
                ctx.check(unsafe {
                    ffi::eccx_point_add(ctx.raw(), $id, n, ax.as_ptr(), ai.as_ptr(), bx.as_ptr(), bi.as_ptr(),
                                        out.as_mut_ptr(), flags.as_mut_ptr(), if subtract { ffi::ECCX_SUBTRACT } else { 0 })
                })?;
                Ok(parse_points(&out, &flags))
            }

            /// `PointAffine::decompress(&xs[i], signs[i])` (curve_macros.rs:221-223): the SEC1 prefix is
            /// the `Sign` (`Negative` = y odd, field_macros.rs:557-565).
            pub fn decompress_batch(ctx: &GpuContext, xs: &[FieldElement], signs: &[Sign])
                                    -> Result<Vec<Unit<PointAffine>>, GpuError> {
                assert_eq!(xs.len(), signs.len());
                let n = xs.len();
                let mut enc = Vec::with_capacity(n * (FB + 1));
                for (x, s) in xs.iter().zip(signs) {
                    enc.push(match s {
                        Sign::Positive => 0x02,
                        Sign::Negative => 0x03,
                    });
                    enc.extend_from_slice(&x.to_bytes());
                }
                let (mut out, mut flags) = (vec![0u8; n * 2 * FB], vec![0u8; n]);
                ctx.check(unsafe {
                    ffi::eccx_point_decompress(ctx.raw(), $id, n, enc.as_ptr(), out.as_mut_ptr(), flags.as_mut_ptr(), 0)
                })?;
                Ok(parse_points(&out, &flags))
            }

            /// `points[i].compress()` as SEC1 bytes: `0x02 | (y odd)` then x (curve_macros.rs:211-213).
            pub fn compress_batch(ctx: &GpuContext, points: &[PointAffine]) -> Result<Vec<[u8; FB + 1]>, GpuError> {
                let n = points.len();
                let mut xy = Vec::with_capacity(n * 2 * FB);
                for p in points {
                    push_point(&mut xy, p);
                }
                let mut out = vec![0u8; n * (FB + 1)];
                ctx.check(unsafe {
                    ffi::eccx_point_compress(ctx.raw(), $id, n, xy.as_ptr(), core::ptr::null(), out.as_mut_ptr(), 0)
                })?;
                Ok(out.chunks_exact(FB + 1).map(|c| c.try_into().unwrap()).collect())
            }
        }
    };
}

gpu_weierstrass_curve!(p256r1, eccoxide::curve::sec2::p256r1, crate::ffi::ECCX_P256R1, 32, 32);
gpu_weierstrass_curve!(p384r1, eccoxide::curve::sec2::p384r1, crate::ffi::ECCX_P384R1, 48, 48);
gpu_weierstrass_curve!(p521r1, eccoxide::curve::sec2::p521r1, crate::ffi::ECCX_P521R1, 66, 66);
