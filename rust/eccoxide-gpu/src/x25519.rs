//! X25519 over a batch: `protocol::x25519::x25519` (src/protocol/x25519.rs:36-45), i.e. the
//! curve25519 Montgomery ladder (src/curve/curve25519.rs:474-541) with RFC 7748 clamping.
use crate::{ffi, GpuContext, GpuError};

/// `out[i] = x25519(&scalars[i], &us[i])`; `us = None` multiplies the base point u = 9
/// (`x25519_base`, x25519.rs:49-51).  `zero[i]` is set where the result is the all-zero string, which
/// a Diffie-Hellman caller must reject (x25519.rs:33-35).  The ladder is uniform: conditional swaps
/// are selects and there is no table.
pub fn x25519_batch(ctx: &GpuContext, scalars: &[[u8; 32]], us: Option<&[[u8; 32]]>)
                    -> Result<(Vec<[u8; 32]>, Vec<bool>), GpuError> {
    let n = scalars.len();
    if let Some(us) = us {
        assert_eq!(us.len(), n);
    }
    let k: Vec<u8> = scalars.iter().flatten().copied().collect();
    let u: Option<Vec<u8>> = us.map(|us| us.iter().flatten().copied().collect());
    let (mut out, mut flags) = (vec![0u8; n * 32], vec![0u8; n]);
    ctx.check(unsafe {
        ffi::eccx_x25519(ctx.raw(), n, k.as_ptr(), u.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()),
                         out.as_mut_ptr(), flags.as_mut_ptr(), 0)
    })?;
    Ok((out.chunks_exact(32).map(|c| c.try_into().unwrap()).collect(), flags.iter().map(|&f| f == 1).collect()))
}
