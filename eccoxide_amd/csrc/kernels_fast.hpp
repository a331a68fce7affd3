// k_batch_to_affine<C>: Point::to_affine (projective.rs:655-682) for a whole batch.  Each lane
// normalises U units with ONE field inversion (Montgomery's trick), and that inversion is by
// division steps (inv_gcd.hpp).  The ladders and combs that feed it are in kernels_unsat.hpp
// (default) and kernels.hpp (reference-mirroring).
#pragma once
#include "inv_gcd.hpp"
#include "kernels.hpp"

namespace eccx {

constexpr int FAST_TABLE_ROWS = 17;  // rows of the per-lane window table (entries 0..16 of the signed 5-bit windows)

// Normalise a batch of un-normalised points (rows of W3 words: X, Y, Z Montgomery limbs).
//   MODE 0: homogeneous x = X/Z, y = Y/Z, big-endian bytes (projective.rs:655-682)
//   MODE 1: Jacobian x = X/Z^2, y = Y/Z^3, big-endian bytes
//   MODE 2: edwards25519: x = X/Z, y = Y/Z, little-endian bytes, flag 1 = neutral element
//           (curve25519.rs:663-666; Z is never 0 on a complete Edwards curve)
// flags[i] on entry: 2 marks a rejected input (kept, zero output); on exit 1 marks infinity.
// Thread t of a workgroup handles units tile + u*WG + t, u = 0..U-1, with one inversion.
//   MODE 3: curve25519 x-only: u = X/Z with 0 for Z = 0, 32 little-endian bytes per unit,
//           flag 1 = result is zero (curve25519.rs:529-532; x25519.rs:33-35)
enum { NORM_HOMOGENEOUS = 0, NORM_JACOBIAN = 1, NORM_EDWARDS = 2, NORM_MONTGOMERY_U = 3, NORM_JACOBIAN_X = 4 };
template <class C, int MODE, int U>
__global__ void __launch_bounds__(WG) k_batch_to_affine(size_t n, const uint32_t* __restrict__ pts,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ flags) {
  constexpr int L = C::L;
  constexpr int FB = C::FB;
  constexpr int W3 = row_words<L>();
  const size_t tile_units = (size_t)WG * U;
  for (size_t tile = (size_t)blockIdx.x * tile_units; tile < n; tile += (size_t)gridDim.x * tile_units) {
    Fe<L> pre[U];  // prefix products of the (substituted) Z values
    Fe<L> one;
    fe_set<C>(one, C::ONE);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      Fe<L> z = one;
      if (i < n) {
        const uint32_t* r = pts + i * (size_t)W3 + 2 * L;
#pragma unroll
        for (int j = 0; j < L; ++j) z.v[j] = r[j];
        if (fe_is_zero<C>(z)) z = one;  // z_inverse_ct substitutes 1 (projective.rs:655-659)
      }
      if (u == 0) pre[0] = z;
      else fe_mul<C>(pre[u], pre[u - 1], z);
    }
    Fe<L> inv;
    fe_inv_fast<C>(inv, pre[U - 1]);  // division steps: a sixth of the Fermat chain (inv_gcd.hpp)
#pragma unroll
    for (int u = U - 1; u >= 0; --u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      Fe<L> x = one, y = one, z = one;
      bool present = false;
      if (i < n) {
        const uint32_t* r = pts + i * (size_t)W3;
#pragma unroll
        for (int j = 0; j < L; ++j) { x.v[j] = r[j]; y.v[j] = r[L + j]; z.v[j] = r[2 * L + j]; }
        present = !fe_is_zero<C>(z);
        if (!present) z = one;
      }
      Fe<L> zi;
      if (u > 0) {
        fe_mul<C>(zi, inv, pre[u - 1]);
        fe_mul<C>(inv, inv, z);
      } else {
        zi = inv;
      }
      Fe<L> ax, ay;
      if constexpr (MODE == NORM_JACOBIAN) {
        Fe<L> zi2;
        fe_sqr<C>(zi2, zi);
        fe_mul<C>(ax, x, zi2);
        fe_mul<C>(zi2, zi2, zi);
        fe_mul<C>(ay, y, zi2);
      } else {
        fe_mul<C>(ax, x, zi);
        fe_mul<C>(ay, y, zi);
      }
      if (i < n) {
        const bool rejected = flags[i] == 2;
        Fe<L> t;
        if constexpr (MODE == NORM_MONTGOMERY_U) {
          // u = X * invert_or_zero(Z) (curve25519.rs:529-532): Z = 0 gives u = 0
          fe_from_mont<C>(t, ax);
          if (!present) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)FB, t);
          flags[i] = fe_is_zero<C>(t) ? 1 : 0;
        } else if constexpr (MODE == NORM_EDWARDS) {
          const bool neutral = fe_is_zero<C>(ax) && fe_eq<C>(ay, one);
          fe_from_mont<C>(t, ax);
          if (rejected) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)(2 * FB), t);
          fe_from_mont<C>(t, ay);
          if (rejected) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)(2 * FB) + FB, t);
          flags[i] = rejected ? 2 : (neutral ? 1 : 0);
        } else {
          const bool ok = present && !rejected;
          fe_from_mont<C>(t, ax);
          if (!ok) fe_zero<C>(t);
          fe_store_be<C>(out + i * (size_t)(2 * FB), t);
          fe_from_mont<C>(t, ay);
          if (!ok) fe_zero<C>(t);
          fe_store_be<C>(out + i * (size_t)(2 * FB) + FB, t);
          flags[i] = rejected ? 2 : (present ? 0 : 1);
        }
      }
    }
  }
}

}  // namespace eccx
