// BLS12-381 G1 kernels built on the curve's endomorphism sigma(x, y) = (beta x, y), which acts on
// the prime-order subgroup G1 as multiplication by lambda = -x^2 (x the curve's seed):
//
//   k_bls_subgroup_check   PointAffine::is_in_subgroup          src/curve/bls12_381/g1.rs:90-109
//       sigma(P) == [-x^2]P with [x^2]P = [|x|]([|x|]P), |x| a 64-bit constant of weight 6
//       (mul_by_abs_x, g1.rs:62-78): 126 doublings + 10 additions instead of the 255 + 66 of [r]P.
//       Run after decompression under ECCX_CHECK_SUBGROUP (from_compressed / from_uncompressed,
//       src/curve/bls12_381/serialize.rs:299-313, :353-366).
//   k_scalarmul_glv_unsat  &Point * &Scalar for bases the caller guarantees to be in G1
//       (ECCX_ASSUME_SUBGROUP): k = k1 + k2 x^2 with k2 = floor(k / x^2), so
//       k P = k1 P + k2 [x^2]P = k1 P + k2 (beta X, -Y): two half-length scalars share one chain of
//       doublings (signed 5-bit windows over a 6-column table row X, Y, Z, Z^2, Z^3, beta X).
//       For a point outside G1 sigma is not [lambda] and the result is NOT k P: opt-in only; the
//       default ladder (k_scalarmul_var_unsat) takes any curve point, like the reference's.
//
// Jacobian coordinates and special-case patches as in kernels_unsat.hpp.
#pragma once
#include "kernels_coz.hpp"

namespace eccx {

// ---- complete Jacobian additions (generic formulas + the patches of the ladder) ----------------
// q += (x2, y2, 1); x2, y2 tight.  All special cases: q at infinity, q == -(x2, y2), q == (x2, y2).
template <class CU>
ECCX_DEV void ujac_madd_full(UJac<CU>& q, const U<CU, 1, 2>& x2, const U<CU, 1, 2>& y2) {
  U<CU, 1, 2> one;
#pragma unroll
  for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
  const bool q_inf = u_limbs_all_zero(q.z);
  UJac<CU> sum;
  bool hz, rz;
  ujac_madd_raw<CU>(sum, hz, rz, q, x2, y2);
  const bool same_x = hz && !q_inf;
  const bool need_dbl = same_x && rz;
  if (same_x && !rz) u_set_zero(sum.z);  // q == -entry
  u_select(sum.x, q_inf, u_as<1, 3>(x2), sum.x);
  u_select(sum.y, q_inf, u_as<1, 3>(y2), sum.y);
  u_select(sum.z, q_inf, u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one), sum.z);
  if (__builtin_amdgcn_ballot_w64(need_dbl) != 0) {  // q == entry: rare, wave-uniform branch
    UJac<CU> t;
    ujac_dbl<CU>(t, q);
    u_select(sum.x, need_dbl, t.x, sum.x);
    u_select(sum.y, need_dbl, t.y, sum.y);
    u_select(sum.z, need_dbl, t.z, sum.z);
  }
  q = sum;
}

// q += e for a Jacobian entry with cached Z^2, Z^3 (e finite or at infinity)
template <class CU>
ECCX_DEV void ujac_add_full(UJac<CU>& q, const UEntry<CU>& e) {
  const bool q_inf = u_limbs_all_zero(q.z);
  const bool e_inf = u_limbs_all_zero(e.z);
  UJac<CU> sum;
  bool hz, rz;
  U<CU, 1, 3> ey;
  ujac_add_raw<CU>(sum, hz, rz, ey, q, e, false);
  const bool same_x = hz && !q_inf && !e_inf;
  const bool need_dbl = same_x && rz;
  if (same_x && !rz) u_set_zero(sum.z);
  u_select(sum.x, q_inf, e.x, sum.x);
  u_select(sum.y, q_inf, e.y, sum.y);
  u_select(sum.z, q_inf, u_as<UJac<CU>::ZK, UJac<CU>::ZV>(e.z), sum.z);
  if (__builtin_amdgcn_ballot_w64(need_dbl) != 0) {
    UJac<CU> t;
    ujac_dbl<CU>(t, q);
    u_select(sum.x, need_dbl, t.x, sum.x);
    u_select(sum.y, need_dbl, t.y, sum.y);
    u_select(sum.z, need_dbl, t.z, sum.z);
  }
  u_select(q.x, e_inf, q.x, sum.x);
  u_select(q.y, e_inf, q.y, sum.y);
  u_select(q.z, e_inf, q.z, sum.z);
}

// ---- subgroup membership -----------------------------------------------------------------------
// xy: n x 2FB affine points as the decompression kernels wrote them; flags: 0 point, 1 infinity
// encoding, 2 rejected.  A record flagged 0 whose point is not in G1 becomes flag 2 with zero bytes.
template <class CU, class G>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_bls_subgroup_check(size_t n, uint8_t* __restrict__ xy,
                                                                                uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    Fe<L> rx, ry;
    fe_load_be<CS>(rx, xy + idx * (size_t)(2 * FB));
    fe_load_be<CS>(ry, xy + idx * (size_t)(2 * FB) + FB);
    const auto px = u_to_mont<CU>(rx);
    const auto py = u_to_mont<CU>(ry);
    U<CU, 1, 2> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    // pass 1: a = [|x|]P, P affine (mixed additions); the top bit of |x| is the starting value
    UJac<CU> a;
    a.x = u_as<1, 3>(px);
    a.y = u_as<1, 3>(py);
    a.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    for (int i = 62; i >= 0; --i) {
      UJac<CU> t;
      ujac_dbl<CU>(t, a);
      a = t;
      if ((G::SEED_ABS >> i) & 1) ujac_madd_full<CU>(a, px, py);  // wave-uniform: the seed is a constant
    }
    // pass 2: q = [|x|]a, a Jacobian.  Only (X, Y, Z) of a stay live across the doublings: Z^2 and Z^3
    // are recomputed at each of the five additions rather than held in registers
    a.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(u_reduce(a.z));  // tight; exact zero stays exact zero
    UJac<CU> q = a;
    for (int i = 62; i >= 0; --i) {
      UJac<CU> t;
      ujac_dbl<CU>(t, q);
      q = t;
      if ((G::SEED_ABS >> i) & 1) {
        UEntry<CU> e;
        e.x = a.x;
        e.y = a.y;
#pragma unroll
        for (int j = 0; j < CU::N; ++j) e.z.v[j] = a.z.v[j];
        e.zz = u_fit<1, 3>(u_sqr(e.z));
        e.zzz = u_fit<1, 3>(u_mul(e.zz, e.z));
        ujac_add_full<CU>(q, e);
      }
    }
    // sigma(P) == -q  <=>  q finite, beta x Z^2 == X and y Z^3 == -Y  (is_equivalent, projective.rs:133-139);
    // P is read again here instead of being kept in registers through the second pass
    const auto qz = u_reduce(q.z);
    const bool q_inf = u_is_zero_mod_p(qz);
    const auto zz = u_sqr(qz);
    const auto zzz = u_mul(zz, qz);
    const uint8_t* again = xy + idx * (size_t)(2 * FB);
    asm volatile("" : "+v"(again));
    fe_load_be<CS>(rx, again);
    fe_load_be<CS>(ry, again + FB);
    const auto lx = u_mul(u_mul_k<CU>(u_to_mont<CU>(rx), CU::BETA), zz);
    const auto ly = u_mul(u_to_mont<CU>(ry), zzz);
    const bool x_ok = u_is_zero_mod_p(u_reduce(u_sub(lx, q.x)));
    const bool y_ok = u_is_zero_mod_p(u_reduce(u_add(ly, q.y)));
    const bool inside = !q_inf && x_ok && y_ok;
    if (active && flags[idx] == 0 && !inside) {
      flags[idx] = 2;
      Fe<L> z;
      fe_zero<CS>(z);
      fe_store_be<CS>(xy + idx * (size_t)(2 * FB), z);
      fe_store_be<CS>(xy + idx * (size_t)(2 * FB) + FB, z);
    }
  }
}

}  // namespace eccx
