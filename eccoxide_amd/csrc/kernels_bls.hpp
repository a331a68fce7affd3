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
#include "kernels_unsat.hpp"

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

// ---- scalar split k = k1 + k2 x^2 ----------------------------------------------------------------
// k: 32 big-endian bytes (any value below 2^256).  k2 = floor(k / x^2) by Barrett with
// MU = floor(2^256 / x^2) (the estimate is at most 2 short), k1 = k - k2 x^2 < x^2.  Both halves
// as five little-endian 32-bit words (k1 < 2^128, k2 < 2^129).
template <class G>
ECCX_DEV void glv_split(uint32_t (&k1)[5], uint32_t (&k2)[5], const uint8_t* __restrict__ kb) {
  uint32_t k[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint8_t* b = kb + 28 - 4 * i;
    k[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
  }
  uint32_t prod[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint64_t t = (uint64_t)k[i] * G::MU[j] + prod[i + j] + carry;
      prod[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    prod[i + 5] = (uint32_t)carry;
  }
  uint32_t q[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = prod[8 + i];
  // r = k - q x^2 modulo 2^160 (the true remainder is below 3 x^2 < 2^130)
  uint32_t qx[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) qx[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i + j < 5) {
        const uint64_t t = (uint64_t)q[i] * G::X2[j] + qx[i + j] + carry;
        qx[i + j] = (uint32_t)t;
        carry = t >> 32;
      }
    }
    if (i + 4 < 5) qx[i + 4] = (uint32_t)carry;  // the row's last carry (only row 0 still lands below 2^160)
  }
  uint32_t r[5];
  {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint64_t t = (uint64_t)k[i] - qx[i] - borrow;
      r[i] = (uint32_t)t;
      borrow = (t >> 32) & 1u;
    }
  }
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    uint32_t d[5];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint64_t t = (uint64_t)r[i] - (i < 4 ? G::X2[i] : 0u) - borrow;
      d[i] = (uint32_t)t;
      borrow = (t >> 32) & 1u;
    }
    const bool ge = borrow == 0;  // r >= x^2
    uint32_t inc = ge ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      r[i] = ge ? d[i] : r[i];
      const uint64_t t = (uint64_t)q[i] + inc;
      q[i] = (uint32_t)t;
      inc = (uint32_t)(t >> 32);
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) { k1[i] = r[i]; k2[i] = q[i]; }
}

template <class CU>
constexpr int urowg_words() { return ((6 * CU::N + 3) / 4) * 4; }

template <class CU>
ECCX_DEV void uentryg_store(uint32_t* __restrict__ row, const UEntry<CU>& p, const U<CU, 1, 3>& bx) {
  constexpr int N = CU::N;
  constexpr int W = urowg_words<CU>();
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    w[i] = p.x.v[i]; w[N + i] = p.y.v[i]; w[2 * N + i] = p.z.v[i]; w[3 * N + i] = p.zz.v[i]; w[4 * N + i] = p.zzz.v[i];
    w[5 * N + i] = bx.v[i];
  }
#pragma unroll
  for (int i = 6 * N; i < W; ++i) w[i] = 0;
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// the entry as the ladder adds it: X (or beta X for the second half), Y, Z, Z^2, Z^3
template <class CU>
ECCX_DEV void uentryg_load(UEntry<CU>& p, const uint32_t* __restrict__ row, bool second) {
  constexpr int N = CU::N;
  constexpr int W = urowg_words<CU>();
  uint32_t w[W];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    p.x.v[i] = second ? w[5 * N + i] : w[i];
    p.y.v[i] = w[N + i]; p.z.v[i] = w[2 * N + i]; p.zz.v[i] = w[3 * N + i]; p.zzz.v[i] = w[4 * N + i];
  }
}

// ---- variable base through the endomorphism (bases in G1) ----------------------------------------
// scratch: [workgroup][row 0..16][thread][urowg_words]; row 0 holds the lane's split scalar
// (k1 in words 0..4, k2 in words 8..12), rows 1..16 the table d P.
template <class CU, class G>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_scalarmul_glv_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                                 const uint8_t* __restrict__ points,
                                                                                 uint32_t* __restrict__ rows_out,
                                                                                 uint8_t* __restrict__ flags,
                                                                                 uint32_t* __restrict__ scratch, uint32_t opts) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int SB = CS::SB;
  static_assert(SB == 32, "the split is written for 256-bit scalars");
  constexpr int NWIN = (G::K_BITS + 1 + 4) / 5;  // signed 5-bit windows over the halves
  constexpr int WR = urowg_words<CU>();
  static_assert(WR >= 16, "row 0 holds the split scalar");
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)WR;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * WR; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;

    {
      uint32_t k1[5], k2[5];
      glv_split<G>(k1, k2, scalars + idx * (size_t)SB);
      uint4* d = reinterpret_cast<uint4*>(row(0));
      d[0] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
      d[1] = make_uint4(k1[4], 0u, 0u, 0u);
      d[2] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
      d[3] = make_uint4(k2[4], 0u, 0u, 0u);
    }
    const uint32_t* __restrict__ kw = row(0);

    UJac<CU> q;
    bool rejected = false;
    {
      Fe<L> rx, ry;
      fe_load_be<CS>(rx, points + idx * (size_t)(2 * FB));
      fe_load_be<CS>(ry, points + idx * (size_t)(2 * FB) + FB);
      if (opts & OPT_VALIDATE) {
        Fe<L> mx, my;
        fe_to_mont<CS>(mx, rx);
        fe_to_mont<CS>(my, ry);
        rejected = !(fe_is_canonical<CS>(rx) && fe_is_canonical<CS>(ry) && on_curve<CS>(mx, my));
      }
      q.x = u_as<1, 3>(u_to_mont<CU>(rx));
      q.y = u_as<1, 3>(u_to_mont<CU>(ry));
    }
    U<CU, 1, 2> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    {
      UEntry<CU> e1;
      e1.x = q.x; e1.y = q.y; e1.z = u_as<1, 3>(one); e1.zz = e1.z; e1.zzz = e1.z;
      uentryg_store<CU>(row(1), e1, u_fit<1, 3>(u_mul_k<CU>(e1.x, CU::BETA)));
    }
    // Booth digit of window w of half `second`: |digit| in 0..16 and its sign.  Bits 5w-1 .. 5w+4
    // of the little-endian words (bit -1 is 0).
    auto booth = [&](int w, bool second, uint32_t& d, bool& neg) {
      const uint32_t* __restrict__ h = kw + (second ? 8 : 0);
      const int pos = 5 * w - 1 + 32;              // one zero word below the scalar
      const int wi = pos >> 5, sh = pos & 31;
      const uint32_t lo = (wi >= 1 && wi <= 5) ? h[wi - 1] : 0u;
      const uint32_t hi = (wi <= 4) ? h[wi] : 0u;
      const uint32_t w6 = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & 0x3fu;
      const uint32_t s = ~((w6 >> 5) - 1u);
      const uint32_t m = (((1u << 6) - w6 - 1u) & s) | (w6 & ~s);
      d = (m >> 1) + (m & 1u);
      neg = (s & 1u) != 0;
    };
    int b = 0;                    // table-build step: 0 -> T[2] = 2P, 1..14 -> T[b+2] = T[b+1] + P
    int win = NWIN - 1, sub = 5;  // the top window needs no doublings; sub 5: add from k1, sub 6: from k2
    bool fix_pending = false, fix_lane = false;
    for (;;) {
      const bool building = b < 15;
      if (!building && win < 0) break;
      const bool do_dbl = fix_pending || (building ? (b == 0) : (sub < 5));
      bool step_done;
      if (do_dbl) {
        UJac<CU> t;
        ujac_dbl<CU>(t, q);
        if (fix_pending) {
          u_select(q.x, fix_lane, t.x, q.x);
          u_select(q.y, fix_lane, t.y, q.y);
          u_select(q.z, fix_lane, t.z, q.z);
          fix_pending = false;
          fix_lane = false;
        } else {
          q = t;
        }
        step_done = true;
      } else {
        const bool second = !building && sub == 6;
        uint32_t d = 1;
        bool neg = false;
        if (!building) booth(win, second, d, neg);
        if (second) neg = !neg;  // [x^2]P = -sigma(P) = (beta X, -Y)
        UEntry<CU> e;
        uentryg_load<CU>(e, row(d ? d : 1), second);
        const bool q_inf = u_limbs_all_zero(q.z);
        const bool e_skip = (d == 0) || u_limbs_all_zero(e.z);
        UJac<CU> sum;
        bool hz, rz;
        {
          U<CU, 1, 3> ey;
          ujac_add_raw<CU>(sum, hz, rz, ey, q, e, neg);
        }
        const bool same_x = hz && !q_inf && !e_skip;
        fix_lane = same_x && rz;
        const bool to_inf = same_x && !rz;
        if (to_inf) u_set_zero(sum.z);
        if (__builtin_amdgcn_ballot_w64(q_inf) != 0) {
          UEntry<CU> e2;
          const uint32_t* again = row(d ? d : 1);
          asm volatile("" : "+v"(again));
          uentryg_load<CU>(e2, again, second);
          U<CU, 2, 4> sy;
          u_select(sy, neg, u_neg(e2.y), u_as<2, 4>(e2.y));
          u_select(sum.x, q_inf, e2.x, sum.x);
          u_select(sum.y, q_inf, u_reduce(sy), sum.y);
          u_select(sum.z, q_inf, u_as<UJac<CU>::ZK, UJac<CU>::ZV>(e2.z), sum.z);
        }
        const bool keep = e_skip || fix_lane;
        u_select(q.x, keep, q.x, sum.x);
        u_select(q.y, keep, q.y, sum.y);
        u_select(q.z, keep, q.z, sum.z);
        fix_pending = __builtin_amdgcn_ballot_w64(fix_lane) != 0;
        step_done = !fix_pending;
      }
      if (step_done) {
        if (building) {
          UEntry<CU> e;
          e.x = q.x; e.y = q.y;
          e.z = u_reduce(q.z);
          e.zz = u_fit<1, 3>(u_sqr(e.z));
          e.zzz = u_fit<1, 3>(u_mul(e.zz, e.z));
          uentryg_store<CU>(row(b + 2), e, u_fit<1, 3>(u_mul_k<CU>(e.x, CU::BETA)));
          if (++b == 15) u_set_zero(q.z);
        } else if (sub < 6) {
          ++sub;
        } else {
          sub = 0;
          --win;
        }
      }
    }
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, u_fit<1, 3>(q.z));
      flags[idx] = rejected ? 2 : 0;
    }
  }
}

}  // namespace eccx
