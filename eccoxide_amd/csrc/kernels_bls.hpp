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

// ---- variable base with a co-Z window table (a = 0 curves) --------------------------------------
// The default ladder (k_scalarmul_var_unsat) adds Jacobian table entries with cached Z^2, Z^3: 11
// products + 3 squares per addition.  On a curve with a = 0 the doubling formulas do not involve the
// curve constants, so the whole ladder can run on an isomorphic curve E': y^2 = x^3 + b zeta^6 on which
// every table entry is AFFINE: bring the 16 entries d P = (X_d, Y_d, Z_d) to the common denominator
// zeta = Z_16 -- (X_d l^2, Y_d l^3) with l = zeta / Z_d is an affine point of E' -- add them with the
// mixed addition (8 products + 3 squares, one of them merged), and multiply the result's Z by zeta at
// the end (the trick libsecp256k1 uses for its odd-multiples tables).  The ratios Z_{d+1} / Z_d fall
// out of the table build for free: the mixed addition T_{d+1} = T_d + P has Z_{d+1} = Z_d * H_d.
//   table build   1 doubling + 14 mixed additions; rows (X_d, Y_d, ratio)
//   common Z      backward pass, per entry l *= ratio, l^2, l^3, X l^2, Y l^3 (4 products + 1 square)
//   ladder        per signed 5-bit window 5 doublings + 1 mixed addition (2 with GLV)
// GLV = true is the ECCX_ASSUME_SUBGROUP form (bases in G1): k = k1 + k2 x^2, the second half added
// from (beta x', -y') -- sigma commutes with the isomorphism.
// A base point of order <= 16 makes the build degenerate (some T_d is the point at infinity or +-P,
// a ratio is zero): such units are only MARKED here (flag FLAG_REDO) and recomputed by the generic
// ladder, launched behind this kernel with OPT_ONLY_MARKED -- no valid G1 point takes that path.
constexpr uint8_t FLAG_REDO = 0xFE;

template <class CU>
constexpr int urowc_words() { return ((3 * CU::N + 3) / 4) * 4; }

// r = p + (x2, +-y2, 1), h = x2 Z1^2 - X1 exported (the ratio Z3 / Z1)
template <class CU>
ECCX_DEV void ujac_madd_h(UJac<CU>& r, U<CU, 1, 3>& hout, bool& h_zero, bool& r_zero, const UJac<CU>& p,
                          const U<CU, 1, 3>& x2, const U<CU, 1, 3>& y2, bool neg) {
  auto z1z1 = u_sqr(p.z);
  auto u2 = u_mul(x2, z1z1);
  auto t = u_mul(p.z, z1z1);
  U<CU, 2, 4> sy;
  u_select(sy, neg, u_neg(y2), u_as<2, 4>(y2));
  auto s2 = u_mul(u_reduce(sy), t);
  auto h = u_reduce(u_sub(u2, p.x));
  auto rr = u_reduce(u_sub(s2, p.y));
  h_zero = u_is_zero_mod_p(h);
  r_zero = u_is_zero_mod_p(rr);
  auto hh = u_sqr(h);
  auto hhh = u_mul(h, hh);
  auto v = u_mul(p.x, hh);
  auto r2 = u_sqr(rr);
  auto x3 = u_reduce(u_sub(u_sub(u_sub(r2, hhh), v), v));
  r.x = x3;
  r.y = u_fit<1, 3>(u_mul_add(rr, u_sub(v, x3), u_neg(p.y), hhh));
  r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(p.z, h));
  hout = h;
}

// scratch: [workgroup][row 0..16][thread][urowc_words]; row 0: split scalar (words 0..15, GLV) and
// zeta (words 16..16+N); rows 1..16: the table
template <class CU, class G, bool GLV>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_scalarmul_coz_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                                 const uint8_t* __restrict__ points,
                                                                                 uint32_t* __restrict__ rows_out,
                                                                                 uint8_t* __restrict__ flags,
                                                                                 uint32_t* __restrict__ scratch, uint32_t opts) {
  using CS = typename CU::Sat;
  static_assert(CS::A0 == 1 && CU::KIND == UK_MONT, "the co-Z table needs a = 0 (and uses the merged product)");
  constexpr int L = CS::L;
  constexpr int N = CU::N;
  constexpr int FB = CS::FB;
  constexpr int SB = CS::SB;
  static_assert(!GLV || SB == 32, "the split is written for 256-bit scalars");
  constexpr int NWIN = GLV ? (G::K_BITS + 1 + 4) / 5 : (8 * SB + 1 + 4) / 5;
  constexpr int WR = urowc_words<CU>();
  static_assert(WR >= 16 + N, "row 0 holds the split scalar and zeta");
  using T = U<CU, 1, 3>;
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)WR;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * WR; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
    if constexpr (GLV) {
      uint32_t k1[5], k2[5];
      glv_split<G>(k1, k2, k);
      uint4* d = reinterpret_cast<uint4*>(row(0));
      d[0] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
      d[1] = make_uint4(k1[4], 0u, 0u, 0u);
      d[2] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
      d[3] = make_uint4(k2[4], 0u, 0u, 0u);
    }
    const uint32_t* __restrict__ kw = row(0);

    T one;
#pragma unroll
    for (int i = 0; i < N; ++i) one.v[i] = CU::ONE[i];
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
    q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    u3_store<CU>(row(1), q.x, q.y, one);

    // Booth digit (|digit| in 0..16, sign) of window w: from the scalar bytes, or from half `second`
    // of the split scalar (little-endian words; bit -1 is 0)
    auto booth = [&](int w, bool second, uint32_t& d, bool& neg) {
      uint32_t w6;
      if constexpr (GLV) {
        const uint32_t* __restrict__ h = kw + (second ? 8 : 0);
        const int pos = 5 * w - 1 + 32;  // one zero word below the scalar
        const int wi = pos >> 5, sh = pos & 31;
        const uint32_t lo = (wi >= 1 && wi <= 5) ? h[wi - 1] : 0u;
        const uint32_t hi = (wi <= 4) ? h[wi] : 0u;
        w6 = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & 0x3fu;
      } else {
        const int pos = 5 * w - 1 + 8;
        const int bi = pos >> 3;
        const uint32_t b0 = (bi >= 1 && bi <= SB) ? k[SB - bi] : 0u;
        const uint32_t b1 = (bi + 1 <= SB) ? k[SB - bi - 1] : 0u;
        w6 = ((b0 | (b1 << 8)) >> (pos & 7)) & 0x3fu;
      }
      const uint32_t s = ~((w6 >> 5) - 1u);
      const uint32_t m = (((1u << 6) - w6 - 1u) & s) | (w6 & ~s);
      d = (m >> 1) + (m & 1u);
      neg = (s & 1u) != 0;
    };

    constexpr int LAST_SUB = GLV ? 6 : 5;  // sub 0..4: doublings, 5: addition (6: second half's addition)
    int b = 0;                             // table-build step: 0 -> T[2] = 2P, 1..14 -> T[b+2] = T[b+1] + P
    int win = NWIN - 1, sub = 5;           // the top window needs no doublings
    bool fix_pending = false, fix_lane = false, degenerate = false;
    for (;;) {
      const bool building = b < 15;
      if (!building && win < 0) break;
      const bool do_dbl = fix_pending || (building ? (b == 0) : (sub < 5));
      bool step_done;
      T ratio = one;  // Z_{b+2} / Z_{b+1} of a build step
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
        if (building) {
          ratio = u_reduce(q.z);  // Z_2 / Z_1 = 2 y
          degenerate = degenerate || u_is_zero_mod_p(ratio);
        }
        step_done = true;
      } else {
        const bool second = GLV && !building && sub == 6;
        uint32_t d = 1;
        bool neg = false;
        if (!building) booth(win, second, d, neg);
        if (second) neg = !neg;  // [x^2]P = -sigma(P) = (beta x, -y)
        T ex, ey, eb;
        u3_load<CU>(ex, ey, eb, row(d ? d : 1));
        if (second) ex = eb;
        const bool q_inf = u_limbs_all_zero(q.z);
        const bool e_skip = (d == 0);
        UJac<CU> sum;
        bool hz, rz;
        ujac_madd_h<CU>(sum, ratio, hz, rz, q, ex, ey, neg);
        if (building) {
          // q = d P and the entry is P: any special case here means P has order <= 16
          degenerate = degenerate || hz || q_inf;
          q = sum;
          step_done = true;
        } else {
          const bool same_x = hz && !q_inf && !e_skip;
          fix_lane = same_x && rz;  // q == entry: needs a doubling
          if (same_x && !rz) u_set_zero(sum.z);  // q == -entry
          if (__builtin_amdgcn_ballot_w64(q_inf) != 0) {  // accumulator at infinity: the sum is the entry
            U<CU, 2, 4> sy;
            u_select(sy, neg, u_neg(ey), u_as<2, 4>(ey));
            u_select(sum.x, q_inf, ex, sum.x);
            u_select(sum.y, q_inf, u_reduce(sy), sum.y);
            u_select(sum.z, q_inf, u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one), sum.z);
          }
          const bool keep = e_skip || fix_lane;
          u_select(q.x, keep, q.x, sum.x);
          u_select(q.y, keep, q.y, sum.y);
          u_select(q.z, keep, q.z, sum.z);
          fix_pending = __builtin_amdgcn_ballot_w64(fix_lane) != 0;
          step_done = !fix_pending;
        }
      }
      if (step_done) {
        if (building) {
          u3_store<CU>(row(b + 2), q.x, q.y, ratio);
          if (++b == 15) {
            // ---- bring the table to the common denominator zeta = Z_16 ----
            const T zeta = u_reduce(q.z);
            {
              uint32_t* zdst = row(0) + 16;
#pragma unroll
              for (int i = 0; i < N; ++i) zdst[i] = zeta.v[i];
            }
            T lam = one, next;
            {
              T x16, y16;
              u3_load<CU>(x16, y16, next, row(16));
              if constexpr (GLV) u3_store<CU>(row(16), x16, y16, u_fit<1, 3>(u_mul_k<CU>(x16, CU::BETA)));
            }
            for (int d = 15; d >= 1; --d) {
              lam = u_fit<1, 3>(u_mul(lam, next));
              T x, y;
              u3_load<CU>(x, y, next, row(d));
              const T l2 = u_fit<1, 3>(u_sqr(lam));
              const T l3 = u_fit<1, 3>(u_mul(l2, lam));
              const T xs = u_fit<1, 3>(u_mul(x, l2));
              const T ys = u_fit<1, 3>(u_mul(y, l3));
              if constexpr (GLV) u3_store<CU>(row(d), xs, ys, u_fit<1, 3>(u_mul_k<CU>(xs, CU::BETA)));
              else u3_store<CU>(row(d), xs, ys, one);
            }
            // the top window's (first) addition has nothing to add to: the accumulator starts as that
            // signed entry -- affine on E', so Z = 1 -- or at infinity for digit 0
            uint32_t d;
            bool neg;
            booth(NWIN - 1, false, d, neg);
            T tx, ty, tb;
            u3_load<CU>(tx, ty, tb, row(d ? d : 1));
            U<CU, 2, 4> sy;
            u_select(sy, neg, u_neg(ty), u_as<2, 4>(ty));
            q.x = tx;
            q.y = u_reduce(sy);
            q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
            if (d == 0) u_set_zero(q.z);
            if constexpr (GLV) {
              win = NWIN - 1;
              sub = 6;  // the second half's addition of the top window comes next
            } else {
              win = NWIN - 2;
              sub = 0;
            }
          }
        } else if (sub < LAST_SUB) {
          ++sub;
        } else {
          sub = 0;
          --win;
        }
      }
    }
    // back from E' to the curve: Z *= zeta
    {
      T zeta;
      const uint32_t* zsrc = row(0) + 16;
#pragma unroll
      for (int i = 0; i < N; ++i) zeta.v[i] = zsrc[i];
      q.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(q.z, zeta));
    }
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, u_fit<1, 3>(q.z));
      flags[idx] = rejected ? 2 : (degenerate ? FLAG_REDO : 0);
    }
  }
}

}  // namespace eccx
