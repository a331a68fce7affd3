// Default kernels: ladders on the unsaturated field (ufe.hpp).
//
//   k_scalarmul_var_unsat<CU>      the reference's &Point * &Scalar (src/curve/fiat/curve_macros.rs:
//       321-327) on Jacobian coordinates with signed 5-bit windows (Booth recoding): table d*P,
//       d = 1..16, in a per-lane HBM slab, then per window (MSB first) 5 doublings + 1 addition of
//       +-d*P.  The reference's fixed 4-bit unsigned window (src/curve/projective.rs:871-896)
//       needs 2*SB additions, this needs ceil((8*SB + 1)/5); k*P is the same point either way.
//       <CU, true> is the fused double-scalar form u1*G + u2*Q.
//   k_scalarmul_base_unsat<CU>     mul_base as a 16-bit-window comb over the engine's own table.
//   k_ed_scalarmul_{var,base}_unsat, k_x25519_ladder_unsat   the edwards25519 / curve25519 forms.
//
// The Jacobian addition is not complete: the cases it misses are patched per lane after the
// generic formulas ran -- accumulator at infinity -> take the table entry; digit 0 (or an entry
// at infinity) -> keep the accumulator; equal x and opposite y -> infinity; equal points -> the
// lane keeps its accumulator and the wavefront runs one extra doubling step.  Table build and
// main loop are ONE loop holding a single doubling body and a single addition body chosen by
// wave-uniform control flow.  Only the field layer carries bounds: 28/29-bit limbs,
// single-instruction MACs, real squarings, carry-free additions, bounds tracked in the types
// (U<C, K, V>, see ufe.hpp), so an unsafe composition does not compile.  Results leave as
// canonical limbs of the saturated field (rows X, Y, Z) for k_batch_to_affine.
#pragma once
#include "kernels_fast.hpp"
#include "ufe.hpp"

namespace eccx {
// comb-table entries start on 128-byte boundaries (32 words): the 16-bit-window tables live in
// HBM / Infinity Cache and every entry is a random access, so an entry that straddles a
// sector boundary costs an extra fetch (Ed25519 mul_base: 0.84 -> 0.81 ms per 2^20)
#ifndef ECCX_ENTRY_ALIGN
#define ECCX_ENTRY_ALIGN 32
#endif

template <class CU>
struct UJac {
  // z may stay lazy (twice a product) where the field's columns have room for it
  static constexpr int ZK = UB<CU>::KLAZY, ZV = ZK == 2 ? 4 : 3;
  U<CU, 1, 3> x, y;
  U<CU, ZK, ZV> z;  // infinity <=> every limb of z is zero
};

// table entry: coordinates reduced to tight form, Z^2 and Z^3 cached
template <class CU>
struct UEntry {
  U<CU, 1, 3> x, y, z, zz, zzz;
};

// un-normalised result rows handed to k_batch_to_affine_unsat: three field elements (X, Y, Z) as
// tight digits, padded to 16 bytes
template <class CU>
constexpr int urow3_words() { return ((3 * CU::N + 3) / 4) * 4; }
template <class CU>
ECCX_DEV void u3_store(uint32_t* __restrict__ row, const U<CU, 1, 3>& x, const U<CU, 1, 3>& y, const U<CU, 1, 3>& z) {
  constexpr int N = CU::N;
  constexpr int W = urow3_words<CU>();
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = i < N ? x.v[i] : (i < 2 * N ? y.v[i - N] : (i < 3 * N ? z.v[i - 2 * N] : 0u));
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <class CU>
ECCX_DEV void u3_load(U<CU, 1, 3>& x, U<CU, 1, 3>& y, U<CU, 1, 3>& z, const uint32_t* __restrict__ row) {
  constexpr int N = CU::N;
  constexpr int W = urow3_words<CU>();
  uint32_t w[W];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    const uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { x.v[i] = w[i]; y.v[i] = w[N + i]; z.v[i] = w[2 * N + i]; }
}

// 2P.  a = -3 (dbl-2001-b with Z3 = 2YZ): 4 products + 4 squares; a = 0 (dbl-2009-l): 2
// products + 5 squares.  The typed operations insert a weak reduction wherever an operand's
// bounds require one; the comments give the (K, V) bounds for P-256, where none is inserted
// beyond the three written out.
template <class CU>
ECCX_DEV void ujac_dbl(UJac<CU>& r, const UJac<CU>& p) {
  if constexpr (CU::Sat::A0 && CU::KIND == UK_MONT) {
    // a = 0 on a general Montgomery field (BLS12-381), where a reduction is half of every product:
    // D = 4 X Y^2 as a product, and Y3 = E (D - X3) - 2 (2 Y^2)^2 in ONE reduction on signed columns
    // (u_mul_sub_2sqr) -- 3 products + 4 squares with 6 reductions instead of 2 + 5 with 7
    auto a = u_sqr(p.x);                                  // A = X^2
    auto b = u_sqr(p.y);                                  // B = Y^2
    auto xb = u_mul(p.x, b);
    auto xb2 = u_add(xb, xb);
    auto d = u_reduce(u_add(xb2, xb2));                   // D = 4 X B
    auto e = u_reduce(u_add(u_add(a, a), a));             // E = 3 A
    auto f = u_sqr(e);
    auto x3 = u_reduce(u_sub(u_sub(f, d), d));            // X3 = E^2 - 2 D
    auto yz = u_mul(p.y, p.z);
    r.x = x3;
    r.y = u_mul_sub_2sqr(e, u_sub(d, x3), u_add(b, b));   // E (D - X3) - 8 B^2
    r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_add(yz, yz));
  } else if constexpr (CU::Sat::A0) {
    auto a = u_sqr(p.x);
    auto b = u_sqr(p.y);
    auto c = u_sqr(b);
    auto xb = u_sqr(u_add(p.x, b));
    auto d0 = u_sub(u_sub(xb, a), c);
    auto d = u_reduce(u_add(d0, d0));                    // 2*((X+B)^2 - A - C)
    auto e = u_add(u_add(a, a), a);                      // 3*A
    auto f = u_sqr(e);
    auto x3 = u_reduce(u_sub(u_sub(f, d), d));
    auto c2 = u_add(c, c);
    auto c4 = u_add(c2, c2);
    auto c8 = u_reduce(u_add(c4, c4));
    auto y3m = u_mul(e, u_sub(d, x3));
    auto yz = u_mul(p.y, p.z);
    r.x = x3;
    r.y = u_reduce(u_sub(y3m, c8));
    r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_add(yz, yz));
  } else if constexpr (UB<CU>::KKMAX == 1) {
    // fields whose multiplier wants tight operands (P-521): 4 products + 4 squares, and the
    // multiples of beta and gamma^2 are formed after the product so that each is reduced once
    auto delta = u_sqr(p.z);
    auto gamma = u_sqr(p.y);
    auto beta = u_mul(p.x, gamma);
    auto t2 = u_add(p.x, delta);
    auto alpha = u_mul(u_sub(p.x, delta), u_add(u_add(t2, t2), t2));  // 3*(X-d)(X+d)
    auto b2 = u_add(beta, beta);
    auto b4 = u_reduce(u_add(b2, b2));
    auto x3 = u_reduce(u_sub(u_sub(u_sqr(alpha), b4), b4));
    auto gg = u_sqr(gamma);
    auto gg2 = u_add(gg, gg);
    auto gg4 = u_reduce(u_add(gg2, gg2));
    auto y3m = u_mul(alpha, u_sub(b4, x3));
    auto yz = u_mul(p.y, p.z);  // not (Y+Z)^2 - g - d: Z = 0 must give the all-zero limbs back
    r.x = x3;
    r.y = u_reduce(u_sub(u_sub(y3m, gg4), gg4));
    r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_add(yz, yz));
  } else {
    auto delta = u_sqr(p.z);                    // (1,2)
    auto gamma = u_sqr(p.y);                    // (1,2)
    auto g2 = u_add(gamma, gamma);              // (2,4)
    auto g4 = u_add(g2, g2);                    // (4,8)
    auto b4 = u_mul(p.x, g4);                   // 4*beta             (1,2)
    auto t1 = u_sub(p.x, delta);                // (3,7)
    auto t2 = u_add(p.x, delta);                // (2,5)
    auto t3 = u_mul(t1, t2);                    // (1,3)
    auto alpha = u_reduce(u_add(u_add(t3, t3), t3));  // 3*(X-d)(X+d)   (3,9) -> (1,3)
    auto x3a = u_sqr(alpha);                    // (1,2)
    auto x3 = u_reduce(u_sub(u_sub(x3a, b4), b4));    // alpha^2 - 8*beta  (5,10) -> (1,3)
    auto yz = u_mul(p.y, p.z);                  // (1,2)
    auto t = u_sub(b4, x3);                     // (3,6)
    r.x = x3;
    if constexpr (UB<CU>::SPARSE) {
      // signed columns (P-384): Y3 = alpha*t - 2*(2 gamma)^2 in ONE reduction -- the square's own
      // reduction, its column extractions and the carry chain of the difference all go away
      r.y = u_mul_sub_2sqr(alpha, t, g2);
    } else {
      auto g8s = u_sqr(g2);                     // 4*gamma^2          (1,2)
      auto y3m = u_mul(alpha, t);               // (1,2)
      r.y = u_reduce(u_sub(u_sub(y3m, g8s), g8s));    // ... - 8*gamma^2   (5,10) -> (1,3)
    }
    r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_add(yz, yz));  // 2*Y*Z      (2,4)
  }
}

// r = p + (+-e) with the generic formulas (add-1998-cmo-2, cached Z2^2, Z2^3): 11 products +
// 3 squares.  ey receives the (signed) entry y in tight form.
template <class CU>
ECCX_DEV void ujac_add_raw(UJac<CU>& r, bool& h_zero, bool& r_zero, U<CU, 1, 3>& ey, const UJac<CU>& p,
                           const UEntry<CU>& e, bool neg) {
  auto z1z1 = u_sqr(p.z);
  auto u1 = u_mul(p.x, e.zz);
  auto u2 = u_mul(e.x, z1z1);
  auto s1 = u_mul(p.y, e.zzz);
  auto t = u_mul(p.z, z1z1);
  U<CU, 2, 4> sy;
  u_select(sy, neg, u_neg(e.y), u_as<2, 4>(e.y));
  ey = u_reduce(sy);
  auto s2 = u_mul(ey, t);
  auto h = u_reduce(u_sub(u2, u1));
  auto rr = u_reduce(u_sub(s2, s1));
  h_zero = u_is_zero_mod_p(h);
  r_zero = u_is_zero_mod_p(rr);
  auto hh = u_sqr(h);
  auto hhh = u_mul(h, hh);
  auto v = u_mul(u1, hh);
  auto r2 = u_sqr(rr);
  auto x3 = u_reduce(u_sub(u_sub(u_sub(r2, hhh), v), v));
  auto tt = u_sub(v, x3);
  auto z3a = u_mul(p.z, e.z);
  r.x = x3;
  if constexpr (CU::KIND == UK_MONT) {
    // Y3 = r*(v - x3) + (4p - s1)*h^3: two products, one reduction (the general Montgomery
    // reduction is half of every product there)
    r.y = u_fit<1, 3>(u_mul_add(rr, tt, u_neg(s1), hhh));
  } else if constexpr (UB<CU>::SPARSE) {
    r.y = u_mul_sub(rr, tt, s1, hhh);           // signed columns: r*(v - x3) - s1*h^3 in one reduction
  } else {
    auto y3a = u_mul(rr, tt);
    auto s1h = u_mul(s1, hhh);
    r.y = u_reduce(u_sub(y3a, s1h));
  }
  r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(z3a, h));
}

template <class CU>
constexpr int urow_words() { return ((5 * CU::N + 3) / 4) * 4; }

template <class CU>
ECCX_DEV void uentry_store(uint32_t* __restrict__ row, const UEntry<CU>& p) {
  constexpr int N = CU::N;
  constexpr int W = urow_words<CU>();
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    w[i] = p.x.v[i]; w[N + i] = p.y.v[i]; w[2 * N + i] = p.z.v[i]; w[3 * N + i] = p.zz.v[i]; w[4 * N + i] = p.zzz.v[i];
  }
#pragma unroll
  for (int i = 5 * N; i < W; ++i) w[i] = 0;
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

template <class CU>
ECCX_DEV void uentry_load(UEntry<CU>& p, const uint32_t* __restrict__ row) {
  constexpr int N = CU::N;
  constexpr int W = urow_words<CU>();
  uint32_t w[W];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    p.x.v[i] = w[i]; p.y.v[i] = w[N + i]; p.z.v[i] = w[2 * N + i]; p.zz.v[i] = w[3 * N + i]; p.zzz.v[i] = w[4 * N + i];
  }
}

// waves per SIMD the register allocator is asked to fit (measured)
#ifndef ECCX_OCC_U14
#define ECCX_OCC_U14 2
#endif
#ifndef ECCX_OCC_U18
#define ECCX_OCC_U18 2
#endif
template <class CU>
constexpr int unsat_occupancy() { return CU::N <= 9 ? 4 : (CU::N <= 14 ? ECCX_OCC_U14 : ECCX_OCC_U18); }

template <class CU>
constexpr int utable_words() { return ((2 * CU::N + ECCX_ENTRY_ALIGN - 1) / ECCX_ENTRY_ALIGN) * ECCX_ENTRY_ALIGN; }

// Window width of the fixed-base combs: entry (w, d) = d * 2^(W*w) * G, ceil(8*SB / W) additions.
// Measured on MI355X, ms per 2^20 units (P-256 / Ed25519): W = 8: 2.33 / 1.51, 10: 2.01 / 1.33,
// 12: 1.78 / 1.18, 14: 1.56 / 1.05, 16: 1.37 / 0.97.  Two scalar bytes per window: tables of
// 84-310 MB per curve (HBM has 288 GB), built once per context in 10-160 ms.
#ifndef ECCX_COMB_BITS
#define ECCX_COMB_BITS 16
#endif
template <class CU>
constexpr int comb_bits() { return ECCX_COMB_BITS; }
template <class CU>
constexpr int comb_windows() { return (8 * CU::Sat::SB + comb_bits<CU>() - 1) / comb_bits<CU>(); }
// digit w of the big-endian SB-byte scalar k in radix 2^W
template <int W, int SB>
ECCX_DEV uint32_t comb_digit(const uint8_t* __restrict__ k, int w) {
  static_assert(W >= 1 && W <= 16, "window of at most three bytes");
  const int pos = w * W;
  const int b = pos >> 3;  // byte that holds the lowest bit, counted from the end
  uint32_t v = k[SB - 1 - b];
  if (b + 1 < SB) v |= (uint32_t)k[SB - 2 - b] << 8;
  if (b + 2 < SB) v |= (uint32_t)k[SB - 3 - b] << 16;
  return (v >> (pos & 7)) & ((1u << W) - 1u);
}
// Booth digit of window w of the big-endian SB-byte scalar k, W-bit windows: value d_w in
// [-2^(W-1), 2^(W-1)] with sum d_w 2^(W w) = k, returned as magnitude and sign.  Which bytes are read
// depends on w only.
template <int W, int SB>
ECCX_DEV void booth_digit(const uint8_t* __restrict__ k, int w, uint32_t& d, bool& neg) {
  static_assert(W >= 2 && W <= 8, "a window and its borrow bit must fit two bytes at any bit offset");
  const int pos = W * w - 1 + 8;  // bit position of the borrow bit, counted from one byte below the scalar
  const int bi = pos >> 3;
  const uint32_t b0 = (bi >= 1 && bi <= SB) ? k[SB - bi] : 0u;
  const uint32_t b1 = (bi + 1 <= SB) ? k[SB - bi - 1] : 0u;
  const uint32_t wv = ((b0 | (b1 << 8)) >> (pos & 7)) & ((1u << (W + 1)) - 1u);
  const uint32_t s = ~((wv >> W) - 1u);  // all ones for a negative digit
  const uint32_t m = (((1u << (W + 1)) - wv - 1u) & s) | (wv & ~s);
  d = (m >> 1) + (m & 1u);
  neg = (s & 1u) != 0;
}

template <class CU, bool FROM_INFINITY = false>
ECCX_DEV void ucomb_accumulate(UJac<CU>& q, const uint8_t* __restrict__ k, const uint32_t* __restrict__ table);

// FUSED: the double-scalar "verify shape" u1*G + u2*Q (src/protocol/ecdsa.rs:215) in one pass:
// after the ladder has produced u2*Q in Jacobian form, the 16-bit-window comb of u1*G is
// accumulated onto the same point (base_scalars, utable as for k_scalarmul_base_unsat), so
// there is one normalisation and no intermediate affine points.
template <class CU, bool FUSED = false>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_scalarmul_var_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                        const uint8_t* __restrict__ points,
                                                                        uint32_t* __restrict__ rows_out,
                                                                        uint8_t* __restrict__ flags,
                                                                        uint32_t* __restrict__ scratch, uint32_t opts,
                                                                        const uint8_t* __restrict__ base_scalars = nullptr,
                                                                        const uint32_t* __restrict__ utable = nullptr) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int SB = CS::SB;
  constexpr int NWIN = (8 * SB + 1 + 4) / 5;
  constexpr int W5 = urow_words<CU>();
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)W5;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * W5; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    if (opts & OPT_ONLY_MARKED) {
      // fix-up pass behind the affine-table ladder (kernels_coz.hpp): only the units it marked are redone
      active = active && flags[idx] == FLAG_REDO;
      if (__builtin_amdgcn_ballot_w64(active) == 0) continue;
    }

    UJac<CU> q;
    bool rejected = false;
    if (opts & OPT_BASE_IS_GENERATOR) {
#pragma unroll
      for (int i = 0; i < CU::N; ++i) { q.x.v[i] = CU::GX[i]; q.y.v[i] = CU::GY[i]; }
    } else {
      Fe<L> rx, ry;
      fe_load_be<CS>(rx, points + idx * (size_t)(2 * FB));
      fe_load_be<CS>(ry, points + idx * (size_t)(2 * FB) + FB);
      if (opts & OPT_VALIDATE) {  // curve equation on the saturated twin (affine.rs:103-119)
        Fe<L> mx, my;
        fe_to_mont<CS>(mx, rx);
        fe_to_mont<CS>(my, ry);
        rejected = !(fe_is_canonical<CS>(rx) && fe_is_canonical<CS>(ry) && on_curve<CS>(mx, my));
      }
      q.x = u_as<1, 3>(u_to_mont<CU>(rx));
      q.y = u_as<1, 3>(u_to_mont<CU>(ry));
    }
    if constexpr (FUSED) {
      if (opts & OPT_NEGATE_B) q.y = u_reduce(u_neg(q.y));  // u1*G - u2*Q
    }
    U<CU, 1, 2> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    {
      UEntry<CU> e1;
      e1.x = q.x; e1.y = q.y; e1.z = u_as<1, 3>(one); e1.zz = e1.z; e1.zzz = e1.z;
      uentry_store<CU>(row(1), e1);
    }
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;

    // Booth digit of window w: |digit| in 0..16 and its sign
    auto booth = [&](int w, uint32_t& d, bool& neg) {
      const int pos = 5 * w - 1 + 8;
      const int bi = pos >> 3;
      const uint32_t b0 = (bi >= 1 && bi <= SB) ? k[SB - bi] : 0u;
      const uint32_t b1 = (bi + 1 <= SB) ? k[SB - bi - 1] : 0u;
      const uint32_t w6 = ((b0 | (b1 << 8)) >> (pos & 7)) & 0x3fu;
      const uint32_t s = ~((w6 >> 5) - 1u);
      const uint32_t m = (((1u << 6) - w6 - 1u) & s) | (w6 & ~s);
      d = (m >> 1) + (m & 1u);
      neg = (s & 1u) != 0;
    };
    // table-build step b makes T[b+2]: an even entry by DOUBLING T[(b+2)/2] (a doubling costs half an
    // addition), an odd one as T[b+1] + P with T[b+1] still in q from the step before
    int b = 0;
    int win = NWIN - 1, sub = 5;  // the top window needs no doublings
    bool fix_pending = false, fix_lane = false;
    for (;;) {
      const bool building = b < 15;
      if (!building && win < 0) break;
      const bool do_dbl = fix_pending || (building ? ((b & 1) == 0) : (sub < 5));
      bool step_done;
      if (do_dbl) {
        if (building && !fix_pending && b > 0) {
          UEntry<CU> half;
          uentry_load<CU>(half, row((b + 2) >> 1));
          q.x = half.x; q.y = half.y;
          q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(half.z);
        }
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
        uint32_t d = 1;
        bool neg = false;
        if (!building) booth(win, d, neg);
        UEntry<CU> e;
        uentry_load<CU>(e, row(d ? d : 1));
        const bool q_inf = u_limbs_all_zero(q.z);
        const bool e_skip = (d == 0) || u_limbs_all_zero(e.z);
        UJac<CU> sum;
        bool hz, rz;
        {
          U<CU, 1, 3> ey;
          ujac_add_raw<CU>(sum, hz, rz, ey, q, e, neg);
        }
        const bool same_x = hz && !q_inf && !e_skip;
        fix_lane = same_x && rz;           // q == +-e with equal y: needs a doubling
        const bool to_inf = same_x && !rz; // opposite points
        if (to_inf) u_set_zero(sum.z);
        // accumulator at infinity (the top window; afterwards only after a cancellation): the sum
        // is the (signed) entry itself.  The entry is read again here rather than kept in
        // registers across the addition -- that is what used to push the body into spilling.
        if (__builtin_amdgcn_ballot_w64(q_inf) != 0) {
          UEntry<CU> e2;
          const uint32_t* again = row(d ? d : 1);
          asm volatile("" : "+v"(again));  // a fresh read: do not keep the first copy alive instead
          uentry_load<CU>(e2, again);
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
          e.z = u_reduce(q.z);                       // tight; exact zero stays exact zero
          e.zz = u_fit<1, 3>(u_sqr(e.z));
          e.zzz = u_fit<1, 3>(u_mul(e.zz, e.z));
          uentry_store<CU>(row(b + 2), e);
          if (++b == 15) {
            // the top window has nothing to double and nothing to add to: the accumulator STARTS as
            // its signed table entry (infinity for digit 0) and the loop goes on with the next window
            uint32_t d;
            bool neg;
            booth(NWIN - 1, d, neg);
            UEntry<CU> top;
            uentry_load<CU>(top, row(d ? d : 1));
            U<CU, 2, 4> sy;
            u_select(sy, neg, u_neg(top.y), u_as<2, 4>(top.y));
            q.x = top.x;
            q.y = u_reduce(sy);
            q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(top.z);
            if (d == 0) u_set_zero(q.z);
            win = NWIN - 2;
            sub = 0;
          }
        } else if (sub < 5) {
          ++sub;
        } else {
          sub = 0;
          --win;
        }
      }
    }
    if constexpr (FUSED) ucomb_accumulate<CU, false>(q, base_scalars + idx * (size_t)SB, utable);
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, u_fit<1, 3>(q.z));
      flags[idx] = rejected ? 2 : 0;
    }
  }
}

// ---- fixed base (comb) on the unsaturated field -------------------------------------------

// r = p + (x2, y2, 1): madd with the affine table entry, 8 products + 3 squares
template <class CU>
ECCX_DEV void ujac_madd_raw(UJac<CU>& r, bool& h_zero, bool& r_zero, const UJac<CU>& p, const U<CU, 1, 2>& x2,
                            const U<CU, 1, 2>& y2) {
  auto z1z1 = u_sqr(p.z);
  auto u2 = u_mul(x2, z1z1);
  auto t = u_mul(p.z, z1z1);
  auto s2 = u_mul(y2, t);
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
  if constexpr (CU::KIND == UK_MONT) {
    r.y = u_fit<1, 3>(u_mul_add(rr, u_sub(v, x3), u_neg(p.y), hhh));  // one reduction for both products
  } else if constexpr (UB<CU>::SPARSE) {
    r.y = u_mul_sub(rr, u_sub(v, x3), p.y, hhh);
  } else {
    auto y3a = u_mul(rr, u_sub(v, x3));
    auto y1h = u_mul(p.y, hhh);
    r.y = u_reduce(u_sub(y3a, y1h));
  }
  r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(p.z, h));
}

// affine points as canonical big-endian bytes (x | y, what the engine's own variable-base path
// produces for the table scalars) -> table entries in the unsaturated field's form
template <class CU>
__global__ void k_affine_to_utable(size_t entries, const uint8_t* __restrict__ affine, uint32_t* __restrict__ utable) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int UW = utable_words<CU>();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= entries) return;
  Fe<L> px, py;
  fe_load_be<CS>(px, affine + i * (size_t)(2 * FB));
  fe_load_be<CS>(py, affine + i * (size_t)(2 * FB) + FB);
  const auto ux = u_to_mont<CU>(px);
  const auto uy = u_to_mont<CU>(py);
  uint32_t* o = utable + i * UW;
#pragma unroll
  for (int k = 0; k < UW; ++k) o[k] = k < CU::N ? ux.v[k] : (k < 2 * CU::N ? uy.v[k - CU::N] : 0u);
}

// q += sum over the scalar's bytes of table[(w, byte)]: the comb loop shared by the fixed-base
// kernel and the fused double-scalar kernel.  Jacobian mixed additions with the same special
// cases as the variable-base ladder (accumulator at infinity, digit 0, q == -entry, q == entry).
// FROM_INFINITY: q is the point at infinity on entry (the fixed-base kernel), so window 0 is LOADED
// (the entry itself with Z = 1, or infinity for digit 0) instead of being added
template <class CU, bool FROM_INFINITY>
ECCX_DEV void ucomb_accumulate(UJac<CU>& q, const uint8_t* __restrict__ k, const uint32_t* __restrict__ table) {
  using CS = typename CU::Sat;
  constexpr int SB = CS::SB;
  constexpr int W = comb_bits<CU>();
  constexpr int NW = comb_windows<CU>();
  constexpr int UW = utable_words<CU>();
  U<CU, 1, 2> one;
#pragma unroll
  for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
  auto entry = [&](int w, uint32_t& d, U<CU, 1, 2>& x2, U<CU, 1, 2>& y2) {
    d = comb_digit<W, SB>(k, w);
    const uint4* __restrict__ e = reinterpret_cast<const uint4*>(table + (((size_t)w << W) + (d ? d : 1)) * UW);
    constexpr int LW = ((2 * CU::N + 3) / 4) * 4;  // words actually read (the rest of the entry is padding)
    uint32_t ew[LW];
#pragma unroll
    for (int i = 0; i < LW / 4; ++i) {
      const uint4 v = e[i];
      ew[4 * i] = v.x; ew[4 * i + 1] = v.y; ew[4 * i + 2] = v.z; ew[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < CU::N; ++i) { x2.v[i] = ew[i]; y2.v[i] = ew[CU::N + i]; }
  };
  if constexpr (FROM_INFINITY) {
    uint32_t d;
    U<CU, 1, 2> x2, y2;
    entry(0, d, x2, y2);
    q.x = u_as<1, 3>(x2);
    q.y = u_as<1, 3>(y2);
    q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    if (d == 0) u_set_zero(q.z);
  }
  for (int w = FROM_INFINITY ? 1 : 0; w < NW; ++w) {
    uint32_t d;
    U<CU, 1, 2> x2, y2;
    entry(w, d, x2, y2);
    const bool q_inf = u_limbs_all_zero(q.z);
    const bool e_skip = (d == 0);
    UJac<CU> sum;
    bool hz, rz;
    ujac_madd_raw<CU>(sum, hz, rz, q, x2, y2);
    const bool same_x = hz && !q_inf && !e_skip;
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
    u_select(q.x, e_skip, q.x, sum.x);
    u_select(q.y, e_skip, q.y, sum.y);
    u_select(q.z, e_skip, q.z, sum.z);
  }
}

// Fixed-base comb: the reference's mul_base (src/curve/fiat/curve_macros.rs) adds one table
// entry per 4-bit window with no doublings; the result k*G does not depend on the window width,
// so this kernel uses W = comb_bits() = 16 bit windows -- entry (w, d) = d * 2^(W*w) * G --
// from a table the engine builds for itself (SB/2 x 65536 entries).  Jacobian mixed
// additions; same special cases as the variable-base ladder.
template <class CU>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_scalarmul_base_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                                  const uint32_t* __restrict__ table,
                                                                                  uint32_t* __restrict__ rows_out,
                                                                                  uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int SB = CS::SB;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
    U<CU, 1, 2> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    UJac<CU> q;
    q.x = u_as<1, 3>(one);
    q.y = u_as<1, 3>(one);
    u_set_zero(q.z);  // infinity
    ucomb_accumulate<CU, true>(q, k, table);
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, u_fit<1, 3>(q.z));
      flags[idx] = 0;
    }
  }
}

// ---- curve25519 x-only Montgomery ladder (X25519) on the unsaturated field -----------------
//   MontgomeryPoint::scale_bytes -> ladder   src/curve/curve25519.rs:535-541, :474-513
//   protocol::x25519::x25519 (clamp, decode_u)  src/protocol/x25519.rs:14-45   [OPT_X25519_RFC]
// One differential add-and-double per scalar bit, MSB first, with the reference's conditional
// swaps; writes (X2, X2, Z2) rows of canonical integers for k_batch_to_affine<NORM_MONTGOMERY_U>,
// which applies invert_or_zero (Z = 0 -> u = 0).  Per step 5 products, 4 squares, one
// small multiple and two weak reductions; sums and differences stay lazy.
template <class CU>
__global__ void __launch_bounds__(WG, 4) k_x25519_ladder_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                               const uint8_t* __restrict__ u_in,
                                                               uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags,
                                                               uint32_t opts) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  using T = U<CU, 1, 3>;
  const bool rfc = (opts & OPT_X25519_RFC) != 0;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    T x1, x2, z2, x3, z3;
    {
      Fe<L> raw;
      if (u_in) {
        fe_load_le<CS>(raw, u_in + idx * 32);
        if (rfc) raw.v[L - 1] &= 0x7fffffffu;  // decode_u masks the unused top bit
      } else {
        fe_zero<CS>(raw);
        raw.v[0] = 9;  // MONT_GU_BYTES
      }
      // any 256-bit string is accepted and taken modulo p
      x1 = u_reduce(u_as<1, 3>(u_from_sat<CU>(raw)));
    }
    u_set_zero(x2);
    x2.v[0] = 1;
    u_set_zero(z2);
    x3 = x1;
    u_set_zero(z3);
    z3.v[0] = 1;
    const uint8_t* __restrict__ k = scalars + idx * 32;
    uint32_t swap = 0;
    for (int t = 255; t >= 0; --t) {
      uint32_t bit;
      if (rfc) {  // little-endian scalar, clamped: k[0] &= 248, k[31] &= 127, k[31] |= 64
        bit = (k[t >> 3] >> (t & 7)) & 1u;
        if (t == 255 || t < 3) bit = 0;
        if (t == 254) bit = 1;
      } else {    // the big-endian string the ladder consumes
        bit = (k[31 - (t >> 3)] >> (t & 7)) & 1u;
      }
      swap ^= bit;
      {
        const bool sw = swap != 0;
        T a = x2, b = x3;
        u_select(x2, sw, b, a);
        u_select(x3, sw, a, b);
        a = z2; b = z3;
        u_select(z2, sw, b, a);
        u_select(z3, sw, a, b);
      }
      swap = bit;
      auto a = u_add(x2, z2);
      auto aa = u_sqr(a);
      auto b = u_reduce(u_sub(x2, z2));
      auto bb = u_sqr(b);
      auto e = u_sub(aa, bb);
      auto c = u_add(x3, z3);
      auto d = u_sub(x3, z3);
      auto da = u_mul(d, a);
      auto cb = u_mul(c, b);
      x3 = u_fit<1, 3>(u_sqr(u_add(da, cb)));
      z3 = u_fit<1, 3>(u_mul(x1, u_sqr(u_sub(da, cb))));
      x2 = u_fit<1, 3>(u_mul(aa, bb));
      auto s = u_add(bb, u_mul_small(e, 121666u));  // bb + a24 * e
      z2 = u_fit<1, 3>(u_mul(e, s));
    }
    {
      const bool sw = swap != 0;
      T a = x2, b = z2;
      u_select(x2, sw, x3, a);
      u_select(z2, sw, z3, b);
    }
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), x2, x2, z2);
      flags[idx] = 0;
    }
  }
}

// ---- edwards25519 fixed base, 16-bit windows, unsaturated field ------------------------------
// The reference's mul_base (curve25519.rs:840-851) adds one table entry per 4-bit window; k*B
// does not depend on the window width, so the default path uses 16-bit windows -- entry (w, d) =
// d * 2^(16 w) * B as the triple (y - x, y + x, 2d*x*y), d = 0 being the neutral (1, 1, 0) -- from a
// table the engine builds for itself (16 x 65536 entries of 9 x 29-bit digits, 28 words each,
// 117 MB): 16 seven-product additions instead of 64.  The only carry chain
// per addition is the reduction of F = 2Z - C, which feeds two products.
constexpr int ED_U_ENTRY_WORDS = ((27 + ECCX_ENTRY_ALIGN - 1) / ECCX_ENTRY_ALIGN) * ECCX_ENTRY_ALIGN;

// (x, y, z, t) += the table point given as (y2 - x2, y2 + x2, 2d*x2*y2), Z2 = 1: Point::add with
// the operand prepared in the table (curve25519.rs:695-729), 7 products
template <class CU, int KT = 1, int VT = 3>
ECCX_DEV void ued_add_niels(U<CU, 1, 3>& qx, U<CU, 1, 3>& qy, U<CU, 1, 3>& qz, U<CU, 1, 3>& qt, const U<CU, 1, 3>& ym,
                            const U<CU, 1, 3>& yp, const U<CU, KT, VT>& t2d) {
  auto aa = u_mul(u_sub(qy, qx), ym);
  auto bb = u_mul(u_add(qy, qx), yp);
  auto cc = u_mul(qt, t2d);
  auto dd = u_add(qz, qz);
  auto e = u_sub(bb, aa);
  auto f = u_reduce(u_sub(dd, cc));
  auto g = u_add(dd, cc);
  auto h = u_add(bb, aa);
  qx = u_fit<1, 3>(u_mul(e, f));
  qy = u_fit<1, 3>(u_mul(g, h));
  qz = u_fit<1, 3>(u_mul(f, g));
  qt = u_fit<1, 3>(u_mul(e, h));
}

// q += sum over the 16-bit windows of k of table[(w, digit)] (complete additions).
// FROM_NEUTRAL: q is the neutral element on entry, so window 0 is not added but LOADED: with
// (ym, yp) = (y - x, y + x) the point (x, y) is (2(yp - ym) : 2(yp + ym) : 4 : (yp - ym)(yp + ym)) in
// extended coordinates -- one product instead of the seven of an addition.
template <class CU, bool FROM_NEUTRAL = false>
ECCX_DEV void ued_comb_accumulate(U<CU, 1, 3>& qx, U<CU, 1, 3>& qy, U<CU, 1, 3>& qz, U<CU, 1, 3>& qt,
                                  const uint8_t* __restrict__ k, const uint32_t* __restrict__ table) {
  constexpr int N = CU::N;
  constexpr int W = comb_bits<CU>();
  auto entry = [&](int w, U<CU, 1, 3>& ym, U<CU, 1, 3>& yp, U<CU, 1, 3>& t2d) {
    const uint32_t d = comb_digit<W, 32>(k, w);  // the big-endian scalar string (curve25519.rs:842-846)
    const uint4* __restrict__ e4 = reinterpret_cast<const uint4*>(table + (((size_t)w << W) + d) * ED_U_ENTRY_WORDS);
    constexpr int LW = ((3 * N + 3) / 4) * 4;  // words actually read (the rest of the entry is padding)
    uint32_t ew[LW];
#pragma unroll
    for (int i = 0; i < LW / 4; ++i) {
      const uint4 v = e4[i];
      ew[4 * i] = v.x; ew[4 * i + 1] = v.y; ew[4 * i + 2] = v.z; ew[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { ym.v[i] = ew[i]; yp.v[i] = ew[N + i]; t2d.v[i] = ew[2 * N + i]; }
  };
  if constexpr (FROM_NEUTRAL) {
    U<CU, 1, 3> ym, yp, t2d;
    entry(0, ym, yp, t2d);
    const auto dx = u_reduce(u_sub(yp, ym));  // 2x
    const auto sy = u_reduce(u_add(yp, ym));  // 2y
    qx = u_reduce(u_add(dx, dx));
    qy = u_reduce(u_add(sy, sy));
    u_set_zero(qz);
    qz.v[0] = 4;
    qt = u_fit<1, 3>(u_mul(dx, sy));          // 4xy = X Y / Z
  }
  for (int w = FROM_NEUTRAL ? 1 : 0; w < comb_windows<CU>(); ++w) {
    U<CU, 1, 3> ym, yp, t2d;
    entry(w, ym, yp, t2d);
    ued_add_niels<CU>(qx, qy, qz, qt, ym, yp, t2d);
  }
}
template <class CU>
__global__ void k_ed_affine_to_niels_unsat(size_t entries, const uint8_t* __restrict__ affine, uint32_t* __restrict__ table,
                                           int stride_words) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int N = CU::N;
  static_assert(3 * N <= ED_U_ENTRY_WORDS, "entry layout");
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= entries) return;
  Fe<L> fx, fy;
  fe_load_le<CS>(fx, affine + i * 64);
  fe_load_le<CS>(fy, affine + i * 64 + 32);
  const auto x = u_from_sat<CU>(fx);
  const auto y = u_from_sat<CU>(fy);
  U<CU, 1, 2> d2;
#pragma unroll
  for (int k = 0; k < N; ++k) d2.v[k] = CU::D2[k];
  const auto ym = u_reduce(u_sub(y, x));
  const auto yp = u_reduce(u_add(y, x));
  const auto t2d = u_reduce(u_mul(u_mul(x, y), d2));
  uint32_t* o = table + i * (size_t)stride_words;
  for (int k = 0; k < stride_words; ++k)
    o[k] = k < N ? ym.v[k] : (k < 2 * N ? yp.v[k - N] : (k < 3 * N ? t2d.v[k - 2 * N] : 0u));
}

template <class CU>
__global__ void __launch_bounds__(WG, 4) k_ed_scalarmul_base_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                   const uint32_t* __restrict__ table,
                                                                   uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags) {
  using T = U<CU, 1, 3>;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * 32;
    T qx, qy, qz, qt;  // the neutral element (0, 1, 1, 0)
    u_set_zero(qx); u_set_zero(qy); u_set_zero(qz); u_set_zero(qt);
    qy.v[0] = 1; qz.v[0] = 1;
    ued_comb_accumulate<CU, true>(qx, qy, qz, qt, k, table);
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), qx, qy, qz);
      flags[idx] = 0;
    }
  }
}

// ---- edwards25519 variable base, signed 5-bit windows, unsaturated field ---------------------
// The reference's Point::scale (src/curve/curve25519.rs:746-762) is a 256-step double-and-add
// with the unified addition (:695-710).  k*P does not depend on the addition chain, so the
// default path uses the same signed 5-bit windows as the Weierstrass ladder: table d*P,
// d = 0..16, in a per-lane HBM slab as (Y - X, Y + X, 2Z, 2d*T); 52 windows of 5 doublings +
// 1 addition.  The a = -1 twisted Edwards formulas are complete on this curve (d is not a
// square), so there are no special cases at all: digit 0 adds the neutral entry, a negative
// digit swaps Y - X with Y + X and negates 2d*T.  Doubling: 4 squares + 3 products (+ 1 for T
// when an addition follows); addition: 8 products (7 in the main loop, where T is not needed).
template <class CU>
struct UEd {
  U<CU, 1, 3> x, y, z, t;
};
template <class CU>
struct UEdCached {
  U<CU, 1, 3> ym, yp, z2, t2d;
};
constexpr int ED_VAR_ROW_WORDS = 36;  // 4 x 9 digits

template <class CU, bool WITH_T>
ECCX_DEV void ued_dbl(UEd<CU>& r, const UEd<CU>& p) {
  auto a = u_sqr(p.x);
  auto b = u_sqr(p.y);
  auto zz = u_sqr(p.z);
  auto xy = u_sqr(u_add(p.x, p.y));
  // two carry chains (F and H) are enough: E (K = 5) and G (K = 3) only meet tight factors
  auto e = u_sub(u_sub(xy, a), b);                // E = (X+Y)^2 - A - B
  auto g = u_sub(b, a);                           // G = D + B = B - A      (D = -A)
  auto f = u_reduce(u_sub(u_sub(g, zz), zz));     // F = G - 2 Z^2
  auto h = u_reduce(u_sub(u_neg(a), b));          // H = D - B = -(A + B)
  r.x = u_fit<1, 3>(u_mul(e, f));
  r.y = u_fit<1, 3>(u_mul(g, h));
  r.z = u_fit<1, 3>(u_mul(f, g));
  if constexpr (WITH_T) r.t = u_fit<1, 3>(u_mul(e, h));
}

template <class CU, bool WITH_T>
ECCX_DEV void ued_add(UEd<CU>& r, const UEd<CU>& p, const UEdCached<CU>& c, bool neg) {
  U<CU, 1, 3> ym, yp;
  u_select(ym, neg, c.yp, c.ym);
  u_select(yp, neg, c.ym, c.yp);
  auto aa = u_mul(u_sub(p.y, p.x), ym);
  auto bb = u_mul(u_add(p.y, p.x), yp);
  auto cc = u_mul(p.t, c.t2d);                    // sign applied below: C -> -C swaps F and G
  auto dd = u_mul(p.z, c.z2);
  auto e = u_sub(bb, aa);
  auto h = u_add(bb, aa);
  auto dmc = u_fit<3, 8>(u_sub(dd, cc));
  auto dpc = u_fit<3, 8>(u_add(dd, cc));
  U<CU, 3, 8> fl, g;
  u_select(fl, neg, dpc, dmc);                    // F = D - C
  u_select(g, neg, dmc, dpc);                     // G = D + C
  auto f = u_reduce(fl);
  r.x = u_fit<1, 3>(u_mul(e, f));
  r.y = u_fit<1, 3>(u_mul(g, h));
  r.z = u_fit<1, 3>(u_mul(f, g));
  if constexpr (WITH_T) r.t = u_fit<1, 3>(u_mul(e, h));
}

// r = p + (+-)(the table point with Z = 1 given as y - x, y + x, 2d x y): 6 products, T not produced; the sign of the
// digit is applied through the opaque selects (secret-scalar ladder over its normalised table)
template <class CU>
ECCX_DEV void ued_add_niels_ct(UEd<CU>& r, const UEd<CU>& p, const U<CU, 1, 3>& ym_, const U<CU, 1, 3>& yp_, const U<CU, 1, 3>& t2d,
                               bool neg) {
  U<CU, 1, 3> ym, yp;
  u_select_ct(ym, neg, yp_, ym_);
  u_select_ct(yp, neg, ym_, yp_);
  auto aa = u_mul(u_sub(p.y, p.x), ym);
  auto bb = u_mul(u_add(p.y, p.x), yp);
  auto cc = u_mul(p.t, t2d);
  auto dd = u_add(p.z, p.z);
  auto e = u_sub(bb, aa);
  auto h = u_add(bb, aa);
  auto dmc = u_fit<3, 8>(u_sub(dd, cc));
  auto dpc = u_fit<3, 8>(u_add(dd, cc));
  U<CU, 3, 8> fl, g;
  u_select_ct(fl, neg, dpc, dmc);
  u_select_ct(g, neg, dmc, dpc);
  auto f = u_reduce(fl);
  r.x = u_fit<1, 3>(u_mul(e, f));
  r.y = u_fit<1, 3>(u_mul(g, h));
  r.z = u_fit<1, 3>(u_mul(f, g));
}

// ued_add<CU, false> with the sign of the digit applied through the opaque selects (secret-scalar ladder)
template <class CU>
ECCX_DEV void ued_add_ct(UEd<CU>& r, const UEd<CU>& p, const UEdCached<CU>& c, bool neg) {
  U<CU, 1, 3> ym, yp;
  u_select_ct(ym, neg, c.yp, c.ym);
  u_select_ct(yp, neg, c.ym, c.yp);
  auto aa = u_mul(u_sub(p.y, p.x), ym);
  auto bb = u_mul(u_add(p.y, p.x), yp);
  auto cc = u_mul(p.t, c.t2d);
  auto dd = u_mul(p.z, c.z2);
  auto e = u_sub(bb, aa);
  auto h = u_add(bb, aa);
  auto dmc = u_fit<3, 8>(u_sub(dd, cc));
  auto dpc = u_fit<3, 8>(u_add(dd, cc));
  U<CU, 3, 8> fl, g;
  u_select_ct(fl, neg, dpc, dmc);
  u_select_ct(g, neg, dmc, dpc);
  auto f = u_reduce(fl);
  r.x = u_fit<1, 3>(u_mul(e, f));
  r.y = u_fit<1, 3>(u_mul(g, h));
  r.z = u_fit<1, 3>(u_mul(f, g));
}

template <class CU>
ECCX_DEV void ued_cache(UEdCached<CU>& c, const UEd<CU>& p) {
  U<CU, 1, 2> d2;
#pragma unroll
  for (int k = 0; k < CU::N; ++k) d2.v[k] = CU::D2[k];
  c.ym = u_reduce(u_sub(p.y, p.x));
  c.yp = u_reduce(u_add(p.y, p.x));
  c.z2 = u_reduce(u_add(p.z, p.z));
  c.t2d = u_fit<1, 3>(u_mul(p.t, d2));
}

template <class CU>
ECCX_DEV void ued_row_store(uint32_t* __restrict__ row, const UEdCached<CU>& c) {
  constexpr int N = CU::N;
  static_assert(4 * N == ED_VAR_ROW_WORDS, "row layout");
  uint32_t w[ED_VAR_ROW_WORDS];
#pragma unroll
  for (int i = 0; i < N; ++i) { w[i] = c.ym.v[i]; w[N + i] = c.yp.v[i]; w[2 * N + i] = c.z2.v[i]; w[3 * N + i] = c.t2d.v[i]; }
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < ED_VAR_ROW_WORDS / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <class CU>
ECCX_DEV void ued_row_load(UEdCached<CU>& c, const uint32_t* __restrict__ row) {
  constexpr int N = CU::N;
  uint32_t w[ED_VAR_ROW_WORDS];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < ED_VAR_ROW_WORDS / 4; ++i) {
    const uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { c.ym.v[i] = w[i]; c.yp.v[i] = w[N + i]; c.z2.v[i] = w[2 * N + i]; c.t2d.v[i] = w[3 * N + i]; }
}

// FUSED: the verify shape [u1]B + [u2]A ([u1]B - [u2]A with OPT_NEGATE_B; src/protocol/ed25519.rs:145)
// in one pass: the ladder keeps T through its additions and the 16-bit comb of u1*B is accumulated
// onto its result (base_scalars, utable as for k_ed_scalarmul_base_unsat).
// CT = true (ECCX_CT_SCAN): the same ladder for secret scalars -- WB-bit windows (narrower: every lookup reads ALL
// 2^(WB-1) rows of the lane's table and keeps one by v_cndmask, so the table's traffic grows with its height), the
// sign applied by selects, digit 0 = the neutral element the scan starts from.  The unified addition is complete
// on this curve: there is nothing else to resolve, and no branch or address depends on the scalar.
#ifndef ECCX_CT_ED_VAR_BITS
#define ECCX_CT_ED_VAR_BITS 3
#endif
template <class CU, bool FUSED = false, int WB = 5, bool CT = false>
__global__ void __launch_bounds__(WG, 4) k_ed_scalarmul_var_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                  const uint8_t* __restrict__ points,
                                                                  uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags,
                                                                  uint32_t* __restrict__ scratch, uint32_t opts,
                                                                  const uint8_t* __restrict__ base_scalars = nullptr,
                                                                  const uint32_t* __restrict__ utable = nullptr) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int N = CU::N;
  static_assert(!(CT && FUSED) && WB >= 2 && WB <= 5, "the secret-scalar form is the plain ladder");
  constexpr int NWIN = (8 * 32 + 1 + WB - 1) / WB;
  constexpr int TBL = 1 << (WB - 1);  // table rows 1 .. TBL (row 0: the neutral element)
  constexpr int RW = ED_VAR_ROW_WORDS;
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)RW;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * RW; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    UEd<CU> p;
    bool rejected = false;
    if (opts & OPT_BASE_IS_GENERATOR) {
#pragma unroll
      for (int i = 0; i < N; ++i) { p.x.v[i] = CU::GX[i]; p.y.v[i] = CU::GY[i]; }
    } else {
      Fe<L> rx, ry;
      fe_load_le<CS>(rx, points + idx * 64);
      fe_load_le<CS>(ry, points + idx * 64 + 32);
      if (opts & OPT_VALIDATE) {
        // -x^2 + y^2 == 1 + d x^2 y^2  <=>  2(y^2 - x^2 - 1) == 2d x^2 y^2  (curve25519.rs:649-660),
        // on the saturated twin
        Fe<L> mx, my, xx, yy, lhs, rhs, one;
        fe_to_mont<CS>(mx, rx);
        fe_to_mont<CS>(my, ry);
        fe_sqr<CS>(xx, mx);
        fe_sqr<CS>(yy, my);
        fe_set<CS>(one, CS::ONE);
        fe_sub<CS>(lhs, yy, xx);
        fe_sub<CS>(lhs, lhs, one);
        fe_add<CS>(lhs, lhs, lhs);
        fe_mul<CS>(rhs, xx, yy);
        fe_mul_k<CS>(rhs, rhs, CS::D2);
        rejected = !(fe_is_canonical<CS>(rx) && fe_is_canonical<CS>(ry) && fe_eq<CS>(lhs, rhs));
      }
      p.x = u_reduce(u_as<1, 3>(u_from_sat<CU>(rx)));
      p.y = u_reduce(u_as<1, 3>(u_from_sat<CU>(ry)));
    }
    if constexpr (FUSED) {
      if (opts & OPT_NEGATE_B) p.x = u_reduce(u_neg(p.x));  // -(x, y) = (-x, y) (curve25519.rs:731-738)
    }
    u_set_zero(p.z);
    p.z.v[0] = 1;
    p.t = u_fit<1, 3>(u_mul(p.x, p.y));  // from_affine (curve25519.rs:638-645)
    UEdCached<CU> c1;
    ued_cache<CU>(c1, p);
    ued_row_store<CU>(row(1), c1);
    {
      UEdCached<CU> c0;  // the neutral element (0, 1, 1, 0): (1, 1, 2, 0)
      u_set_zero(c0.ym); u_set_zero(c0.yp); u_set_zero(c0.z2); u_set_zero(c0.t2d);
      c0.ym.v[0] = 1; c0.yp.v[0] = 1; c0.z2.v[0] = 2;
      ued_row_store<CU>(row(0), c0);
    }
    // table: T[2] = 2P, T[d+1] = T[d] + P
    UEd<CU> q;
    ued_dbl<CU, true>(q, p);
    for (int d = 2; d <= TBL; ++d) {
      UEdCached<CU> c;
      ued_cache<CU>(c, q);
      ued_row_store<CU>(row(d), c);
      if (d < TBL) {
        UEd<CU> s;
        ued_add<CU, true>(s, q, c1, false);
        q = s;
      }
    }
    // Secret-scalar form: every lookup reads the WHOLE table, so the table is made small -- each entry brought to Z = 1
    // (one inversion per unit by division steps, shared by Montgomery's trick) and written as (y - x, y + x, 2d x y), 28
    // words, into the rows of the slab the build leaves free: 112 B per entry instead of 144, and an addition of 6
    // products instead of 7.
    constexpr int NW = ((3 * N + 3) / 4) * 4;
    static_assert(!CT || (size_t)TBL * NW <= (size_t)(FAST_TABLE_ROWS - 1 - TBL) * RW, "the compact table must fit the free rows");
    uint32_t* cslab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS + (TBL + 1)) * WG * (size_t)RW + (size_t)threadIdx.x * NW;
    auto crow = [&](uint32_t e) { return cslab + (size_t)(e - 1) * WG * NW; };
    if constexpr (CT) {
      auto store3 = [&](uint32_t e, const U<CU, 1, 3>& a, const U<CU, 1, 3>& b, const U<CU, 1, 3>& c) {
        uint32_t w[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = i < N ? a.v[i] : (i < 2 * N ? b.v[i - N] : (i < 3 * N ? c.v[i - 2 * N] : 0u));
        uint4* dst = reinterpret_cast<uint4*>(crow(e));
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
      };
      store3(1, c1.ym, c1.yp, c1.t2d);  // P itself: Z = 1 already
      if constexpr (TBL >= 2) {
        U<CU, 1, 3> pre[TBL];  // pre[d] = z2_2 * ... * z2_d (z2 = 2 Z)
#pragma unroll
        for (int d = 2; d <= TBL; ++d) {
          UEdCached<CU> c;
          ued_row_load<CU>(c, row(d));
          pre[d - 1] = d == 2 ? c.z2 : u_fit<1, 3>(u_mul(pre[d - 2], c.z2));
        }
        U<CU, 1, 3> inv;
        {
          Fe<L> cz;
          u_to_canonical<CU>(cz, pre[TBL - 1]);
          fe_inv_gcd<CS>(cz, cz);
          inv = u_reduce(u_as<1, 3>(u_from_sat<CU>(cz)));
        }
#pragma unroll
        for (int d = TBL; d >= 2; --d) {
          UEdCached<CU> c;
          ued_row_load<CU>(c, row(d));
          U<CU, 1, 3> zi;  // 1 / (2 Z_d)
          if (d > 2) {
            zi = u_fit<1, 3>(u_mul(inv, pre[d - 2]));
            inv = u_fit<1, 3>(u_mul(inv, c.z2));
          } else {
            zi = inv;
          }
          const auto w2 = u_reduce(u_add(zi, zi));  // 1 / Z_d
          store3((uint32_t)d, u_fit<1, 3>(u_mul(c.ym, w2)), u_fit<1, 3>(u_mul(c.yp, w2)), u_fit<1, 3>(u_mul(c.t2d, w2)));
        }
      }
    }
    const uint8_t* __restrict__ k = scalars + idx * 32;
    auto booth = [&](int w, uint32_t& d, bool& neg) { booth_digit<WB, 32>(k, w, d, neg); };
    // the neutral element
    u_set_zero(q.x); u_set_zero(q.y); u_set_zero(q.z); u_set_zero(q.t);
    q.y.v[0] = 1; q.z.v[0] = 1;
    for (int win = NWIN - 1; win >= 0; --win) {
      if (win != NWIN - 1) {
        for (int j = 0; j < WB - 1; ++j) {
          UEd<CU> s;
          ued_dbl<CU, false>(s, q);
          q.x = s.x; q.y = s.y; q.z = s.z;
        }
        UEd<CU> s;
        ued_dbl<CU, true>(s, q);
        q = s;
      }
      uint32_t d;
      bool neg;
      booth(win, d, neg);
      UEdCached<CU> c;
      UEd<CU> s;
      if constexpr (CT) {
        // select_from_table (curve25519.rs:862-869): every row read, the digit's row kept; no match = the neutral element
        U<CU, 1, 3> sym, syp, st;
        u_set_zero(sym); u_set_zero(syp); u_set_zero(st);
        sym.v[0] = 1; syp.v[0] = 1;
#pragma unroll 2
        for (int j = 1; j <= TBL; ++j) {
          uint32_t w[NW];
          const uint4* src = reinterpret_cast<const uint4*>(crow(j));
#pragma unroll
          for (int i = 0; i < NW / 4; ++i) {
            const uint4 v4 = src[i];
            w[4 * i] = v4.x; w[4 * i + 1] = v4.y; w[4 * i + 2] = v4.z; w[4 * i + 3] = v4.w;
          }
          U<CU, 1, 3> rym, ryp, rt;
#pragma unroll
          for (int i = 0; i < N; ++i) { rym.v[i] = w[i]; ryp.v[i] = w[N + i]; rt.v[i] = w[2 * N + i]; }
          const uint64_t m = __builtin_amdgcn_uicmp(d, (uint32_t)j, 32 /* ICMP_EQ */);
          u_cmov_ct(sym, m, rym);
          u_cmov_ct(syp, m, ryp);
          u_cmov_ct(st, m, rt);
        }
        ued_add_niels_ct<CU>(s, q, sym, syp, st, neg);
      } else {
        ued_row_load<CU>(c, row(d));
        ued_add<CU, FUSED>(s, q, c, neg);  // the comb that follows needs T
      }
      q.x = s.x; q.y = s.y; q.z = s.z;
      if constexpr (FUSED) q.t = s.t;
    }
    if constexpr (FUSED) ued_comb_accumulate<CU>(q.x, q.y, q.z, q.t, base_scalars + idx * 32, utable);
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, q.z);
      flags[idx] = rejected ? 2 : 0;
    }
  }
}

// ---- reference-mirroring variable-base ladder on the unsaturated field ----------------------
// Same algorithm, operation for operation, as k_scalarmul_var (kernels.hpp): the reference's
// scalar_mul_fixed_window_{am3,a0} (src/curve/projective.rs:871-896 / :842-867) -- table d*P,
// d = 0..15, then per nibble (high first) 4 doublings, lookup, 1 complete addition -- with the
// Renes-Costello-Batina formulas add_different_{am3,a0} (:340-423 / :268-338) and double_{am3,a0}
// (:586-646 / :544-583) in homogeneous coordinates.  Only the limb representation differs, so the
// un-normalised (X : Y : Z) it ends with are the reference's, residue for residue; the result is
// handed to the saturated tail (store_result) as canonical Montgomery limbs.
template <class CU>
struct UPt {
  U<CU, 1, 3> x, y, z;
};

template <class CU>
ECCX_DEV void upt_add(UPt<CU>& r, const UPt<CU>& p, const UPt<CU>& q) {
  auto t0 = u_mul(p.x, q.x);
  auto t1 = u_mul(p.y, q.y);
  auto t2 = u_mul(p.z, q.z);
  auto t3 = u_sub(u_mul(u_add(p.x, p.y), u_add(q.x, q.y)), u_add(t0, t1));
  auto t4 = u_sub(u_mul(u_add(p.y, p.z), u_add(q.y, q.z)), u_add(t1, t2));
  auto y3 = u_sub(u_mul(u_add(p.x, p.z), u_add(q.x, q.z)), u_add(t0, t2));
  if constexpr (CU::Sat::A0) {
    auto t0x3 = u_add(u_add(t0, t0), t0);
    auto t2b = u_mul_k<CU>(t2, CU::CB3);
    auto z3 = u_add(t1, t2b);
    auto t1b = u_sub(t1, t2b);
    auto y3b = u_mul_k<CU>(y3, CU::CB3);
    auto x3 = u_sub(u_mul(t3, t1b), u_mul(t4, y3b));
    auto y3c = u_add(u_mul(t1b, z3), u_mul(y3b, t0x3));
    auto z3b = u_add(u_mul(z3, t4), u_mul(t0x3, t3));
    r.x = u_fit<1, 3>(x3); r.y = u_fit<1, 3>(y3c); r.z = u_fit<1, 3>(z3b);
  } else {
    auto x3a = u_sub(y3, u_mul_k<CU>(t2, CU::CB));
    auto x3b = u_add(u_add(x3a, x3a), x3a);
    auto z3 = u_sub(t1, x3b);
    auto x3 = u_add(t1, x3b);
    auto t2x3 = u_add(u_add(t2, t2), t2);
    auto y3a = u_sub(u_sub(u_mul_k<CU>(y3, CU::CB), t2x3), t0);
    auto y3b = u_add(u_add(y3a, y3a), y3a);
    auto t0b = u_sub(u_add(u_add(t0, t0), t0), t2x3);
    auto x3f = u_sub(u_mul(t3, x3), u_mul(t4, y3b));
    auto y3f = u_add(u_mul(x3, z3), u_mul(t0b, y3b));
    auto z3f = u_add(u_mul(t4, z3), u_mul(t3, t0b));
    r.x = u_fit<1, 3>(x3f); r.y = u_fit<1, 3>(y3f); r.z = u_fit<1, 3>(z3f);
  }
}

template <class CU>
ECCX_DEV void upt_dbl(UPt<CU>& r, const UPt<CU>& p) {
  if constexpr (CU::Sat::A0) {
    auto t0 = u_sqr(p.y);
    auto z2 = u_add(t0, t0);
    auto z4 = u_add(z2, z2);
    auto z8 = u_add(z4, z4);
    auto t1 = u_mul(p.y, p.z);
    auto t2 = u_mul_k<CU>(u_sqr(p.z), CU::CB3);
    auto x3a = u_mul(t2, z8);
    auto y3a = u_add(t0, t2);
    auto z3 = u_mul(t1, z8);
    auto t2x3 = u_add(u_add(t2, t2), t2);
    auto t0b = u_sub(t0, t2x3);
    auto y3 = u_add(x3a, u_mul(t0b, y3a));
    auto x3m = u_mul(t0b, u_mul(p.x, p.y));
    r.x = u_fit<1, 3>(u_add(x3m, x3m)); r.y = u_fit<1, 3>(y3); r.z = u_fit<1, 3>(z3);
  } else {
    auto t0 = u_sqr(p.x);
    auto t1 = u_sqr(p.y);
    auto t2 = u_sqr(p.z);
    auto xy = u_mul(p.x, p.y);
    auto t3 = u_add(xy, xy);
    auto xz = u_mul(p.x, p.z);
    auto z3a = u_add(xz, xz);
    auto y3a = u_sub(u_mul_k<CU>(t2, CU::CB), z3a);
    auto y3b = u_add(u_add(y3a, y3a), y3a);
    auto x3a = u_sub(t1, y3b);
    auto y3c = u_add(t1, y3b);
    auto y3d = u_mul(x3a, y3c);
    auto x3b = u_mul(x3a, t3);
    auto t2x3 = u_add(u_add(t2, t2), t2);
    auto z3b = u_sub(u_sub(u_mul_k<CU>(z3a, CU::CB), t2x3), t0);
    auto z3c = u_add(u_add(z3b, z3b), z3b);
    auto t0b = u_sub(u_add(u_add(t0, t0), t0), t2x3);
    auto y3 = u_add(y3d, u_mul(t0b, z3c));
    auto yz = u_mul(p.y, p.z);
    auto t0c = u_add(yz, yz);
    auto x3 = u_sub(x3b, u_mul(t0c, z3c));
    auto z3d = u_mul(t0c, t1);
    auto z3e = u_add(z3d, z3d);
    r.x = u_fit<1, 3>(x3); r.y = u_fit<1, 3>(y3); r.z = u_fit<1, 3>(u_add(z3e, z3e));
  }
}

template <class CU>
ECCX_DEV void upt_store(uint32_t* __restrict__ row, const UPt<CU>& p) { u3_store<CU>(row, p.x, p.y, p.z); }
template <class CU>
ECCX_DEV void upt_load(UPt<CU>& p, const uint32_t* __restrict__ row) { u3_load<CU>(p.x, p.y, p.z, row); }

template <class CU>
__global__ void __launch_bounds__(WG, unsat_occupancy<CU>()) k_scalarmul_var_mirror_unsat(
    size_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points, uint8_t* __restrict__ out,
    uint8_t* __restrict__ flags, uint8_t* __restrict__ proj, uint32_t* __restrict__ scratch, uint32_t opts) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int SB = CS::SB;
  constexpr int W = urow3_words<CU>();
  // [workgroup][entry 0..15][thread 0..255][W words]
  uint32_t* slab = scratch + ((size_t)blockIdx.x * 16 * WG + threadIdx.x) * (size_t)W;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * W; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    if (opts & OPT_ONLY_MARKED) {
      // behind the secret-scalar affine-table ladder: only the units it marked (from the base point alone)
      active = active && flags[idx] == FLAG_REDO;
      if (__builtin_amdgcn_ballot_w64(active) == 0) continue;
    }
    U<CU, 1, 3> one, zero;
    u_set_zero(zero);
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    UPt<CU> q;
    bool rejected = false;
    if (opts & OPT_BASE_IS_GENERATOR) {
#pragma unroll
      for (int i = 0; i < CU::N; ++i) { q.x.v[i] = CU::GX[i]; q.y.v[i] = CU::GY[i]; }
    } else {
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
    q.z = one;
    {
      UPt<CU> inf;  // (0 : 1 : 0), projective.rs:152-156
      inf.x = zero; inf.y = one; inf.z = zero;
      upt_store<CU>(row(0), inf);
      upt_store<CU>(row(1), q);
    }
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
    // steps 0..13 build table[2..15]; then 2*SB windows of (4 doublings + 1 addition)
    constexpr int NW = 2 * SB;
    constexpr int NSTEPS = 14 + 5 * NW;
    int win = 0, sub = 0;
    for (int s = 0; s < NSTEPS; ++s) {
      const bool building = s < 14;
      bool do_dbl;
      if (building) {
        do_dbl = (s == 0);
      } else {
        if (s == 14) { q.x = zero; q.y = one; q.z = zero; }
        do_dbl = sub < 4;
      }
      if (do_dbl) {
        UPt<CU> t;
        upt_dbl<CU>(t, q);
        q = t;
      } else {
        uint32_t e = 1;
        if (!building) {
          const uint32_t byte = k[win >> 1];
          e = (win & 1) ? (byte & 0x0f) : (byte >> 4);  // high nibble first (projective.rs:885)
        }
        UPt<CU> sel, t;
        if (!building && (opts & OPT_CT_SCAN)) {
          // select_from_table (projective.rs:427-434): every row of the lane's table is read and
          // the wanted one kept by select -- no address depends on the digit
          upt_load<CU>(sel, row(0));
          for (uint32_t j = 1; j < 16; ++j) {
            UPt<CU> c;
            upt_load<CU>(c, row(j));
            const bool take = (j == e);
            u_select(sel.x, take, c.x, sel.x);
            u_select(sel.y, take, c.y, sel.y);
            u_select(sel.z, take, c.z, sel.z);
          }
        } else {
          upt_load<CU>(sel, row(e));
        }
        upt_add<CU>(t, q, sel);
        q = t;
      }
      if (building) {
        upt_store<CU>(row(s + 2), q);
      } else {
        if (++sub == 5) { sub = 0; ++win; }
      }
    }
    if (active) {
      // the saturated tail expects canonical Montgomery limbs
      Pt<CS> res;
      u_to_sat_mont<CU>(res.x, q.x);
      u_to_sat_mont<CU>(res.y, q.y);
      u_to_sat_mont<CU>(res.z, q.z);
      store_result<CS>(idx, res, rejected, out, flags, proj, opts);
    }
  }
}

// ---- edwards25519 fixed base with the comb table in LDS (BASELINE.json configs[2]) -------------
// The widest table the 160 KiB of LDS can hold: signed 6-bit windows, 43 windows x 33 entries
// (digit 0 = the neutral element, digits 1..32; a negative digit swaps y - x with y + x and negates
// 2d*x*y) x 28 words = 155.2 KiB, one 1024-thread workgroup per CU, lanes read their entry with
// ds_read_b128.  43 seven-product additions per unit against 16 for the default path, whose
// 134 MB table cannot live in LDS: this variant exists to measure exactly that trade
// (ECCX_TABLE_IN_LDS).
constexpr int ED_LDS_BLOCK = 1024;
constexpr int ED_LDS_BITS = 6;
constexpr int ED_LDS_WINDOWS = (256 + 1 + ED_LDS_BITS - 1) / ED_LDS_BITS;   // 43
constexpr int ED_LDS_DIGITS = (1 << (ED_LDS_BITS - 1)) + 1;                  // 0..32
constexpr int ED_LDS_ENTRY_WORDS = 28;
constexpr size_t ED_LDS_BYTES = (size_t)ED_LDS_WINDOWS * ED_LDS_DIGITS * ED_LDS_ENTRY_WORDS * sizeof(uint32_t);
static_assert(ED_LDS_BYTES <= 160 * 1024, "the table must fit the LDS of one CU");

template <class CU>
__global__ void __launch_bounds__(ED_LDS_BLOCK) k_ed_scalarmul_base_lds6(size_t n, const uint8_t* __restrict__ scalars,
                                                                         const uint32_t* __restrict__ table,
                                                                         uint32_t* __restrict__ rows_out,
                                                                         uint8_t* __restrict__ flags) {
  constexpr int N = CU::N;
  constexpr int EW = ED_LDS_ENTRY_WORDS;
  static_assert(3 * N <= EW, "entry layout");
  using T = U<CU, 1, 3>;
  extern __shared__ uint4 lds4[];
  {
    const uint4* src = reinterpret_cast<const uint4*>(table);
    for (int i = threadIdx.x; i < ED_LDS_WINDOWS * ED_LDS_DIGITS * EW / 4; i += ED_LDS_BLOCK) lds4[i] = src[i];
  }
  __syncthreads();
  for (size_t base = (size_t)blockIdx.x * ED_LDS_BLOCK; base < n; base += (size_t)gridDim.x * ED_LDS_BLOCK) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * 32;
    T qx, qy, qz, qt;  // the neutral element (0, 1, 1, 0)
    u_set_zero(qx); u_set_zero(qy); u_set_zero(qz); u_set_zero(qt);
    qy.v[0] = 1; qz.v[0] = 1;
    for (int w = 0; w < ED_LDS_WINDOWS; ++w) {
      // signed (Booth) digit of window w of the big-endian scalar: bits 6w-1 .. 6w+5
      const int pos = ED_LDS_BITS * w - 1 + 8;
      const int bi = pos >> 3;
      const uint32_t b0 = (bi >= 1 && bi <= 32) ? k[32 - bi] : 0u;
      const uint32_t b1 = (bi + 1 <= 32) ? k[32 - bi - 1] : 0u;
      const uint32_t w7 = ((b0 | (b1 << 8)) >> (pos & 7)) & 0x7fu;
      const uint32_t sgn = ~((w7 >> ED_LDS_BITS) - 1u);
      const uint32_t m = (((1u << (ED_LDS_BITS + 1)) - w7 - 1u) & sgn) | (w7 & ~sgn);
      const uint32_t d = (m >> 1) + (m & 1u);
      const bool neg = (sgn & 1u) != 0;
      const uint4* e = lds4 + (size_t)(w * ED_LDS_DIGITS + d) * (EW / 4);
      uint32_t ew[EW];
#pragma unroll
      for (int i = 0; i < EW / 4; ++i) {
        const uint4 v = e[i];
        ew[4 * i] = v.x; ew[4 * i + 1] = v.y; ew[4 * i + 2] = v.z; ew[4 * i + 3] = v.w;
      }
      T a, b, t2d, ym, yp;
#pragma unroll
      for (int i = 0; i < N; ++i) { a.v[i] = ew[i]; b.v[i] = ew[N + i]; t2d.v[i] = ew[2 * N + i]; }
      u_select(ym, neg, b, a);
      u_select(yp, neg, a, b);
      U<CU, 2, 4> t2;
      u_select(t2, neg, u_neg(t2d), u_as<2, 4>(t2d));
      ued_add_niels<CU, 2, 4>(qx, qy, qz, qt, ym, yp, t2);
    }
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), qx, qy, qz);
      flags[idx] = 0;
    }
  }
}

// ---- group law on affine inputs (eccx_point_add) -----------------------------------------------
//   impl Add / Sub / Neg for Point, CurveGroup::double   src/curve/fiat/curve_macros.rs:297-411,
//   src/curve/group.rs:28-70 -> the complete RCB addition (projective.rs:340-423 / :268-338), which
//   also covers a == b, a == -b and the point at infinity; edwards25519: Point::add
//   (curve25519.rs:695-710).  Same formulas as the saturated k_point_add / k_ed_point_add
//   (kernels.hpp, kept as the ECCX_MIRROR_REFERENCE variant), on the unsaturated field; rows of
//   (X, Y, Z) for k_batch_to_affine_unsat<NORM_HOMOGENEOUS / NORM_EDWARDS>.
template <class CU>
__global__ void __launch_bounds__(WG) k_point_add_unsat(size_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ a_inf,
                                                        const uint8_t* __restrict__ b, const uint8_t* __restrict__ b_inf,
                                                        uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags, uint32_t opts) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    U<CU, 1, 3> one;
#pragma unroll
    for (int k = 0; k < CU::N; ++k) one.v[k] = CU::ONE[k];
    auto load = [&](UPt<CU>& p, const uint8_t* __restrict__ src, bool inf, bool negate) {
      Fe<L> raw;
      fe_load_be<CS>(raw, src);
      p.x = u_as<1, 3>(u_to_mont<CU>(raw));
      fe_load_be<CS>(raw, src + FB);
      p.y = u_as<1, 3>(u_to_mont<CU>(raw));
      if (negate) p.y = u_reduce(u_neg(p.y));
      p.z = one;
      if (inf) {  // (0 : 1 : 0), projective.rs:152-156
        u_set_zero(p.x);
        p.y = one;
        u_set_zero(p.z);
      }
    };
    UPt<CU> p, q, r;
    load(p, a + i * (size_t)(2 * FB), a_inf && a_inf[i] == 1, false);
    load(q, b + i * (size_t)(2 * FB), b_inf && b_inf[i] == 1, (opts & OPT_NEGATE_B) != 0);
    upt_add<CU>(r, p, q);
    u3_store<CU>(rows_out + i * (size_t)urow3_words<CU>(), r.x, r.y, r.z);
    flags[i] = ((a_inf && a_inf[i] == 2) || (b_inf && b_inf[i] == 2)) ? 2 : 0;  // rejected operands stay rejected
  }
}

template <class CU>
__global__ void __launch_bounds__(WG) k_ed_point_add_unsat(size_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ a_fl,
                                                           const uint8_t* __restrict__ b, const uint8_t* __restrict__ b_fl,
                                                           uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags, uint32_t opts) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    auto load = [&](UEd<CU>& p, const uint8_t* __restrict__ src) {  // from_affine, curve25519.rs:638-645
      Fe<L> raw;
      fe_load_le<CS>(raw, src);
      p.x = u_reduce(u_as<1, 3>(u_from_sat<CU>(raw)));
      fe_load_le<CS>(raw, src + 32);
      p.y = u_reduce(u_as<1, 3>(u_from_sat<CU>(raw)));
      u_set_zero(p.z);
      p.z.v[0] = 1;
      p.t = u_fit<1, 3>(u_mul(p.x, p.y));
    };
    UEd<CU> p, q, r;
    load(p, a + i * 64);
    load(q, b + i * 64);
    UEdCached<CU> c;
    ued_cache<CU>(c, q);
    ued_add<CU, false>(r, p, c, (opts & OPT_NEGATE_B) != 0);  // -(x, y) = (-x, y): the cached form swaps y -+ x
    u3_store<CU>(rows_out + i * (size_t)urow3_words<CU>(), r.x, r.y, r.z);
    flags[i] = ((a_fl && a_fl[i] == 2) || (b_fl && b_fl[i] == 2)) ? 2 : 0;
  }
}

// ---- batched normalisation on the unsaturated field ----------------------------------------------
// Point::to_affine (projective.rs:655-682; curve25519.rs:663-666, :529-532) for a whole batch of
// rows written by the kernels above (X, Y, Z as tight digits of the working form).  Thread t of a
// workgroup handles units tile + u*WG + t, u = 0..U-1, with ONE inversion (Montgomery's trick),
// by division steps (inv_gcd.hpp).  Modes as for k_batch_to_affine (kernels_fast.hpp):
//   NORM_HOMOGENEOUS  x = X/Z, y = Y/Z, big-endian bytes; Z = 0 is infinity (rows of the group law)
//   NORM_JACOBIAN     x = X/Z^2, y = Y/Z^3, big-endian bytes; Z = 0 is infinity
//   NORM_EDWARDS      x = X/Z, y = Y/Z, little-endian bytes, flag 1 = neutral element
//   NORM_MONTGOMERY_U u = X/Z with 0 for Z = 0 (mod p), 32 little-endian bytes, flag 1 = zero result
//   NORM_JACOBIAN_X   x = X/Z^2 only, FB big-endian bytes per unit (to_affine_x_ct, src/curve/projective.rs:690: what
//                     ECDSA verification consumes, src/protocol/ecdsa.rs:383); Z = 0 is infinity
// flags[i] on entry: 2 marks a rejected input (kept, zero output).
template <class CU, int MODE, int UN>
__global__ void __launch_bounds__(WG) k_batch_to_affine_unsat(size_t n, const uint32_t* __restrict__ rows,
                                                              uint8_t* __restrict__ out, uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int W3 = urow3_words<CU>();
  using T = U<CU, 1, 3>;
  static_assert(MODE == NORM_HOMOGENEOUS || MODE == NORM_JACOBIAN || MODE == NORM_EDWARDS || MODE == NORM_MONTGOMERY_U ||
                    MODE == NORM_JACOBIAN_X, "mode");
  const size_t tile_units = (size_t)WG * UN;
  for (size_t tile = (size_t)blockIdx.x * tile_units; tile < n; tile += (size_t)gridDim.x * tile_units) {
    T one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    // Z = 0 means any multiple of p: the kernels encode infinity as all-zero limbs, but a unit
    // fed a point that is not on the curve (rejected under ECCX_VALIDATE_POINTS, or garbage
    // without it) can arrive with Z = p or 2p -- e.g. an Edwards "point" with x = +-y doubles to
    // Z = 0 -- and must not zero the shared inverse of the units normalised beside it
    // (decided without a data-dependent branch: the presence test and every substitution below are selects,
    // as z_inverse_ct / to_affine_ct are in the reference, src/curve/projective.rs:655-682)
    auto z_present = [&](const T& z) { return !u_is_zero_mod_p_ct(u_reduce(z)); };
    T pre[UN];  // prefix products of the (substituted) Z values
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      T x, y, z = one;
      if (i < n) {
        u3_load<CU>(x, y, z, rows + i * (size_t)W3);
        u_select(z, z_present(z), z, one);  // z_inverse_ct substitutes 1 (projective.rs:655-659)
      }
      if (u == 0) pre[0] = z;
      else pre[u] = u_fit<1, 3>(u_mul(pre[u - 1], z));
    }
    T inv;
    {
      Fe<L> c;
      u_to_canonical<CU>(c, pre[UN - 1]);
      fe_inv_gcd<CS>(c, c);
      inv = u_as<1, 3>(u_to_mont<CU>(c));
    }
#pragma unroll
    for (int u = UN - 1; u >= 0; --u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      T x = one, y = one, z = one;
      bool present = false;
      if (i < n) {
        u3_load<CU>(x, y, z, rows + i * (size_t)W3);
        present = z_present(z);
        u_select(z, present, z, one);
      }
      T zi;
      if (u > 0) {
        zi = u_fit<1, 3>(u_mul(inv, pre[u - 1]));
        inv = u_fit<1, 3>(u_mul(inv, z));
      } else {
        zi = inv;
      }
      Fe<L> ax, ay;
      if constexpr (MODE == NORM_JACOBIAN) {
        auto zi2 = u_sqr(zi);
        u_to_canonical<CU>(ax, u_mul(x, zi2));
        u_to_canonical<CU>(ay, u_mul(y, u_mul(zi2, zi)));
      } else if constexpr (MODE == NORM_JACOBIAN_X) {
        u_to_canonical<CU>(ax, u_mul(x, u_sqr(zi)));
      } else {
        u_to_canonical<CU>(ax, u_mul(x, zi));
        if constexpr (MODE == NORM_EDWARDS || MODE == NORM_HOMOGENEOUS) u_to_canonical<CU>(ay, u_mul(y, zi));
      }
      if (i < n) {
        const uint8_t fl = flags[i];
        const bool rejected = fl == 2;
        if constexpr (MODE == NORM_MONTGOMERY_U) {
#pragma unroll
          for (int k = 0; k < L; ++k) ax.v[k] = present ? ax.v[k] : 0u;
          fe_store_le<CS>(out + i * (size_t)FB, ax);
          flags[i] = fe_is_zero<CS>(ax) ? 1 : 0;
        } else if constexpr (MODE == NORM_EDWARDS) {
          bool neutral = fe_is_zero<CS>(ax);
#pragma unroll
          for (int k = 0; k < L; ++k) neutral = neutral & (ay.v[k] == (k == 0 ? 1u : 0u));
#pragma unroll
          for (int k = 0; k < L; ++k) { ax.v[k] = rejected ? 0u : ax.v[k]; ay.v[k] = rejected ? 0u : ay.v[k]; }
          fe_store_le<CS>(out + i * (size_t)(2 * FB), ax);
          fe_store_le<CS>(out + i * (size_t)(2 * FB) + FB, ay);
          flags[i] = rejected ? 2 : (neutral ? 1 : 0);
        } else if constexpr (MODE == NORM_JACOBIAN_X) {
          const bool ok = present & !rejected;
#pragma unroll
          for (int k = 0; k < L; ++k) ax.v[k] = ok ? ax.v[k] : 0u;
          fe_store_be<CS>(out + i * (size_t)FB, ax);
          flags[i] = rejected ? 2 : (present ? 0 : 1);
        } else if (fl != FLAG_REDO) {  // a unit marked from its BASE POINT is redone by the kernel launched behind
          const bool ok = present & !rejected;
#pragma unroll
          for (int k = 0; k < L; ++k) { ax.v[k] = ok ? ax.v[k] : 0u; ay.v[k] = ok ? ay.v[k] : 0u; }
          fe_store_be<CS>(out + i * (size_t)(2 * FB), ax);
          fe_store_be<CS>(out + i * (size_t)(2 * FB) + FB, ay);
          flags[i] = rejected ? 2 : (present ? 0 : 1);
        }
      }
    }
  }
}

}  // namespace eccx
