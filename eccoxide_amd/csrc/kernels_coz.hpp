// Variable base over an AFFINE window table (all Weierstrass curves of the engine).
//
// The generic ladder (k_scalarmul_var_unsat) adds Jacobian table entries: 11 products + 3 squares
// per addition (Z^2, Z^3 of the entry cached).  Here the 16 entries d P are brought to a common
// denominator first, so that every ladder addition is a MIXED addition (8 products + 3 squares) and the
// table itself is built with co-Z additions (4 products + 2 squares):
//   a = 0 (BLS12-381 G1)   the ladder runs on the isomorphic curve E': y^2 = x^3 + b zeta^6 on which
//                          the rescaled entries are affine; Z *= zeta at the end.  No inversion.
//   a = -3 (P-256/384/521) the doubling uses a, so the ladder stays on the curve: the common
//                          denominator zeta = Z_16 is INVERTED once per unit (division steps,
//                          inv_gcd.hpp: about 60 products' worth of issue slots) and folded into the
//                          rescaling pass, which then yields true affine entries.
// k_scalarmul_coz_unsat is launched in front of k_scalarmul_var_unsat, which redoes the (rare) units
// this kernel marks as degenerate.
#pragma once
#include <type_traits>
#include "inv_gcd.hpp"
#include "kernels_ct.hpp"

namespace eccx {

// the GLV template argument of curves without an endomorphism
struct NoGlv {
  static constexpr int K_BITS = 0;
};
// NoGlv for the secret-scalar ladder over bases the caller vouches to have PRIME order (ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP on
// a curve with a cofactor): see coz_full_windows()
struct PrimeOrderBases : NoGlv {};

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

// ---- variable base over an affine window table ----------------------------------------------------
// The table is built with co-Z additions (uzaddu below): T_{d+1} = P + T_d with both operands over the same
// denominator costs 4 products + 2 squares (+ 1 product where the denominator itself is tracked), leaves P
// over the new denominator Z_{d+1} = Z_d (X_P - X_T) for the next step, and the ratio Z_{d+1} / Z_d falls
// out for free.
//   table build   2P from the affine P (2 products + 4 squares; 4 x y^2 and 8 y^4 ARE P over the denominator
//                 2 y of 2P) + 14 co-Z additions; rows (X_d, Y_d, ratio), each over the denominator of its step
//   common Z      backward pass, per entry l *= ratio, l^2, l^3, X l^2, Y l^3 (4 products + 1 square);
//                 l starts at 1 (a = 0: entries affine on E', zeta = Z_16 kept for the end) or at
//                 1 / Z_16 (a = -3: entries affine on the curve itself)
//   ladder        per signed 5-bit window 5 doublings + 1 mixed addition (2 with GLV)
// GLV = true is the ECCX_ASSUME_SUBGROUP form (BLS12-381 bases in G1): k = k1 + k2 x^2, the second
// half added from (beta x', -y') -- sigma commutes with the isomorphism.
// A base point of order <= 16 makes the build degenerate (some T_d is the point at infinity or +-P,
// a ratio is zero): such units are only MARKED here (flag FLAG_REDO) and recomputed by the generic
// ladder, launched behind this kernel with OPT_ONLY_MARKED -- no point of a prime-order curve and no
// valid G1 point takes that path.

template <class CU>
constexpr int urowc_words() { return ((3 * CU::N + 3) / 4) * 4; }

// the first two columns of a table row (the ladder without GLV reads nothing else)
template <class CU>
ECCX_DEV void u2_load(U<CU, 1, 3>& x, U<CU, 1, 3>& y, const uint32_t* __restrict__ row) {
  constexpr int N = CU::N;
  constexpr int W = ((2 * N + 3) / 4) * 4;
  uint32_t w[W];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    const uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { x.v[i] = w[i]; y.v[i] = w[N + i]; }
}

// the two coordinates of a table entry as a row of their own (the secret-scalar ladder's compact table)
template <class CU>
constexpr int urow2_words() { return ((2 * CU::N + 3) / 4) * 4; }
#ifndef ECCX_CT_VAR_PK
#define ECCX_CT_VAR_PK 0  // 1: the secret-scalar ladders scan their table two words per instruction (kernels_ct.hpp ct_fsel2) --
                          // measured 0.5-1.7 % SLOWER than v_cndmask here (the rows come from HBM, not LDS; the pairs cost registers): off
#endif
// TAGGED: every word | CT_F32_TAG, the form the two-word select reads (limbs stay below 3 * 2^28: see ct_scan_lds_pk)
template <class CU, bool TAGGED = false>
ECCX_DEV void u2_store(uint32_t* __restrict__ row, const U<CU, 1, 3>& x, const U<CU, 1, 3>& y) {
  constexpr int N = CU::N;
  constexpr int W = urow2_words<CU>();
  constexpr uint32_t T = TAGGED ? CT_F32_TAG : 0u;
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) w[i] = (i < N ? x.v[i] : (i < 2 * N ? y.v[i - N] : 0u)) | T;
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// r = p + (x2, +-y2, 1)
template <class CU>
ECCX_DEV void ujac_madd_signed(UJac<CU>& r, bool& h_zero, bool& r_zero, const UJac<CU>& p,
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

// Co-Z addition with update (Goundar-Joye-Miyaji ZADDU): p = (x1, y1) and t = (x2, y2) over the SAME
// denominator Z.  t <- p + t and p <- p, both over the new denominator Z d with d = x1 - x2 (exported:
// the ratio of the denominators).  4 products + 2 squares; Z itself is not touched (the caller multiplies
// it by d where it needs the denominator).
template <class CU>
ECCX_DEV void uzaddu(U<CU, 1, 3>& x1, U<CU, 1, 3>& y1, U<CU, 1, 3>& x2, U<CU, 1, 3>& y2, U<CU, 1, 3>& d_out, bool& d_zero) {
  const auto d = u_reduce(u_sub(x1, x2));
  d_zero = u_is_zero_mod_p(d);
  const auto c = u_sqr(d);
  const U<CU, 1, 3> w1 = u_fit<1, 3>(u_mul(x1, c));
  const auto w2 = u_mul(x2, c);
  const auto e = u_reduce(u_sub(y1, y2));
  const auto dd = u_sqr(e);
  const U<CU, 1, 3> a1 = u_fit<1, 3>(u_mul(y1, u_sub(w1, w2)));
  const auto x3 = u_reduce(u_sub(u_sub(dd, w1), w2));
  const auto y3m = u_mul(e, u_sub(w1, x3));
  x2 = x3;
  y2 = u_reduce(u_sub(y3m, a1));
  x1 = w1;
  y1 = a1;
  d_out = d;
}

// r = p + (x2, y2, 1) with y2 already signed; h_zero / r_zero without early exits (secret-scalar form)
template <class CU>
ECCX_DEV void ujac_madd_ct(UJac<CU>& r, bool& h_zero, bool& r_zero, const UJac<CU>& p, const U<CU, 1, 3>& x2,
                           const U<CU, 1, 3>& y2) {
  auto z1z1 = u_sqr(p.z);
  auto u2 = u_mul(x2, z1z1);
  auto t = u_mul(p.z, z1z1);
  auto s2 = u_mul(y2, t);
  auto h = u_reduce(u_sub(u2, p.x));
  auto rr = u_reduce(u_sub(s2, p.y));
  h_zero = u_is_zero_mod_p_ct(h);
  r_zero = u_is_zero_mod_p_ct(rr);
  auto hh = u_sqr(h);
  auto hhh = u_mul(h, hh);
  auto v = u_mul(p.x, hh);
  auto r2 = u_sqr(rr);
  auto x3 = u_reduce(u_sub(u_sub(u_sub(r2, hhh), v), v));
  r.x = x3;
  if constexpr (CU::KIND == UK_MONT) {
    r.y = u_fit<1, 3>(u_mul_add(rr, u_sub(v, x3), u_neg(p.y), hhh));
  } else if constexpr (UB<CU>::SPARSE) {
    r.y = u_mul_sub(rr, u_sub(v, x3), p.y, hhh);
  } else {
    auto y3a = u_mul(rr, u_sub(v, x3));
    auto y1h = u_mul(p.y, hhh);
    r.y = u_reduce(u_sub(y3a, y1h));
  }
  r.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(p.z, h));
}

// 2 (x, y) from AFFINE coordinates, Jacobian result with Z3 = 2 y (2 products + 4 squares).  As a by-product
// S = 4 x y^2 and 8 y^4 are (x, y) itself over the denominator 2 y -- what the co-Z table build starts from.
template <class CU>
ECCX_DEV void ujac_dbl_affine(U<CU, 1, 3>& x3, U<CU, 1, 3>& y3, U<CU, 1, 3>& z3, U<CU, 1, 3>& sv, U<CU, 1, 3>& y8v,
                              const U<CU, 1, 3>& px, const U<CU, 1, 3>& py) {
  const auto xx = u_sqr(px);
  const auto yy = u_sqr(py);
  const auto yyyy = u_sqr(yy);
  const auto xyy = u_mul(px, yy);
  const auto s2 = u_add(xyy, xyy);
  sv = u_reduce(u_add(s2, s2));
  const auto y4 = u_add(yyyy, yyyy);
  const auto y8 = u_add(y4, y4);
  y8v = u_reduce(u_add(y8, y8));
  if constexpr (CU::Sat::A0 == 1) {
    const auto m = u_reduce(u_add(u_add(xx, xx), xx));  // 3 x^2
    x3 = u_reduce(u_sub(u_sub(u_sqr(m), sv), sv));
    y3 = u_reduce(u_sub(u_mul(m, u_sub(sv, x3)), y8v));
  } else {
    U<CU, 1, 3> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    const auto xm = u_sub(xx, one);                      // a = -3: 3 (x^2 - 1)
    const auto m = u_reduce(u_add(u_add(xm, xm), xm));
    x3 = u_reduce(u_sub(u_sub(u_sqr(m), sv), sv));
    y3 = u_reduce(u_sub(u_mul(m, u_sub(sv, x3)), y8v));
  }
  z3 = u_reduce(u_add(py, py));
}

// Secret scalars (CT = true, ECCX_CT_SCAN).  Same ladder; what changes:
//   lookup      every row 1 .. 2^(WB-1) of the lane's table is read and the digit's row kept by v_cndmask
//               (select_from_table, src/curve/projective.rs:427-434): no address depends on a digit.  The
//               table is private to the lane (HBM slab), so the scan multiplies its traffic by the number
//               of rows: the secret-scalar form uses narrower windows (WB = 4: 8 rows, 65 additions for 256
//               bits) than the default (WB = 5: 16 rows, 52 additions).
//   branches    none on scalar-derived data: the accumulator-at-infinity patch runs unconditionally, and
//               accumulator == +-entry is resolved by selects -- -entry: infinity; entry: 2 * entry computed
//               from the entry's affine coordinates (ujac_dbl_affine).  Before window w (5 doublings done)
//               the accumulator is 2^WB S P with |S| <= k / 2^(WB (w + 1)) + 0.6 and the entry is d P with
//               |d| <= 2^(WB-1): on a point of prime order n, 2^WB S = +-d (mod n) needs
//               2^(WB w) <= 2^(8 SB) / (n - 35), i.e. w <= (8 SB - NBITS + 1) / WB.  Only those bottom windows
//               (coz_full_windows(): 1, P-521: 2-3) compute the doubling; curves with a cofactor
//               (BLS12-381: points of order 33, 121, ... exist) compute it in every window.
//   what still depends on data: the BASE POINT (public in every protocol the reference implements):
//               rejected / degenerate units are flagged from the point alone and redone by the
//               reference-mirroring scan kernel.
// ASSUME_PRIME: the caller vouches that the bases lie in the subgroup of prime order NBITS bits long (ECCX_CT_SCAN |
// ECCX_ASSUME_SUBGROUP on bls12_381_g1: a hash-to-curve output, a deserialised and checked point) -- the bound of the
// prime-order curves then holds there too.
template <class CU, int WB, bool ASSUME_PRIME = false>
constexpr int coz_full_windows() {
  using CS = typename CU::Sat;
  constexpr int NWIN = (8 * CS::SB + 1 + WB - 1) / WB;
  return (CS::PRIME_ORDER || ASSUME_PRIME) ? (8 * CS::SB - CS::NBITS + 1) / WB + 1 : NWIN;
}
#ifndef ECCX_CT_VAR_BITS
#define ECCX_CT_VAR_BITS 4
#endif

// Waves per SIMD the kernel is compiled for: the mixed additions keep two coordinates of an entry live where
// the generic ladder keeps five, so the 14-limb fields fit three waves at the price of spilling (448 B of scratch
// per lane against 76-80 at two waves).  Same-box A/B with the counters beside the time (round 3,
// profiles/r03_occ14_ab.json):
//   P-384      3 waves 18.94 ms, 4.55 GB written   2 waves 19.71 ms (+4.1 %), 3.42 GB     -> three
//   BLS12-381  3 waves 34.31 ms, 12.82 GB written  2 waves 34.57 ms (+0.7 %), 6.53 GB     -> two: the 6.3 GB are the
//              spill frame rewritten ~13 times per unit (the resident frames, 88 MB, do not fit the 32 MB of L2), and the
//              higher clock the two-wave build holds (2.30 against 2.25 GHz) returns what its 3 % more cycles per
//              instruction cost.  The GLV form, whose rows carry a third column, was 0.7 % slower at three already.
// 9 limbs: three equal to four, five +5.5 %; 18 limbs: three +16 %.
#ifndef ECCX_COZ_OCC_U14
#define ECCX_COZ_OCC_U14 3
#endif
#ifndef ECCX_COZ_OCC_BLS
#define ECCX_COZ_OCC_BLS 2
#endif
#ifndef ECCX_COZ_OCC_U9
#define ECCX_COZ_OCC_U9 4
#endif
template <class CU, bool GLV>
constexpr int coz_occupancy() {
  return CU::N <= 9 ? ECCX_COZ_OCC_U9
                    : (CU::N <= 14 ? (GLV ? ECCX_OCC_U14 : (CU::KIND == UK_MONT ? ECCX_COZ_OCC_BLS : ECCX_COZ_OCC_U14)) : ECCX_OCC_U18);
}

// scratch: [workgroup][row 0..16][thread][urowc_words]; row 0: split scalar (words 0..15, GLV) and
// zeta (words 16..16+N); rows 1..16: the table.
// FUSED: the verify shape u1*G + u2*Q (u1*G - u2*Q with OPT_NEGATE_B; src/protocol/ecdsa.rs:215) in one pass, as in
// k_scalarmul_var_unsat<CU, true>: the ladder computes u2*(+-Q), the 16-bit comb of u1*G is accumulated onto it.
template <class CU, class G, bool GLV, bool FUSED = false, int WB = 5, bool CT = false>
__global__ void __launch_bounds__(WG, (coz_occupancy<CU, GLV>())) k_scalarmul_coz_unsat(size_t n, const uint8_t* __restrict__ scalars,
                                                                                 const uint8_t* __restrict__ points,
                                                                                 uint32_t* __restrict__ rows_out,
                                                                                 uint8_t* __restrict__ flags,
                                                                                 uint32_t* __restrict__ scratch, uint32_t opts,
                                                                                 const uint8_t* __restrict__ base_scalars = nullptr,
                                                                                 const uint32_t* __restrict__ utable = nullptr) {
  static_assert(!(GLV && FUSED), "the verify shape takes any curve point");
  static_assert(!(CT && FUSED), "the verify shape has no secret-scalar form");
  constexpr bool ASSUME_PRIME = std::is_same<G, PrimeOrderBases>::value;
  static_assert(!ASSUME_PRIME || CT, "PrimeOrderBases only narrows the secret-scalar form's collision windows");
  static_assert(WB >= 3 && WB <= 5, "table rows 1 .. 2^(WB-1) must fit the slab");
  constexpr int TBL = 1 << (WB - 1);  // table rows 1 .. TBL
  using CS = typename CU::Sat;
  constexpr bool ISO = CS::A0 == 1;  // a = 0: isomorphic curve, no inversion
  constexpr int L = CS::L;
  constexpr int N = CU::N;
  constexpr int FB = CS::FB;
  constexpr int SB = CS::SB;
  static_assert(!GLV || SB == 32, "the split is written for 256-bit scalars");
  constexpr int NWIN = GLV ? (G::K_BITS + 1 + WB - 1) / WB : (8 * SB + 1 + WB - 1) / WB;
  constexpr int WR = urowc_words<CU>();
  static_assert(!ISO || WR >= 16 + N, "row 0 holds the split scalar and zeta");
  using T = U<CU, 1, 3>;
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)WR;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * WR; };
  // Secret-scalar form: the finished table (x, y per entry) is written a second time as rows of 2 N words in the part of
  // the slab the build does not use (rows TBL + 1 .. 16): every lookup reads ALL of it, so its rows are packed -- a
  // wavefront's reads of one entry are then one contiguous run (the 3-column build rows leave a quarter of every
  // 128-byte line unused: 67 GB per 2^20 P-256 units against 44 GB).
  constexpr int WC = urow2_words<CU>();
  static_assert(!CT || (size_t)TBL * WC <= (size_t)(FAST_TABLE_ROWS - 1 - TBL) * WR, "the compact table must fit the free rows");
  uint32_t* cslab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS + (TBL + 1)) * WG * (size_t)WR + (size_t)threadIdx.x * WC;
  auto crow = [&](uint32_t e) { return cslab + (size_t)(e - 1) * WG * WC; };
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
    const uint32_t* kw = row(0);  // written above through another pointer: no restrict

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
    if constexpr (FUSED) {
      if (opts & OPT_NEGATE_B) q.y = u_reduce(u_neg(q.y));  // u1*G - u2*Q
    }
    q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
    u3_store<CU>(row(1), q.x, q.y, one);

    // Booth digit (|digit| in 0..16, sign) of window w: from the scalar bytes, or from half `second`
    // of the split scalar (little-endian words; bit -1 is 0)
    auto booth = [&](int w, bool second, uint32_t& d, bool& neg) {
      uint32_t w6;
      if constexpr (GLV) {
        const uint32_t* h = kw + (second ? 8 : 0);
        const int pos = WB * w - 1 + 32;  // one zero word below the scalar
        const int wi = pos >> 5, sh = pos & 31;
        const uint32_t lo = (wi >= 1 && wi <= 5) ? h[wi - 1] : 0u;
        const uint32_t hi = (wi <= 4) ? h[wi] : 0u;
        w6 = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & ((1u << (WB + 1)) - 1u);
        const uint32_t s = ~((w6 >> WB) - 1u);
        const uint32_t m = (((1u << (WB + 1)) - w6 - 1u) & s) | (w6 & ~s);
        d = (m >> 1) + (m & 1u);
        neg = (s & 1u) != 0;
      } else {
        booth_digit<WB, SB>(k, w, d, neg);
      }
    };

    // ---- table build: rows d = 1..16 hold (X_d, Y_d, Z_d / Z_{d-1}), each over the denominator of its step ----
    bool degenerate = false;
    {
      T px = q.x, py = q.y;  // P, following the table's denominator
      T tx, ty, ratio;       // T_d
      T zacc = one;          // Z_d (a = -3 only: a = 0 gets zeta as the product of the ratios)
      {
        // 2P from the affine P, co-Z with P for free: S = 4 x y^2 and 8 y^4 are P's coordinates over Z_2 = 2 y
        T x3, y3, sv, y8v;
        ujac_dbl_affine<CU>(x3, y3, ratio, sv, y8v, px, py);
        degenerate = u_is_zero_mod_p(ratio);
        tx = x3;
        ty = y3;
        px = sv;
        py = y8v;
        if constexpr (!ISO) zacc = ratio;
        u3_store<CU>(row(2), tx, ty, ratio);
      }
      for (int d = 3; d <= TBL; ++d) {
        bool dz;
        uzaddu<CU>(px, py, tx, ty, ratio, dz);  // T_d = P + T_{d-1}
        degenerate = degenerate || dz;
        if constexpr (!ISO) zacc = u_fit<1, 3>(u_mul(zacc, ratio));
        u3_store<CU>(row(d), tx, ty, ratio);
      }
      // ---- common denominator zeta = Z_TBL ----
      T lam = one, next = ratio;
      if constexpr (ISO) {
        if constexpr (CT) u2_store<CU, (ECCX_CT_VAR_PK != 0)>(crow(TBL), tx, ty);  // the top entry is affine on E' as it stands
        else if constexpr (GLV) u3_store<CU>(row(TBL), tx, ty, u_fit<1, 3>(u_mul_k<CU>(tx, CU::BETA)));
      } else {
        // 1 / zeta (zeta = 0 only for degenerate units, which are redone anyway: the inverse of 0 is 0)
        Fe<L> c;
        u_to_canonical<CU>(c, zacc);
        fe_inv_gcd<CS>(c, c);
        lam = u_as<1, 3>(u_to_mont<CU>(c));
        const T l2 = u_fit<1, 3>(u_sqr(lam));
        const T l3 = u_fit<1, 3>(u_mul(l2, lam));
        const T xt = u_fit<1, 3>(u_mul(tx, l2)), yt = u_fit<1, 3>(u_mul(ty, l3));
        if constexpr (CT) u2_store<CU, (ECCX_CT_VAR_PK != 0)>(crow(TBL), xt, yt);
        else u3_store<CU>(row(TBL), xt, yt, one);
      }
      for (int d = TBL - 1; d >= 1; --d) {
        lam = u_fit<1, 3>(u_mul(lam, next));
        T x, y;
        u3_load<CU>(x, y, next, row(d));
        const T l2 = u_fit<1, 3>(u_sqr(lam));
        const T l3 = u_fit<1, 3>(u_mul(l2, lam));
        const T xs = u_fit<1, 3>(u_mul(x, l2));
        const T ys = u_fit<1, 3>(u_mul(y, l3));
        if constexpr (CT) u2_store<CU, (ECCX_CT_VAR_PK != 0)>(crow(d), xs, ys);
        else if constexpr (GLV) u3_store<CU>(row(d), xs, ys, u_fit<1, 3>(u_mul_k<CU>(xs, CU::BETA)));
        else u3_store<CU>(row(d), xs, ys, one);
      }
      if constexpr (ISO) {  // zeta = Z_16 / Z_1: the product of all ratios
        uint32_t* zdst = row(0) + 16;
#pragma unroll
        for (int i = 0; i < N; ++i) zdst[i] = lam.v[i];
      }
    }

    // ---- ladder: per window WB doublings and one mixed addition (two with GLV) ----
    if constexpr (CT) {
      // secret scalars: a fixed schedule, every table row read at every lookup, selects only.
      // With GLV (bases of prime order by contract): two additions per window, the second of [x^2]P = -sigma(P) =
      // (beta x, -y) computed from the looked-up entry (one product); accumulator == +-entry is possible in the
      // bottom window only (the partial sums S1 + S2 x^2 are integers below r until the last window is in:
      // tests/ct_model.py glv_events, exhaustive on small curves of the family).
      constexpr int FULLW = GLV ? 1 : coz_full_windows<CU, WB, ASSUME_PRIME>();
      constexpr int HALVES = GLV ? 2 : 1;
      auto lookup = [&](int w, bool second, T& ex, T& ey, uint32_t& d) {
        bool neg;
        booth(w, second, d, neg);
        neg = neg != second;  // `second` is the same for every lane
#if ECCX_CT_VAR_PK
        {
          // two words per select: the rows hold tagged limbs (exact under v_pk_fma_f32 against 1.0 / 0.0, kernels_ct.hpp)
          uint64_t acc[N];  // 2N words
#pragma unroll
          for (int p = 0; p < N; ++p) acc[p] = 0;
#pragma unroll 2
          for (int j = 1; j <= TBL; ++j) {
            uint32_t m;
            asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(m) : "s"(__builtin_amdgcn_uicmp(d, (uint32_t)j, 32 /* ICMP_EQ */)));
            const uint64_t m2 = m;
            const uint4* src = reinterpret_cast<const uint4*>(crow(j));
            uint4 v[WC / 4];
#pragma unroll
            for (int c = 0; c < WC / 4; ++c) v[c] = src[c];
#pragma unroll
            for (int p = 0; p < N; ++p) {
              const uint4 qv = v[p / 2];
              ct_fsel2(acc[p], (p & 1) ? (((uint64_t)qv.w << 32) | qv.z) : (((uint64_t)qv.y << 32) | qv.x), m2);
            }
          }
#pragma unroll
          for (int i = 0; i < N; ++i) {
            ex.v[i] = (uint32_t)(acc[i / 2] >> (32 * (i & 1))) & ~CT_F32_TAG;
            ey.v[i] = (uint32_t)(acc[(N + i) / 2] >> (32 * ((N + i) & 1))) & ~CT_F32_TAG;
          }
        }
#else
        u_set_zero(ex);
        u_set_zero(ey);
#pragma unroll 2
        for (int j = 1; j <= TBL; ++j) {
          T cx, cy;
          u2_load<CU>(cx, cy, crow(j));
          const uint64_t m = __builtin_amdgcn_uicmp(d, (uint32_t)j, 32 /* ICMP_EQ */);
          u_cmov_ct(ex, m, cx);
          u_cmov_ct(ey, m, cy);
        }
#endif
        if constexpr (GLV) {
          if (second) ex = u_fit<1, 3>(u_mul_k<CU>(ex, CU::BETA));
        }
        U<CU, 2, 4> sy;
        u_select_ct(sy, neg, u_neg(ey), u_as<2, 4>(ey));
        ey = u_reduce(sy);
      };
      T zero;
      u_set_zero(zero);
      using TZ = U<CU, UJac<CU>::ZK, UJac<CU>::ZV>;
      {
        // the top window has nothing to add to: the accumulator starts as its signed entry (Z = 1), or at
        // infinity for digit 0
        T ex, ey;
        uint32_t d;
        lookup(NWIN - 1, false, ex, ey, d);
        q.x = ex;
        q.y = ey;
        u_select_ct(q.z, d == 0, u_as<UJac<CU>::ZK, UJac<CU>::ZV>(zero), u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one));
      }
#pragma unroll 1
      for (int win = NWIN - 1; win >= 0; --win) {
        if (win != NWIN - 1) {
#pragma unroll 1
          for (int j = 0; j < WB; ++j) {
            UJac<CU> t;
            ujac_dbl<CU>(t, q);
            q = t;
          }
        }
#pragma unroll 1
        for (int half = (win == NWIN - 1) ? 1 : 0; half < HALVES; ++half) {
          T ex, ey;
          uint32_t d;
          lookup(win, half != 0, ex, ey, d);
          const bool q_inf = u_limbs_all_zero(q.z);
          const bool e_skip = d == 0;
          UJac<CU> sum;
          bool hz, rz;
          ujac_madd_ct<CU>(sum, hz, rz, q, ex, ey);
          if (win < FULLW) {  // loop counter: the same for every lane and every scalar
            T dx, dy, dz, sv, y8v;
            ujac_dbl_affine<CU>(dx, dy, dz, sv, y8v, ex, ey);
            const bool same_x = hz & !q_inf & !e_skip;
            const uint64_t mt = ct_mask(same_x & rz);    // accumulator == entry
            const uint64_t mc = ct_mask(same_x & !rz);   // accumulator == -entry
            u_cmov_ct(sum.x, mt, dx);
            u_cmov_ct(sum.y, mt, dy);
            u_cmov_ct(sum.z, mt, TZ(u_as<UJac<CU>::ZK, UJac<CU>::ZV>(dz)));
            u_cmov_ct(sum.z, mc, TZ(u_as<UJac<CU>::ZK, UJac<CU>::ZV>(zero)));
          }
          const uint64_t mi = ct_mask(q_inf), mk = ct_mask(!e_skip);
          u_cmov_ct(sum.x, mi, ex);  // accumulator at infinity: the sum is the entry
          u_cmov_ct(sum.y, mi, ey);
          u_cmov_ct(sum.z, mi, TZ(u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one)));
          u_cmov_ct(q.x, mk, sum.x);
          u_cmov_ct(q.y, mk, sum.y);
          u_cmov_ct(q.z, mk, sum.z);
        }
      }
    } else {
      constexpr int LAST_SUB = GLV ? WB + 1 : WB;  // sub 0..WB-1: doublings, WB: addition (WB+1: second half's addition)
      int win, sub;
      {
        // the top window's (first) addition has nothing to add to: the accumulator starts as that signed
        // entry -- affine, so Z = 1 -- or at infinity for digit 0
        uint32_t d;
        bool neg;
        booth(NWIN - 1, false, d, neg);
        T tx, ty;
        u2_load<CU>(tx, ty, row(d ? d : 1));
        U<CU, 2, 4> sy;
        u_select(sy, neg, u_neg(ty), u_as<2, 4>(ty));
        q.x = tx;
        q.y = u_reduce(sy);
        q.z = u_as<UJac<CU>::ZK, UJac<CU>::ZV>(one);
        if (d == 0) u_set_zero(q.z);
        if constexpr (GLV) {
          win = NWIN - 1;
          sub = WB + 1;  // the second half's addition of the top window comes next
        } else {
          win = NWIN - 2;
          sub = 0;
        }
      }
      bool fix_pending = false, fix_lane = false;
      while (win >= 0) {
        bool step_done;
        if (fix_pending || sub < WB) {
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
          const bool second = GLV && sub == WB + 1;
          uint32_t d;
          bool neg;
          booth(win, second, d, neg);
          if (second) neg = !neg;  // [x^2]P = -sigma(P) = (beta x, -y)
          T ex, ey;
          if constexpr (GLV) {
            T eb;
            u3_load<CU>(ex, ey, eb, row(d ? d : 1));
            if (second) ex = eb;
          } else {
            u2_load<CU>(ex, ey, row(d ? d : 1));
          }
          const bool q_inf = u_limbs_all_zero(q.z);
          const bool e_skip = (d == 0);
          UJac<CU> sum;
          bool hz, rz;
          ujac_madd_signed<CU>(sum, hz, rz, q, ex, ey, neg);
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
        if (step_done) {
          if (sub < LAST_SUB) {
            ++sub;
          } else {
            sub = 0;
            --win;
          }
        }
      }
    }
    // back from E' to the curve: Z *= zeta
    if constexpr (ISO) {
      T zeta;
      const uint32_t* zsrc = row(0) + 16;
#pragma unroll
      for (int i = 0; i < N; ++i) zeta.v[i] = zsrc[i];
      q.z = u_fit<UJac<CU>::ZK, UJac<CU>::ZV>(u_mul(q.z, zeta));
    }
    if constexpr (FUSED) ucomb_accumulate<CU, false>(q, base_scalars + idx * (size_t)SB, utable);
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), q.x, q.y, u_fit<1, 3>(q.z));
      flags[idx] = rejected ? 2 : (degenerate ? FLAG_REDO : 0);
    }
  }
}

}  // namespace eccx
