// Unsaturated field arithmetic for the fast kernels: N limbs of B < 32 bits in 32-bit
// registers.
//
// Why: on gfx950 a carry instruction costs as much as a multiply (profiles/r01_valu_rates.jsonl),
// so the saturated multiplier (fe.hpp) spends half its issue slots on v_addc_co_u32.  With
// 28/29-bit limbs a whole column of the product fits a 64-bit accumulator: the unit of work
// is ONE v_mad_u64_u32, squares really cost half the cross products, additions are N
// independent v_add_u32 and there is no conditional subtraction.
//
// Four kinds of field (C::KIND), all reproducing the reference's values modulo p only
// (canonical bytes are produced at the very end, u_to_canonical):
//   0  Montgomery, R = 2^(B*N), p = -1 mod 2^B: the Montgomery factor m is the low limb of
//      the accumulator itself and "+ m*p" becomes "drop that limb, + m*(p+1)"; p + 1 has few
//      non-zero digits for P-256 (4 of 9); for P-384 (12 of 14 digits but 4 signed powers of two)
//      it is added as signed shifted copies of m into signed columns (SPARSE_*)
//                                                     (src/curve/fiat/p256_64.rs, p384_64.rs)
//   1  Montgomery, general p (BLS12-381): m = acc * (-p^-1) mod 2^B   (src/curve/fiat/bls12_381_*.rs)
//   2  p = 2^k - 1 (P-521), plain representation: the wrapped half of the product is
//      accumulated into the same columns with weight 2^(B*N - k)          (src/curve/fiat/p521_64.rs)
//   3  p = 2^255 - 19, plain representation, 9 x 29 = 261 bits: the upper columns are carried
//      into digits first and enter the lower columns times 2^261 mod p = 1216
//                                                          (src/curve/fiat/curve25519_64.rs)
//
// The price is bookkeeping, done at compile time.  Every value carries two bounds in its
// type, U<C, K, V>:
//   K  every limb is < K * 2^B (+ a few units)   (K = 1: "tight"; after u_reduce the digits
//                                                are the unique radix-2^B digits)
//   V  the integer value is < V * p
// Each operation knows what it needs (UB<C>) and, where an operand is too loose, reduces it
// first (if constexpr) -- so a formula written once is valid for every field, and an unsafe
// composition cannot be expressed:
//   mul / sqr   N*K1*K2*2^(2B) (+ N*2^(2B) reduction products) + carry < 2^64; the Montgomery
//               result is < (V1*V2/(R/p) + 1) * p, required < 3p; tight output
//   add         K1+K2 <= KMAX (limbs below 2^32), V1+V2
//   sub         a + BIAS - b with BIAS = 4p spread so no limb borrows; b tight and < 3p;
//               returns K1+2, V1+4
//   reduce      one carry chain that also takes off q*p, q ~ value / p estimated from the top
//               limbs (shifted copies of q for the Solinas primes, 32-bit unsigned for kinds 2
//               and 3, signed 64-bit otherwise): returns exact tight digits of a value < 3p
//   merged products (one Montgomery reduction for two products; where a formula adds or subtracts
//   two products -- the Y3 of the point formulas -- the second product's reduction, its column
//   extractions and the carry chain of the difference go away):
//     mul_add        a*b + c*d, general Montgomery (BLS12-381), unsigned columns; K1K2 + K3K4 <= KKMAX;
//                    a subtraction is written (4p - c)*d
//     mul_sub        a*b - c*d + p and
//     mul_sub_2sqr   a*b - 2*c^2 + p on SIGNED columns (P-384, whose sparse reduction is signed anyway;
//                    BLS12-381): the subtracted product goes in with v_mad_i64_i32 against negated
//                    limbs, each side of a column within 63 bits (K1K2 <= UBS::KKS per product), p
//                    added above R so that the result is in (0, 3p).  P-256 (9 x 29 bits) and P-521
//                    have no room for either form.
#pragma once
#include "fe.hpp"

#ifndef ECCX_SOLINAS_REDUCE
#define ECCX_SOLINAS_REDUCE 1  // 0: A/B switch back to the generic signed 64-bit chain
#endif

namespace eccx {

#include "umad_chunks.inc"

// acc += a * k for a NEGATIVE constant k (signed columns of the sparse reduction)
__device__ __forceinline__ void smad1_k(uint64_t& acc, uint32_t a, int32_t k) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}

// acc += a * b, both signed 32-bit register operands
__device__ __forceinline__ void smad1_v(uint64_t& acc, int32_t a, int32_t b) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}

enum : int { UK_MONT_PP1 = 0, UK_MONT = 1, UK_MERSENNE = 2, UK_PM19 = 3 };

template <class C, int K, int V>
struct U {
  static_assert(K >= 1 && K < (1 << (32 - C::B)), "limb bound out of range (limbs must stay below 2^32)");
  static_assert(V >= 1 && V <= 4096, "value bound out of range");
  uint32_t v[C::N];
};

// what the multiplier columns and the 32-bit limbs can take
template <class C>
struct UB {
  static constexpr bool MONT = C::KIND == UK_MONT_PP1 || C::KIND == UK_MONT;
  static constexpr bool WRAPPED = C::KIND == UK_MERSENNE;  // both halves of the product share the columns
  static constexpr int KMAX = (1 << (32 - C::B)) - 1;
  static constexpr uint64_t COL = (uint64_t)1 << (2 * C::B);  // one product of tight limbs
  // largest K1*K2 a product may have: Montgomery columns hold N products + N reduction
  // products; the Mersenne columns hold N direct products + N - 1 doubled wrapped ones
  // (kind 3: N products + one folded digit times 1216)
  // (sparse reduction: the columns are signed, one bit less)
  static constexpr bool SPARSE = C::KIND == UK_MONT_PP1 && C::SPARSE_N > 0;
  static constexpr int KKMAX = WRAPPED  ? (int)(~(uint64_t)0 / ((uint64_t)(2 * C::N - 1) * COL))
                               : SPARSE ? (int)((~(uint64_t)0 >> 1) / ((uint64_t)C::N * COL)) - 1
                                        : (int)(~(uint64_t)0 / ((uint64_t)C::N * COL)) - 1;
  static_assert(KKMAX >= 1, "limbs too wide for this many columns");
  static constexpr bool kk_ok(int k1, int k2) { return k1 * k2 <= KKMAX; }
  // squares also shift the operand left by one (two for the wrapped cross terms)
  static constexpr int KSQ_SHIFT = WRAPPED ? 2 : 1;
  static constexpr bool ksq_ok(int k) { return k * k <= KKMAX && (k << KSQ_SHIFT) <= KMAX + 1; }
  // bound of a product of values below v1*p and v2*p
  static constexpr int vout(int v1, int v2) {
    return MONT ? (int)(((uint32_t)(v1 * v2) + C::RP - 1) / C::RP) + 1 : (C::KIND == UK_PM19 ? 2 : 3);
  }
  // laziest limb bound a value may have and still be squared / multiplied by a tight value
  static constexpr int KLAZY = ksq_ok(2) ? 2 : 1;
};

// queue of pending single-instruction MACs (see MacQ in fe.hpp)
template <bool KC>
struct UMacQ {
  uint32_t x[9], y[9];
  int n = 0;
  ECCX_DEV void push(uint64_t& acc, uint32_t a, uint32_t b) {
    x[n] = a;
    y[n] = b;
    if (++n == 9) flush(acc);
  }
  ECCX_DEV void flush(uint64_t& acc) {
#define ECCX_Q(i) x[i], y[i]
    if constexpr (KC) {
      switch (n) {
        case 1: umad1_k(acc, ECCX_Q(0)); break;
        case 2: umad2_k(acc, ECCX_Q(0), ECCX_Q(1)); break;
        case 3: umad3_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2)); break;
        case 4: umad4_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3)); break;
        case 5: umad5_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4)); break;
        case 6: umad6_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5)); break;
        case 7: umad7_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6)); break;
        case 8: umad8_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7)); break;
        case 9: umad9_k(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7), ECCX_Q(8)); break;
        default: break;
      }
    } else {
      switch (n) {
        case 1: umad1_v(acc, ECCX_Q(0)); break;
        case 2: umad2_v(acc, ECCX_Q(0), ECCX_Q(1)); break;
        case 3: umad3_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2)); break;
        case 4: umad4_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3)); break;
        case 5: umad5_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4)); break;
        case 6: umad6_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5)); break;
        case 7: umad7_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6)); break;
        case 8: umad8_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7)); break;
        case 9: umad9_v(acc, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7), ECCX_Q(8)); break;
        default: break;
      }
    }
#undef ECCX_Q
    n = 0;
  }
};

// ---- product cores on raw limb arrays; bounds are checked by the typed wrappers -----------

// Montgomery product / square (kinds 0 and 1): product scanning, one 64-bit column at a time
template <class C, bool SQR, bool BCONST>
ECCX_DEV void u_mul_core_mont(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N]) {
  constexpr int N = C::N;
  constexpr bool PP1 = C::KIND == UK_MONT_PP1;
  // digits added per Montgomery factor: p + 1 (whose digit 0 is zero) or p
  auto PR = [](int j) constexpr { return PP1 ? C::PP1[j] : C::P[j]; };
  uint32_t m[N], t[N], a2[N];
  if constexpr (SQR) {
#pragma unroll
    for (int i = 0; i < N; ++i) a2[i] = a[i] << 1;
  }
  uint64_t acc = 0;
  UMacQ<BCONST> qa;
  UMacQ<true> qm;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    if constexpr (SQR) {
#pragma unroll
      for (int i = lo; 2 * i < k; ++i) qa.push(acc, a[i], a2[k - i]);
      if ((k & 1) == 0) qa.push(acc, a[k / 2], a[k / 2]);
    } else {
#pragma unroll
      for (int i = lo; i <= hi; ++i) qa.push(acc, a[i], b[k - i]);
    }
    qa.flush(acc);
    if constexpr (UB<C>::SPARSE) {
      // m*(p+1) as signed shifted copies of m: term t lands in column i + OFF[t]
#pragma unroll
      for (int t = 0; t < C::SPARSE_N; ++t) {
        const int i = k - C::SPARSE_OFF[t];
        if (i >= lo && i <= (k < N ? k - 1 : N - 1)) {
          if (C::SPARSE_SIGN[t] > 0) qm.push(acc, m[i], 1u << C::SPARSE_SHIFT[t]);
        }
      }
      qm.flush(acc);
#pragma unroll
      for (int t = 0; t < C::SPARSE_N; ++t) {
        const int i = k - C::SPARSE_OFF[t];
        if (i >= lo && i <= (k < N ? k - 1 : N - 1)) {
          if (C::SPARSE_SIGN[t] < 0) smad1_k(acc, m[i], -(int32_t)(1u << C::SPARSE_SHIFT[t]));
        }
      }
    } else {
#pragma unroll
      for (int i = lo; i <= (k < N ? k - 1 : N - 1); ++i) {
        if (PR(k - i) != 0) qm.push(acc, m[i], PR(k - i));
      }
      qm.flush(acc);
    }
    if (k < N) {
      if constexpr (PP1) {
        // -p^-1 mod 2^B = 1, so m[k] is the low limb itself, and m*p = m*(p+1) - m: the "- m"
        // clears that limb (the shift drops it), m*(p+1) goes to the columns above
        m[k] = (uint32_t)acc & C::MASK;
      } else {
        m[k] = ((uint32_t)acc * C::N0B) & C::MASK;
        umad1_k(acc, m[k], C::P[0]);
      }
    } else {
      t[k - N] = (uint32_t)acc & C::MASK;
    }
    if constexpr (UB<C>::SPARSE) acc = (uint64_t)((int64_t)acc >> C::B);  // columns may be negative
    else acc >>= C::B;
  }
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

// (a*b + c*d) / R with ONE reduction (general Montgomery, kind 1): both products are accumulated
// into the same columns before the Montgomery factor of a column is taken.  Where a formula adds
// two products -- Y3 = r*(v - x3) - s1*h^3 -- this saves the N*N reduction products and N column
// extractions of the second one; a subtraction is written as (4p - c)*d (u_neg).
template <class C>
ECCX_DEV void u_mul2_core_mont(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N],
                               const uint32_t (&c)[C::N], const uint32_t (&d)[C::N]) {
  constexpr int N = C::N;
  static_assert(C::KIND == UK_MONT, "written for the general Montgomery fields");
  uint32_t m[N], t[N];
  uint64_t acc = 0;
  UMacQ<false> qa;
  UMacQ<true> qm;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
#pragma unroll
    for (int i = lo; i <= hi; ++i) qa.push(acc, a[i], b[k - i]);
#pragma unroll
    for (int i = lo; i <= hi; ++i) qa.push(acc, c[i], d[k - i]);
    qa.flush(acc);
#pragma unroll
    for (int i = lo; i <= (k < N ? k - 1 : N - 1); ++i) {
      if (C::P[k - i] != 0) qm.push(acc, m[i], C::P[k - i]);
    }
    qm.flush(acc);
    if (k < N) {
      m[k] = ((uint32_t)acc * C::N0B) & C::MASK;
      umad1_k(acc, m[k], C::P[0]);
    } else {
      t[k - N] = (uint32_t)acc & C::MASK;
    }
    acc >>= C::B;
  }
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

// (a*b - c*d) / R + p, or (a*b - 2*c*c) / R + p with NEG_SQ, in ONE reduction, for the fields whose
// Montgomery reduction runs on signed columns already (SPARSE: P-384).  The subtracted product is
// accumulated with v_mad_i64_i32 against negated limbs (as a square: cross terms once, against the
// doubled operand), p comes in through its sparse form p = (p + 1) - 1 added to the columns above R, so
// the result is positive whatever the operands.  Both sides of a column stay inside 63 bits as long as
// each product's K1*K2 is within KKMAX (the bound already leaves room for the reduction's terms).
template <class C, bool NEG_SQ>
ECCX_DEV void u_mul_sub_core_sparse(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N],
                                    const uint32_t (&c)[C::N], const uint32_t (&d)[C::N]) {
  constexpr int N = C::N;
  static_assert(C::KIND == UK_MONT_PP1 && C::SPARSE_N > 0, "signed columns: the sparse Montgomery fields");
  uint32_t m[N], t[N];
  int32_t nc[N];  // -c (NEG_SQ: -2c)
#pragma unroll
  for (int i = 0; i < N; ++i) nc[i] = NEG_SQ ? -(int32_t)(c[i] << 1) : -(int32_t)c[i];
  // p = (p + 1) - 1 as what it adds to result digit j (column N + j)
  auto offset = [](int j) constexpr -> int64_t {
    int64_t v = (j == 0) ? -1 : 0;
    for (int tm = 0; tm < C::SPARSE_N; ++tm)
      if (C::SPARSE_OFF[tm] == j) v += (int64_t)C::SPARSE_SIGN[tm] * ((int64_t)1 << C::SPARSE_SHIFT[tm]);
    return v;
  };
  uint64_t acc = 0;
  UMacQ<false> qa;
  UMacQ<true> qm;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    if (k >= N && offset(k - N) != 0) acc = (uint64_t)((int64_t)acc + offset(k - N));
#pragma unroll
    for (int i = lo; i <= hi; ++i) qa.push(acc, a[i], b[k - i]);
    qa.flush(acc);
    if constexpr (NEG_SQ) {
      // -2 c^2: i < j terms once against 2c (nc = -2c, so nc_i * c_j * 2 = nc_i * (2 c_j)); diagonal nc_i * c_i
#pragma unroll
      for (int i = lo; 2 * i < k; ++i) smad1_v(acc, nc[i], (int32_t)(c[k - i] << 1));
      if ((k & 1) == 0) smad1_v(acc, nc[k / 2], (int32_t)c[k / 2]);
    } else {
#pragma unroll
      for (int i = lo; i <= hi; ++i) smad1_v(acc, nc[i], (int32_t)d[k - i]);
    }
    // m*(p+1) as signed shifted copies of m (as u_mul_core_mont)
#pragma unroll
    for (int tm = 0; tm < C::SPARSE_N; ++tm) {
      const int i = k - C::SPARSE_OFF[tm];
      if (i >= lo && i <= (k < N ? k - 1 : N - 1)) {
        if (C::SPARSE_SIGN[tm] > 0) qm.push(acc, m[i], 1u << C::SPARSE_SHIFT[tm]);
      }
    }
    qm.flush(acc);
#pragma unroll
    for (int tm = 0; tm < C::SPARSE_N; ++tm) {
      const int i = k - C::SPARSE_OFF[tm];
      if (i >= lo && i <= (k < N ? k - 1 : N - 1)) {
        if (C::SPARSE_SIGN[tm] < 0) smad1_k(acc, m[i], -(int32_t)(1u << C::SPARSE_SHIFT[tm]));
      }
    }
    if (k < N) m[k] = (uint32_t)acc & C::MASK;
    else t[k - N] = (uint32_t)acc & C::MASK;
    acc = (uint64_t)((int64_t)acc >> C::B);
  }
  if (offset(N - 1) != 0) acc = (uint64_t)((int64_t)acc + offset(N - 1));
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

// The same for the general Montgomery fields (BLS12-381), whose columns are unsigned in the plain
// product: here they are read as signed -- 63 bits, K1*K2 <= KKS on each side -- and p enters as its
// digits added to the columns above R.
template <class C, bool NEG_SQ>
ECCX_DEV void u_mul_sub_core_mont(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N],
                                  const uint32_t (&c)[C::N], const uint32_t (&d)[C::N]) {
  constexpr int N = C::N;
  static_assert(C::KIND == UK_MONT, "general Montgomery fields");
  uint32_t m[N], t[N];
  int32_t nc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) nc[i] = NEG_SQ ? -(int32_t)(c[i] << 1) : -(int32_t)c[i];
  uint64_t acc = 0;
  UMacQ<false> qa;
  UMacQ<true> qm;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    if (k >= N) acc += (uint64_t)C::P[k - N];  // + p*R
#pragma unroll
    for (int i = lo; i <= hi; ++i) qa.push(acc, a[i], b[k - i]);
    qa.flush(acc);
    if constexpr (NEG_SQ) {
#pragma unroll
      for (int i = lo; 2 * i < k; ++i) smad1_v(acc, nc[i], (int32_t)(c[k - i] << 1));
      if ((k & 1) == 0) smad1_v(acc, nc[k / 2], (int32_t)c[k / 2]);
    } else {
#pragma unroll
      for (int i = lo; i <= hi; ++i) smad1_v(acc, nc[i], (int32_t)d[k - i]);
    }
#pragma unroll
    for (int i = lo; i <= (k < N ? k - 1 : N - 1); ++i) {
      if (C::P[k - i] != 0) qm.push(acc, m[i], C::P[k - i]);
    }
    qm.flush(acc);
    if (k < N) {
      m[k] = ((uint32_t)acc * C::N0B) & C::MASK;
      umad1_k(acc, m[k], C::P[0]);
    } else {
      t[k - N] = (uint32_t)acc & C::MASK;
    }
    acc = (uint64_t)((int64_t)acc >> C::B);
  }
  acc += (uint64_t)C::P[N - 1];
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

// Mersenne product / square (kind 2), p = 2^k - 1 with B*N - k = S: the product's upper half
// has weight 2^(B*N) = 2^S (mod p), so a_i * b_j with i + j >= N is accumulated into column
// i + j - N times 2^S.  Inputs tight; output tight (digit 1 may exceed 2^B by the last carry).
template <class C, bool SQR, bool BCONST>
ECCX_DEV void u_mul_core_mers(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N]) {
  constexpr int N = C::N;
  constexpr int S = C::B * N - C::PBITS;
  static_assert(S >= 1 && S <= 2, "wrap factor must keep shifted limbs below 2^32");
  uint32_t t[N], bw[N], a2[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (SQR) {
      a2[i] = a[i] << 1;  // also serves the wrapped cross terms: 2 (cross) * 2^S (wrap) = a2_i * a2_j for S = 1
      bw[i] = (S == 1) ? 0u : a[i] << (S + 1);
    } else {
      a2[i] = 0;
      bw[i] = b[i] << S;
    }
  }
  uint64_t acc = 0;
  UMacQ<BCONST && !SQR> qa;
  UMacQ<false> qw;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if constexpr (SQR) {
      // direct terms i + j = k, wrapped terms i + j = k + N; i < j doubled, i == j once
#pragma unroll
      for (int i = 0; 2 * i < k; ++i) qa.push(acc, a[i], a2[k - i]);
      if ((k & 1) == 0) qa.push(acc, a[k / 2], a[k / 2]);
      qa.flush(acc);
#pragma unroll
      for (int i = k + 1; 2 * i < k + N; ++i) {
        if constexpr (S == 1) qw.push(acc, a2[i], a2[k + N - i]);
        else qw.push(acc, a[i], bw[k + N - i]);
      }
      if (((k + N) & 1) == 0) {
        if constexpr (S == 1) qw.push(acc, a[(k + N) / 2], a2[(k + N) / 2]);
        else qw.push(acc, a2[(k + N) / 2], a2[(k + N) / 2]);
      }
      qw.flush(acc);
    } else {
#pragma unroll
      for (int i = 0; i <= k; ++i) qa.push(acc, a[i], b[k - i]);
      qa.flush(acc);
#pragma unroll
      for (int i = k + 1; i < N; ++i) qw.push(acc, a[i], bw[k + N - i]);
      qw.flush(acc);
    }
    t[k] = (uint32_t)acc & C::MASK;
    acc >>= C::B;
  }
  // what is left has weight 2^(B*N) = 2^S again
  acc = (acc << S) + t[0];
  t[0] = (uint32_t)acc & C::MASK;
  t[1] += (uint32_t)(acc >> C::B);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

// 2^255 - 19 (kind 3): tail shared by the product and the small multiple.  t holds the lower
// digits, c what the column chain left above them (weight 2^(B*N) = FOLD); afterwards the bits
// of the top limb above 2^255 are folded too (times 19), so the value is below 2^255 + 2^46.
template <class C>
ECCX_DEV void u_pm_tail(uint32_t (&t)[C::N], uint64_t c) {
  constexpr int N = C::N;
  static_assert(C::FOLD == (19u << (C::B * N - 255)), "fold factor");
  uint64_t acc = c * C::FOLD + t[0];
  t[0] = (uint32_t)acc & C::MASK;
  t[1] += (uint32_t)(acc >> C::B);
  const uint32_t q = t[N - 1] >> C::TOPSHIFT;
  t[N - 1] &= (1u << C::TOPSHIFT) - 1u;
  t[0] += 19u * q;
}

template <class C, bool SQR, bool BCONST>
ECCX_DEV void u_mul_core_pm(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N]) {
  constexpr int N = C::N;
  uint32_t h[N], t[N], a2[N];
  if constexpr (SQR) {
#pragma unroll
    for (int i = 0; i < N; ++i) a2[i] = a[i] << 1;
  }
  UMacQ<BCONST> qa;
  // upper columns k = N .. 2N-2 as a number of their own: digits h[0 .. N-1]
  uint64_t acc = 0;
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
    if constexpr (SQR) {
#pragma unroll
      for (int i = k - N + 1; 2 * i < k; ++i) qa.push(acc, a[i], a2[k - i]);
      if ((k & 1) == 0) qa.push(acc, a[k / 2], a[k / 2]);
    } else {
#pragma unroll
      for (int i = k - N + 1; i < N; ++i) qa.push(acc, a[i], b[k - i]);
    }
    qa.flush(acc);
    h[k - N] = (uint32_t)acc & C::MASK;
    acc >>= C::B;
  }
  h[N - 1] = (uint32_t)acc;
  // lower columns, each taking its upper digit times 2^(B*N) mod p
  acc = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if constexpr (SQR) {
#pragma unroll
      for (int i = 0; 2 * i < k; ++i) qa.push(acc, a[i], a2[k - i]);
      if ((k & 1) == 0) qa.push(acc, a[k / 2], a[k / 2]);
    } else {
#pragma unroll
      for (int i = 0; i <= k; ++i) qa.push(acc, a[i], b[k - i]);
    }
    qa.flush(acc);
    umad1_k(acc, h[k], C::FOLD);
    t[k] = (uint32_t)acc & C::MASK;
    acc >>= C::B;
  }
  u_pm_tail<C>(t, acc);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

template <class C, bool SQR, bool BCONST>
ECCX_DEV void u_mul_core(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N]) {
  if constexpr (C::KIND == UK_MERSENNE) u_mul_core_mers<C, SQR, BCONST>(r, a, b);
  else if constexpr (C::KIND == UK_PM19) u_mul_core_pm<C, SQR, BCONST>(r, a, b);
  else u_mul_core_mont<C, SQR, BCONST>(r, a, b);
}

// ---- weak reduction -----------------------------------------------------------------------
// Propagate carries and take off q*p, q ~ value / p estimated from the two top limbs (never
// too large, at most one or two too small), in one signed chain.  Result: the exact radix-2^B
// digits of a value in [0, 3p).
template <class C, int K1, int V1>
ECCX_DEV U<C, 1, 3> u_reduce(const U<C, K1, V1>& a) {
  constexpr int N = C::N;
  const uint32_t top = a.v[N - 1] + (a.v[N - 2] >> C::B);
  uint32_t q;
  if constexpr (C::QMUL != 0) q = __umulhi(top, C::QMUL);
  else q = top >> C::TOPSHIFT;
  U<C, 1, 3> r;
  if constexpr (C::KIND == UK_MERSENNE || C::KIND == UK_PM19) {
    // q * p = q * 2^PBITS - q * (2^PBITS - p): q (or 19 q) enters at the bottom, q << TOPSHIFT
    // leaves at the top; every partial sum is non-negative and below 2^32, so the chain is
    // plain 32-bit arithmetic
    uint32_t c = (C::KIND == UK_PM19) ? 19u * q : q;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      const uint32_t t = a.v[i] + c;
      r.v[i] = t & C::MASK;
      c = t >> C::B;
    }
    r.v[N - 1] = a.v[N - 1] + c - (q << C::TOPSHIFT);
  } else if constexpr (ECCX_SOLINAS_REDUCE && C::KIND == UK_MONT_PP1 && C::SOL_N > 0 && K1 < UB<C>::KMAX && V1 < 64) {
    // Solinas prime: -q*p = q - q*(p+1), and p + 1 is a handful of powers of two, so q*p comes
    // off as shifted copies of q -- no multiplications, 32-bit arithmetic.  Limbs that lose a
    // copy are biased by 2^B, the bias being returned one limb higher (sum of the biases = 0),
    // so no partial sum is negative; K1 < KMAX keeps "limb + 2^B + copies of q" below 2^32.
    uint32_t c = q;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      uint32_t t = a.v[i] + c;
      if (i == C::SOL_BIAS_FROM) t += C::MASK + 1u;
      if (i > C::SOL_BIAS_FROM) t += C::MASK;
#pragma unroll
      for (int j = 0; j < C::SOL_N; ++j) {
        if (C::SOL_LIMB[j] == i) {
          if (C::SOL_SIGN[j] > 0) t -= q << C::SOL_SHIFT[j];
          else t += q << C::SOL_SHIFT[j];
        }
      }
      r.v[i] = t & C::MASK;
      c = t >> C::B;
    }
    r.v[N - 1] = a.v[N - 1] + c - (q << C::TOPSHIFT) - (C::SOL_BIAS_FROM < N - 1 ? 1u : 0u);
  } else {
    const int32_t nq = -(int32_t)q;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      {  // acc += a[i]: one v_mad_u64_u32 by 1 instead of an add-with-carry pair
        uint64_t u = (uint64_t)acc;
        umad1_k(u, a.v[i], 1u);
        acc = (int64_t)u;
      }
      if (C::P[i] != 0) acc += (int64_t)nq * (int64_t)(int32_t)C::P[i];
      if (i < N - 1) {
        r.v[i] = (uint32_t)acc & C::MASK;
        acc >>= C::B;  // arithmetic shift: a negative partial sum borrows from the next limb
      } else {
        r.v[i] = (uint32_t)acc;
      }
    }
  }
  return r;
}

// widen the type of a value without touching it (e.g. a tight value used where lazy is allowed)
template <int K2, int V2, class C, int K1, int V1>
ECCX_DEV U<C, K2, V2> u_as(const U<C, K1, V1>& a) {
  static_assert(K2 >= K1 && V2 >= V1, "u_as may only loosen bounds");
  U<C, K2, V2> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = a.v[i];
  return r;
}

// bring a value into the bounds (K2, V2) >= (1, 3): as is if it fits, reduced otherwise
template <int K2, int V2, class C, int K1, int V1>
ECCX_DEV U<C, K2, V2> u_fit(const U<C, K1, V1>& a) {
  static_assert(K2 >= 1 && V2 >= 3, "target must admit a reduced value");
  if constexpr (K1 <= K2 && V1 <= V2) return u_as<K2, V2>(a);
  else return u_as<K2, V2>(u_reduce(a));
}

// ---- typed operations (reduce an operand first where its bounds require it) ----------------
template <class C, int K1, int V1, int K2, int V2>
ECCX_DEV auto u_mul(const U<C, K1, V1>& a, const U<C, K2, V2>& b) {
  if constexpr (!UB<C>::kk_ok(K1, K2) || (UB<C>::vout(V1, V2) > 3)) {
    if constexpr (K1 > K2 || (K1 == K2 && V1 >= V2)) return u_mul(u_reduce(a), b);
    else return u_mul(a, u_reduce(b));
  } else {
    U<C, 1, UB<C>::vout(V1, V2)> r;
    u_mul_core<C, false, false>(r.v, a.v, b.v);
    return r;
  }
}

// a*b + c*d with one reduction (general Montgomery fields only).  The caller brings operands within
// the column budget: N*(K1*K2 + K3*K4)*2^(2B) + the reduction's N*2^(2B) must fit 64 bits.
template <class C, int K1, int V1, int K2, int V2, int K3, int V3, int K4, int V4>
ECCX_DEV auto u_mul_add(const U<C, K1, V1>& a, const U<C, K2, V2>& b, const U<C, K3, V3>& c, const U<C, K4, V4>& d) {
  static_assert(C::KIND == UK_MONT, "merged products are implemented for the general Montgomery fields");
  static_assert(K1 * K2 + K3 * K4 <= UB<C>::KKMAX, "two products overflow the columns: reduce an operand first");
  constexpr int VO = (int)(((uint32_t)(V1 * V2 + V3 * V4) + C::RP - 1) / C::RP) + 1;
  static_assert(VO <= 3, "result too large: reduce an operand first");
  U<C, 1, VO> r;
  u_mul2_core_mont<C>(r.v, a.v, b.v, c.v, d.v);
  return r;
}

// Signed-column budget of the merged differences: K1*K2 a product may have on either side of a column
// (63 bits; one product's worth is left for the reduction's terms)
template <class C>
struct UBS {
  static constexpr bool OK = UB<C>::SPARSE || C::KIND == UK_MONT;
  static constexpr int KKS = (int)((~(uint64_t)0 >> 1) / ((uint64_t)C::N * UB<C>::COL)) - 1;
};

// a*b - c*d (+ p) with one reduction: sparse (P-384) and general (BLS12-381) Montgomery fields
template <class C, int K1, int V1, int K2, int V2, int K3, int V3, int K4, int V4>
ECCX_DEV auto u_mul_sub(const U<C, K1, V1>& a, const U<C, K2, V2>& b, const U<C, K3, V3>& c, const U<C, K4, V4>& d) {
  static_assert(UBS<C>::OK, "implemented for the Montgomery fields with room for signed columns");
  static_assert(K1 * K2 <= UBS<C>::KKS && K3 * K4 <= UBS<C>::KKS, "a product overflows the signed columns");
  static_assert(K3 <= 7 && K4 <= 7, "negated / signed operands must stay below 2^31");
  static_assert((uint32_t)(V1 * V2) < C::RP && (uint32_t)(V3 * V4) < C::RP, "result outside (0, 3p)");
  U<C, 1, 3> r;
  if constexpr (UB<C>::SPARSE) u_mul_sub_core_sparse<C, false>(r.v, a.v, b.v, c.v, d.v);
  else u_mul_sub_core_mont<C, false>(r.v, a.v, b.v, c.v, d.v);
  return r;
}
// a*b - 2*c^2 (+ p) with one reduction
template <class C, int K1, int V1, int K2, int V2, int K3, int V3>
ECCX_DEV auto u_mul_sub_2sqr(const U<C, K1, V1>& a, const U<C, K2, V2>& b, const U<C, K3, V3>& c) {
  static_assert(UBS<C>::OK, "implemented for the Montgomery fields with room for signed columns");
  static_assert(K1 * K2 <= UBS<C>::KKS && 2 * K3 * K3 <= UBS<C>::KKS, "a product overflows the signed columns");
  static_assert(2 * K3 <= 7, "doubled / negated operands must stay below 2^31");
  static_assert((uint32_t)(V1 * V2) < C::RP && (uint32_t)(2 * V3 * V3) < C::RP, "result outside (0, 3p)");
  U<C, 1, 3> r;
  if constexpr (UB<C>::SPARSE) u_mul_sub_core_sparse<C, true>(r.v, a.v, b.v, c.v, c.v);
  else u_mul_sub_core_mont<C, true>(r.v, a.v, b.v, c.v, c.v);
  return r;
}

template <class C, int K1, int V1>
ECCX_DEV auto u_sqr(const U<C, K1, V1>& a) {
  if constexpr (!UB<C>::ksq_ok(K1) || (UB<C>::vout(V1, V1) > 3)) {
    return u_sqr(u_reduce(a));
  } else {
    U<C, 1, UB<C>::vout(V1, V1)> r;
    u_mul_core<C, true, false>(r.v, a.v, a.v);
    return r;
  }
}

// multiply by a compile-time constant given as tight digits of a value below p (R^2, 1)
template <class C, int K1, int V1>
ECCX_DEV auto u_mul_k(const U<C, K1, V1>& a, const uint32_t (&k)[C::N]) {
  if constexpr (!UB<C>::kk_ok(K1, 1) || (UB<C>::vout(V1, 1) > 3)) {
    return u_mul_k(u_reduce(a), k);
  } else {
    U<C, 1, UB<C>::vout(V1, 1)> r;
    u_mul_core<C, false, true>(r.v, a.v, k);
    return r;
  }
}

// multiply by a small constant k < 2^20 (kind 3 only: the ladder's (A + 2) / 4): one mad per limb
template <class C, int K1, int V1>
ECCX_DEV U<C, 1, 2> u_mul_small(const U<C, K1, V1>& a, uint32_t k) {
  static_assert(C::KIND == UK_PM19, "written for 2^255 - 19");
  constexpr int N = C::N;
  uint32_t t[N];
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    umad1_v(acc, a.v[i], k);
    t[i] = (uint32_t)acc & C::MASK;
    acc >>= C::B;
  }
  u_pm_tail<C>(t, acc);
  U<C, 1, 2> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = t[i];
  return r;
}

template <class C, int K1, int V1, int K2, int V2>
ECCX_DEV auto u_add(const U<C, K1, V1>& a, const U<C, K2, V2>& b) {
  if constexpr (K1 + K2 > UB<C>::KMAX) {
    if constexpr (K1 >= K2) return u_add(u_reduce(a), b);
    else return u_add(a, u_reduce(b));
  } else {
    U<C, K1 + K2, V1 + V2> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
  }
}

// a - b (+ 4p): b must be tight and below 3p so that BIAS - b has no negative limb
template <class C, int K1, int V1, int K2, int V2>
ECCX_DEV auto u_sub(const U<C, K1, V1>& a, const U<C, K2, V2>& b) {
  if constexpr (K2 != 1 || V2 > 3) {
    return u_sub(a, u_reduce(b));
  } else if constexpr (K1 + 2 > UB<C>::KMAX) {
    return u_sub(u_reduce(a), b);
  } else {
    U<C, K1 + 2, V1 + 4> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = a.v[i] + (C::BIAS[i] - b.v[i]);
    return r;
  }
}

template <class C, int K2, int V2>
ECCX_DEV auto u_neg(const U<C, K2, V2>& b) {
  if constexpr (K2 != 1 || V2 > 3) {
    return u_neg(u_reduce(b));
  } else {
    U<C, 2, 4> r;
#pragma unroll
    for (int i = 0; i < C::N; ++i) r.v[i] = C::BIAS[i] - b.v[i];
    return r;
  }
}

#ifndef ECCX_SELECT_E64
#define ECCX_SELECT_E64 0  // 1: A/B -- every u_select through v_cndmask_b32_e64 with an SGPR-pair mask (VCC-masked selects issue 3-4x slower)
#endif
template <class C, int K, int V>
ECCX_DEV void u_select(U<C, K, V>& r, bool take_a, const U<C, K, V>& a, const U<C, K, V>& b) {
#if ECCX_SELECT_E64
  const uint64_t m = __builtin_amdgcn_uicmp((uint32_t)take_a, 1u, 32);
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    uint32_t t = b.v[i];
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(t) : "v"(a.v[i]), "s"(m));
    r.v[i] = t;
  }
#else
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = take_a ? a.v[i] : b.v[i];
#endif
}

template <class C, int K, int V>
ECCX_DEV bool u_limbs_all_zero(const U<C, K, V>& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) acc |= a.v[i];
  return acc == 0;
}

template <class C, int K, int V>
ECCX_DEV void u_set_zero(U<C, K, V>& a) {
#pragma unroll
  for (int i = 0; i < C::N; ++i) a.v[i] = 0;
}

// value == 0 (mod p) for the OUTPUT OF u_reduce (exact digits, below 3p): its digits equal
// those of 0, p or 2p; the low limb filters nearly everything before the full compare.
template <class C>
ECCX_DEV bool u_is_zero_mod_p(const U<C, 1, 3>& a) {
  const uint32_t l0 = a.v[0];
  if (!(l0 == 0 || l0 == C::P[0] || l0 == C::P2[0])) return false;
  uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    d0 |= a.v[i];
    d1 |= a.v[i] ^ C::P[i];
    d2 |= a.v[i] ^ C::P2[i];
  }
  return d0 == 0 || d1 == 0 || d2 == 0;
}

// value == 0 (mod p) for the output of u_reduce, without the early exit of u_is_zero_mod_p
template <class C>
ECCX_DEV bool u_is_zero_mod_p_ct(const U<C, 1, 3>& a) {
  uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    d0 |= a.v[i];
    d1 |= a.v[i] ^ C::P[i];
    d2 |= a.v[i] ^ C::P2[i];
  }
  return (d0 == 0) | (d1 == 0) | (d2 == 0);
}

// ---- selects the compiler cannot turn into control flow (secret-scalar kernels) -----------------------
// hipcc is free to compile `take ? a : b` as an EXEC-masked region with an s_cbranch_execz around whatever
// feeds it (it does: a table scan written with ?: came out with its LDS reads inside such regions, skipped
// when no lane of the wave takes the entry -- a branch on scalar digits).  These forms are opaque: the data
// goes through v_cndmask_b32 inside an asm statement, so both operands are always computed, every load that
// feeds them is always issued, and nothing is skipped.  (Not `volatile`:
// a volatile asm is a scheduling barrier, and a scan whose every 16-byte load waits for the select before it
// runs at LDS latency -- measured 5x slower.  Without it the statement is still opaque and still consumes both
// operands wherever it executes; tools/isa_histogram.py --branches is the check that no conditional branch
// was built around any of them.)
ECCX_DEV uint64_t ct_mask(bool take) { return __builtin_amdgcn_uicmp((uint32_t)take, 1u, 32 /* ICMP_EQ */); }
// o[i] = lane in m ? a[i] : o[i], four words.  The mask stays in an SGPR pair and the select is the VOP3 form:
// on gfx950 a v_cndmask_b32 that takes its mask from VCC issues every 16-18 cycles, the same instruction with
// an SGPR-pair mask (or v_bfi_b32 with a VGPR mask) every 4.5-5 (profiles/r03_select_rates.jsonl).
ECCX_DEV void ct_cmov4(uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3,
                       uint64_t m) {
  asm("v_cndmask_b32_e64 %0, %0, %4, %8\n\t"
      "v_cndmask_b32_e64 %1, %1, %5, %8\n\t"
      "v_cndmask_b32_e64 %2, %2, %6, %8\n\t"
      "v_cndmask_b32_e64 %3, %3, %7, %8"
      : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(m));
}
ECCX_DEV void ct_cmov1(uint32_t& o, uint32_t a, uint64_t m) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(o) : "v"(a), "s"(m));
}
// r = take ? a : r
template <class C, int K, int V>
ECCX_DEV void u_cmov_ct(U<C, K, V>& r, uint64_t m, const U<C, K, V>& a) {
  constexpr int N = C::N;
#pragma unroll
  for (int i = 0; i + 4 <= N; i += 4) ct_cmov4(r.v[i], r.v[i + 1], r.v[i + 2], r.v[i + 3], a.v[i], a.v[i + 1], a.v[i + 2], a.v[i + 3], m);
#pragma unroll
  for (int i = N - N % 4; i < N; ++i) ct_cmov1(r.v[i], a.v[i], m);
}
// r = take_a ? a : b
template <class C, int K, int V>
ECCX_DEV void u_select_ct(U<C, K, V>& r, bool take_a, const U<C, K, V>& a, const U<C, K, V>& b) {
  U<C, K, V> t = b;
  u_cmov_ct(t, ct_mask(take_a), a);
  r = t;
}

// ---- conversions with the saturated representation (plain integers, L x 32) --------------
// plain saturated integer -> tight digits (value unchanged; below 2p for canonical input)
template <class C>
ECCX_DEV U<C, 1, 2> u_from_sat(const Fe<C::Sat::L>& s) {
  constexpr int L = C::Sat::L;
  static_assert(C::B * C::N >= C::PBITS, "digits must cover the field");
  U<C, 1, 2> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    const int bit = C::B * i;
    const int w = bit >> 5, sh = bit & 31;
    uint64_t lo = 0;
    if (w < L) lo = s.v[w];
    if (w + 1 < L) lo |= (uint64_t)s.v[w + 1] << 32;
    r.v[i] = (uint32_t)(lo >> sh) & C::MASK;
  }
  return r;
}

// exact tight digits of a value below 2^(32L) -> saturated limbs
template <class C, int V>
ECCX_DEV void u_to_sat(Fe<C::Sat::L>& s, const U<C, 1, V>& a) {
#pragma unroll
  for (int w = 0; w < C::Sat::L; ++w) {
    // bits [32w, 32w+32) come from at most three digits
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < C::N; ++i) {
      const int bit = C::B * i;
      if (bit + C::B > 32 * w && bit < 32 * w + 32) {
        if (bit >= 32 * w) acc |= (uint64_t)a.v[i] << (bit - 32 * w);
        else acc |= (uint64_t)a.v[i] >> (32 * w - bit);
      }
    }
    s.v[w] = (uint32_t)acc;
  }
}

// plain canonical integer (saturated) -> the field's working form (Montgomery or plain), tight
template <class C>
ECCX_DEV U<C, 1, 2> u_to_mont(const Fe<C::Sat::L>& plain) {
  if constexpr (UB<C>::MONT) return u_as<1, 2>(u_mul_k<C>(u_from_sat<C>(plain), C::R2));
  else return u_from_sat<C>(plain);
}

// working form -> canonical plain integer in saturated limbs (the unique residue < p)
template <class C, int K, int V>
ECCX_DEV void u_to_canonical(Fe<C::Sat::L>& out, const U<C, K, V>& a) {
  Fe<C::Sat::L> s;
  if constexpr (UB<C>::MONT) {
    uint32_t one[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) one[i] = (i == 0) ? 1u : 0u;
    // (a + m p) / R with a < 3p << R: the result is in [0, p]
    auto r = u_mul_k<C>(u_reduce(a), one);
    u_to_sat<C>(s, r);
  } else {
    // twice: the second pass sees exact digits, so its quotient is exact and the result <= p + 2
    auto r = u_reduce(u_reduce(a));
    u_to_sat<C>(s, r);
  }
  uint32_t t[C::Sat::L];
#pragma unroll
  for (int i = 0; i < C::Sat::L; ++i) t[i] = s.v[i];
  cond_sub_p<typename C::Sat>(out, t, 0u);  // p -> 0
}

// working form -> the SATURATED twin's working form (canonical Montgomery limbs for the Montgomery
// fields, canonical plain integers otherwise): what fe.hpp code (normalisation, result tails) takes
template <class C, int K, int V>
ECCX_DEV void u_to_sat_mont(Fe<C::Sat::L>& out, const U<C, K, V>& a) {
  if constexpr (UB<C>::MONT) {
    // (a * R_sat + m p) / R with a < 3p, R_sat < p: below p (1 + 3p/R), one conditional subtraction
    auto r = u_mul_k<C>(u_reduce(a), C::RS);
    Fe<C::Sat::L> s;
    u_to_sat<C>(s, r);
    uint32_t t[C::Sat::L];
#pragma unroll
    for (int i = 0; i < C::Sat::L; ++i) t[i] = s.v[i];
    cond_sub_p<typename C::Sat>(out, t, 0u);
  } else {
    u_to_canonical<C>(out, a);
  }
}

}  // namespace eccx