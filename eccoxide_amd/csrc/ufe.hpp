// Unsaturated field arithmetic for the fast kernels: N limbs of B < 32 bits in 32-bit
// registers, Montgomery with R = 2^(B*N).
//
// Why: on gfx950 a carry instruction costs as much as a multiply (profiles/r01_valu_rates.jsonl),
// so the saturated multiplier (fe.hpp) spends half its issue slots on v_addc_co_u32.  With
// 29-bit limbs a column of the product -- at most 9 limb products plus 9 reduction products,
// each below 2^58 -- fits a 64-bit accumulator with room to spare: the unit of work is ONE
// v_mad_u64_u32, squares really cost half the cross products, additions are 9 independent
// v_add_u32 and there is no conditional subtraction.  Measured (tools/ubench/fe_bench29.hip,
// 4 waves per SIMD): 773 cycles per P-256 product and 631 per square against 1046 for the
// saturated form.
//
// The price is bookkeeping, done at compile time.  Every value carries two bounds in its
// type, U<C, K, V>:
//   K  every limb is < K * 2^B          (K = 1: "tight", the unique radix-2^B digits;
//                                         the top limb holds whatever is left above)
//   V  the integer value is < V * p
// and each operation states what it needs and what it returns (static_assert):
//   mul / sqr   need 9*K1*K2*2^(2B) + 9*2^(2B) + carry < 2^64  <=>  K1*K2 <= 6 (sqr: K <= 2),
//               and V1*V2 <= 64; return tight, V = 2 (V1*V2 <= 32) or 3
//   add         K1+K2, V1+V2, no instructions beyond N adds
//   sub         a + BIAS - b with BIAS = 4p spread so no limb borrows; needs b tight, Vb <= 3;
//               returns K1+2, V1+4
//   reduce      one signed carry chain that also subtracts q*p, q ~ value >> 256: needs K <= 7,
//               V <= 31; returns tight, V = 3
// Values of the reference (fiat-crypto Montgomery, src/curve/fiat/p256_64.rs) are reproduced
// modulo p only; canonical bytes are produced at the very end (u_to_canonical).
#pragma once
#include "fe.hpp"

namespace eccx {

#include "umad_chunks.inc"

template <class C, int K, int V>
struct U {
  static_assert(K >= 1 && K <= 7, "limb bound out of range (limbs must stay below 2^32)");
  static_assert(V >= 1 && V <= 31, "value bound out of range (must stay below 2^(B*N))");
  uint32_t v[C::N];
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

// Montgomery product / square core on raw limb arrays; bounds are checked by the typed wrappers.
template <class C, bool SQR, bool BCONST>
ECCX_DEV void u_mul_core(uint32_t (&r)[C::N], const uint32_t (&a)[C::N], const uint32_t (&b)[C::N]) {
  constexpr int N = C::N;
  static_assert(N <= 9, "column queue holds 9 products");
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
#pragma unroll
    for (int i = lo; i <= (k < N ? k - 1 : N - 1); ++i) {
      if (C::PP1[k - i] != 0) qm.push(acc, m[i], C::PP1[k - i]);
    }
    qm.flush(acc);
    // -p^-1 mod 2^B = 1, so m[k] is the low limb itself, and m*p = m*(p+1) - m: the "- m"
    // clears that limb (the shift drops it), m*(p+1) goes to the columns above
    if (k < N) m[k] = (uint32_t)acc & C::MASK;
    else t[k - N] = (uint32_t)acc & C::MASK;
    acc >>= C::B;
  }
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

template <class C, int K1, int V1, int K2, int V2>
ECCX_DEV U<C, 1, (V1 * V2 <= 32 ? 2 : 3)> u_mul(const U<C, K1, V1>& a, const U<C, K2, V2>& b) {
  static_assert(K1 * K2 <= 6, "product column would overflow 64 bits");
  static_assert(V1 * V2 <= 64, "Montgomery output would not stay below 3p");
  U<C, 1, (V1 * V2 <= 32 ? 2 : 3)> r;
  u_mul_core<C, false, false>(r.v, a.v, b.v);
  return r;
}

template <class C, int K1, int V1>
ECCX_DEV U<C, 1, (V1 * V1 <= 32 ? 2 : 3)> u_sqr(const U<C, K1, V1>& a) {
  static_assert(K1 <= 2, "square column would overflow 64 bits");
  static_assert(V1 * V1 <= 64, "Montgomery output would not stay below 3p");
  U<C, 1, (V1 * V1 <= 32 ? 2 : 3)> r;
  u_mul_core<C, true, false>(r.v, a.v, a.v);
  return r;
}

// multiply by a compile-time constant given as tight digits (R^2, 1)
template <class C, int K1, int V1>
ECCX_DEV U<C, 1, 2> u_mul_k(const U<C, K1, V1>& a, const uint32_t (&k)[C::N]) {
  static_assert(K1 <= 6 && V1 <= 31, "bounds");
  U<C, 1, 2> r;
  u_mul_core<C, false, true>(r.v, a.v, k);
  return r;
}

template <class C, int K1, int V1, int K2, int V2>
ECCX_DEV U<C, K1 + K2, V1 + V2> u_add(const U<C, K1, V1>& a, const U<C, K2, V2>& b) {
  U<C, K1 + K2, V1 + V2> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}

// a - b (+ 4p): b must be tight and below 3p so that BIAS - b has no negative limb
template <class C, int K1, int V1, int V2>
ECCX_DEV U<C, K1 + 2, V1 + 4> u_sub(const U<C, K1, V1>& a, const U<C, 1, V2>& b) {
  static_assert(V2 <= 3, "subtrahend too large for the 4p bias");
  U<C, K1 + 2, V1 + 4> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = a.v[i] + (C::BIAS[i] - b.v[i]);
  return r;
}

template <class C, int V2>
ECCX_DEV U<C, 2, 4> u_neg(const U<C, 1, V2>& b) {
  static_assert(V2 <= 3, "operand too large for the 4p bias");
  U<C, 2, 4> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = C::BIAS[i] - b.v[i];
  return r;
}

// Weak reduction: propagate carries and take off q*p with q = (value >> 256) estimated from
// the two top limbs (it may be one too small), in one signed chain.  Result: tight digits of
// a value in [0, 2^256 + p) -- below 3p.
template <class C, int K1, int V1>
ECCX_DEV U<C, 1, 3> u_reduce(const U<C, K1, V1>& a) {
  constexpr int N = C::N;
  constexpr int TOPSHIFT = 256 - C::B * (N - 1);  // bit 256 inside the top limb (24 for 9 x 29)
  static_assert(C::B * (N - 1) < 256 && C::B * N > 256, "layout assumed by the quotient estimate");
  const uint32_t q = (a.v[N - 1] + (a.v[N - 2] >> C::B)) >> TOPSHIFT;
  const int32_t nq = -(int32_t)q;
  U<C, 1, 3> r;
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    acc += (int64_t)a.v[i];
    if (C::P[i] != 0) acc += (int64_t)nq * (int64_t)(int32_t)C::P[i];
    if (i < N - 1) {
      r.v[i] = (uint32_t)acc & C::MASK;
      acc >>= C::B;  // arithmetic shift: a negative partial sum borrows from the next limb
    } else {
      r.v[i] = (uint32_t)acc;
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

template <class C, int K, int V>
ECCX_DEV void u_select(U<C, K, V>& r, bool take_a, const U<C, K, V>& a, const U<C, K, V>& b) {
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = take_a ? a.v[i] : b.v[i];
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

// value == 0 (mod p) for a TIGHT value below 3p: its digits equal those of 0, p or 2p.
// k*p = -k (mod 2^B), so the low limb filters all but 3 in 2^B values before the full compare.
template <class C>
ECCX_DEV bool u_is_zero_mod_p(const U<C, 1, 3>& a) {
  const uint32_t l0 = a.v[0];
  if (!(l0 == 0 || l0 >= C::MASK - 1)) return false;
  uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    d0 |= a.v[i];
    d1 |= a.v[i] ^ C::P[i];
    d2 |= a.v[i] ^ C::P2[i];
  }
  return d0 == 0 || d1 == 0 || d2 == 0;
}

// ---- conversions with the saturated representation (plain integers, 8 x 32) ------------
// plain saturated integer (< 2^256) -> tight digits (value unchanged)
template <class C>
ECCX_DEV U<C, 1, 2> u_from_sat(const Fe<C::Sat::L>& s) {
  static_assert(C::Sat::L * 32 == 256, "conversion written for 256-bit fields");
  U<C, 1, 2> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) {
    const int bit = C::B * i;
    const int w = bit >> 5, sh = bit & 31;
    uint64_t lo = s.v[w];
    if (w + 1 < C::Sat::L) lo |= (uint64_t)s.v[w + 1] << 32;
    r.v[i] = (uint32_t)(lo >> sh) & C::MASK;
  }
  return r;
}

// tight digits of a value < 2^256 -> saturated limbs
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

// plain canonical integer (saturated) -> Montgomery form, tight
template <class C>
ECCX_DEV U<C, 1, 2> u_to_mont(const Fe<C::Sat::L>& plain) {
  return u_mul_k<C>(u_from_sat<C>(plain), C::R2);
}

// Montgomery form -> canonical plain integer in saturated limbs (the unique residue < p)
template <class C, int K, int V>
ECCX_DEV void u_to_canonical(Fe<C::Sat::L>& out, const U<C, K, V>& a) {
  uint32_t one[C::N];
#pragma unroll
  for (int i = 0; i < C::N; ++i) one[i] = (i == 0) ? 1u : 0u;
  // (a + m p) / R with a < 31p << R: the result is in [0, p]
  U<C, 1, 2> r = u_mul_k<C>(a, one);
  Fe<C::Sat::L> s;
  u_to_sat<C>(s, r);
  uint32_t t[C::Sat::L];
#pragma unroll
  for (int i = 0; i < C::Sat::L; ++i) t[i] = s.v[i];
  cond_sub_p<typename C::Sat>(out, t, 0u);  // p -> 0
}

}  // namespace eccx
