// Field arithmetic for the gfx950 scalar-multiplication kernels: one field
// element per lane, saturated 32-bit limbs held in VGPRs, word-by-word Montgomery
// with R = 2^(32*L).
//
// Replaces (value-for-value) the reference's fiat-crypto backends
//   src/curve/fiat/p256_64.rs:265-549 (mul), :562-845 (square), :859-927 (add),
//   :941-1000 (sub), :1013-1071 (opp), :1259/:1084 (to/from_montgomery)
// and the same entry points of p384_64.rs, bls12_381_64.rs; for the reference's
// unsaturated-Solinas fields (p521_64.rs, curve25519_64.rs) only the VALUE of each
// operation is reproduced (parity is on canonical bytes).
// Every result is the unique canonical residue in [0, p), like fiat's outputs
// (p256_64.rs:8-13), so un-normalised (X:Y:Z) coordinates can be compared too.
//
// Machine model (profiles/r01_valu_rates.jsonl, measured on MI355X): v_mad_u64_u32
// issues at half the v_add_u32 rate (4.8 vs 2.4 cycles per wave-instruction per
// SIMD) and so do the carry instructions (v_add_co/v_addc_co, 4.2).  The multiply
// is therefore organised as one product-scanning (Comba) chain whose unit of work
// is a 2-instruction MAC: v_mad_u64_u32 into a 64-bit column accumulator plus one
// v_addc_co_u32 into a third word.  Montgomery reduction is interleaved in the
// same column chain (FIPS form), and limbs of the modulus that are zero are
// skipped at compile time (P-256: 3 of 8, P-384: 2 of 12).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eccx {

#define ECCX_DEV __device__ __forceinline__

template <int L>
struct Fe {
  uint32_t v[L];
};

#include "mac_chunks.inc"

// Queue of pending multiply-accumulates for one column of the product: (hi:lo) +=
// sum x[i]*y[i].  hipcc pads a wait state after every asm statement, so the MACs of
// a column are issued in chunks of up to 8 per statement.  All indices are
// compile-time constants once the callers' loops are unrolled, so the queue is
// pure register renaming.  K: second factors are wave-uniform constants (SGPRs).
template <bool K>
struct MacQ {
  uint32_t x[8], y[8];
  int n = 0;
  ECCX_DEV void push(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    x[n] = a;
    y[n] = b;
    if (++n == 8) flush(lo, hi);
  }
  ECCX_DEV void flush(uint64_t& lo, uint32_t& hi) {
#define ECCX_Q(i) x[i], y[i]
    if constexpr (K) {
      switch (n) {
        case 1: mac1_k(lo, hi, ECCX_Q(0)); break;
        case 2: mac2_k(lo, hi, ECCX_Q(0), ECCX_Q(1)); break;
        case 3: mac3_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2)); break;
        case 4: mac4_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3)); break;
        case 5: mac5_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4)); break;
        case 6: mac6_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5)); break;
        case 7: mac7_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6)); break;
        case 8: mac8_k(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7)); break;
        default: break;
      }
    } else {
      switch (n) {
        case 1: mac1_v(lo, hi, ECCX_Q(0)); break;
        case 2: mac2_v(lo, hi, ECCX_Q(0), ECCX_Q(1)); break;
        case 3: mac3_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2)); break;
        case 4: mac4_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3)); break;
        case 5: mac5_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4)); break;
        case 6: mac6_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5)); break;
        case 7: mac7_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6)); break;
        case 8: mac8_v(lo, hi, ECCX_Q(0), ECCX_Q(1), ECCX_Q(2), ECCX_Q(3), ECCX_Q(4), ECCX_Q(5), ECCX_Q(6), ECCX_Q(7)); break;
        default: break;
      }
    }
#undef ECCX_Q
    n = 0;
  }
};

// column shift: (hi:lo) >>= 32
ECCX_DEV void col_shift(uint64_t& lo, uint32_t& hi) {
  lo = (lo >> 32) | ((uint64_t)hi << 32);
  hi = 0;
}

ECCX_DEV uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) {
  uint32_t co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
}
ECCX_DEV uint32_t subb(uint32_t a, uint32_t b, uint32_t& bw) {
  uint32_t bo;
  uint32_t r = __builtin_subc(a, b, bw, &bo);
  bw = bo;
  return r;
}

// r = (carry:t) >= P ? t - P : t
template <class C>
ECCX_DEV void cond_sub_p(Fe<C::L>& r, const uint32_t (&t)[C::L], uint32_t carry) {
  uint32_t u[C::L];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) u[i] = subb(t[i], C::P[i], bw);
  bool use_u = (carry != 0) | (bw == 0);
#pragma unroll
  for (int i = 0; i < C::L; ++i) r.v[i] = use_u ? u[i] : t[i];
}

template <class C>
ECCX_DEV void fe_add(Fe<C::L>& r, const Fe<C::L>& a, const Fe<C::L>& b) {
  uint32_t t[C::L];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) t[i] = addc(a.v[i], b.v[i], c);
  cond_sub_p<C>(r, t, c);
}

template <class C>
ECCX_DEV void fe_sub(Fe<C::L>& r, const Fe<C::L>& a, const Fe<C::L>& b) {
  uint32_t t[C::L];
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) t[i] = subb(a.v[i], b.v[i], bw);
  uint32_t mask = 0u - bw;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) r.v[i] = addc(t[i], C::P[i] & mask, c);
}

template <class C>
ECCX_DEV void fe_neg(Fe<C::L>& r, const Fe<C::L>& a) {
  Fe<C::L> z;
#pragma unroll
  for (int i = 0; i < C::L; ++i) z.v[i] = 0;
  fe_sub<C>(r, z, a);
}

template <class C>
ECCX_DEV bool fe_is_zero(const Fe<C::L>& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) acc |= a.v[i];
  return acc == 0;
}

template <class C>
ECCX_DEV bool fe_eq(const Fe<C::L>& a, const Fe<C::L>& b) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) acc |= a.v[i] ^ b.v[i];
  return acc == 0;
}

template <class C>
ECCX_DEV void fe_set(Fe<C::L>& r, const uint32_t (&k)[C::L]) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) r.v[i] = k[i];
}

template <class C>
ECCX_DEV void fe_zero(Fe<C::L>& r) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) r.v[i] = 0;
}

template <class C>
ECCX_DEV void fe_select(Fe<C::L>& r, bool take_a, const Fe<C::L>& a, const Fe<C::L>& b) {
#pragma unroll
  for (int i = 0; i < C::L; ++i) r.v[i] = take_a ? a.v[i] : b.v[i];
}

// Montgomery product r = a*b/R mod P, canonical.  Product-scanning with the
// reduction multiples m[i]*P[j] accumulated in the same column chain.
// Mersenne prime p = 2^K - 1 (P-521): plain representation, full product then fold
// (2^K = 1 mod p).  L^2 MACs instead of the 2 L^2 of a Montgomery product.  Stands in for
// the reference's unsaturated-Solinas fiat_p521_carry_mul (src/curve/fiat/p521_64.rs:297)
// by value; inputs canonical (< p), output canonical.
template <class C, bool BCONST>
ECCX_DEV void fe_mul_mersenne(Fe<C::L>& r, const Fe<C::L>& a, const uint32_t (&b)[C::L]) {
  constexpr int L = C::L;
  constexpr int TOPB = C::MERSENNE - 32 * (L - 1);  // bits in the top limb (9 for P-521)
  constexpr uint32_t TOPMASK = (1u << TOPB) - 1;
  uint32_t t[2 * L];
  uint64_t lo = 0;
  uint32_t hi = 0;
  MacQ<BCONST> qa;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; ++k) {
#pragma unroll
    for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); ++i) qa.push(lo, hi, a.v[i], b[k - i]);
    qa.flush(lo, hi);
    t[k] = (uint32_t)lo;
    col_shift(lo, hi);
  }
  t[2 * L - 1] = (uint32_t)lo;
  // T = hi * 2^K + lo  ->  lo + hi
  uint32_t s[L];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    uint32_t h = (t[L - 1 + i] >> TOPB) | (t[L + i] << (32 - TOPB));
    uint32_t l = (i == L - 1) ? (t[i] & TOPMASK) : t[i];
    s[i] = addc(l, h, c);
  }
  // at most one bit above 2^K is left: fold it once more (cannot overflow again)
  uint32_t top = s[L - 1] >> TOPB;
  s[L - 1] &= TOPMASK;
  c = 0;
  s[0] = addc(s[0], top, c);
#pragma unroll
  for (int i = 1; i < L; ++i) s[i] = addc(s[i], 0u, c);
  cond_sub_p<C>(r, s, 0u);
}

// p = 2^255 - 19: plain representation, full 8x8 product, then fold with 2^256 = 38 and
// 2^255 = 19 (mod p).  72 multiply instructions instead of the 128 of a Montgomery product.
// Stands in for fiat_25519_carry_mul (src/curve/fiat/curve25519_64.rs:217) by value.
template <class C, bool BCONST>
ECCX_DEV void fe_mul_pm19(Fe<C::L>& r, const Fe<C::L>& a, const uint32_t (&b)[C::L]) {
  constexpr int L = C::L;  // 8
  uint32_t t[2 * L];
  uint64_t lo = 0;
  uint32_t hi = 0;
  MacQ<BCONST> qa;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; ++k) {
#pragma unroll
    for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); ++i) qa.push(lo, hi, a.v[i], b[k - i]);
    qa.flush(lo, hi);
    t[k] = (uint32_t)lo;
    col_shift(lo, hi);
  }
  t[2 * L - 1] = (uint32_t)lo;
  // u = lo + 38 * hi  (no 64-bit overflow: 38 * 2^32 + 2^32 + 2^38 < 2^64)
  uint32_t u[L];
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    acc = (uint64_t)t[L + i] * 38u + t[i] + (acc >> 32);
    u[i] = (uint32_t)acc;
  }
  // remaining weight: (acc >> 32) * 2^256 + bit 255 of u  ->  19 * (2c + bit255)
  uint32_t top = ((uint32_t)(acc >> 32) << 1) | (u[L - 1] >> 31);
  u[L - 1] &= 0x7fffffffu;
  uint32_t c = 0;
  u[0] = addc(u[0], top * 19u, c);
#pragma unroll
  for (int i = 1; i < L; ++i) u[i] = addc(u[i], 0u, c);
  cond_sub_p<C>(r, u, 0u);  // u < 2^255 + 19 * 80 < 2p
}


template <class C, bool BCONST = false>
ECCX_DEV void fe_mul_impl(Fe<C::L>& r, const Fe<C::L>& a, const uint32_t (&b)[C::L]) {
  if constexpr (C::MERSENNE != 0) {
    fe_mul_mersenne<C, BCONST>(r, a, b);
    return;
  }
  if constexpr (C::PM19 != 0) {
    fe_mul_pm19<C, BCONST>(r, a, b);
    return;
  }
  constexpr int L = C::L;
  // reduction digits: p + 1 when -p^-1 mod 2^32 = 1 (p = -1 mod 2^32; fewer non-zero limbs), else p
  auto PR = [](int j) constexpr { return (C::N0 == 1u) ? C::PP1[j] : C::P[j]; };
  uint32_t m[L];
  uint32_t t[L];
  uint64_t lo = 0;
  uint32_t hi = 0;
  MacQ<BCONST> qa;  // a[i] * b[j]
  MacQ<true> qm;    // m[i] * PR[j]
#pragma unroll
  for (int k = 0; k < L; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) qa.push(lo, hi, a.v[i], b[k - i]);
    qa.flush(lo, hi);
#pragma unroll
    for (int i = 0; i < k; ++i) {
      if (PR(k - i) != 0) qm.push(lo, hi, m[i], PR(k - i));
    }
    qm.flush(lo, hi);
    if constexpr (C::N0 == 1u) {
      m[k] = (uint32_t)lo;  // m*p = m*(p+1) - m: "- m" clears the low word, the shift drops it
    } else {
      m[k] = (uint32_t)lo * C::N0;
      mac1_k(lo, hi, m[k], C::P[0]);
    }
    col_shift(lo, hi);
  }
#pragma unroll
  for (int k = L; k < 2 * L; ++k) {
#pragma unroll
    for (int i = k - L + 1; i < L; ++i) qa.push(lo, hi, a.v[i], b[k - i]);
    qa.flush(lo, hi);
#pragma unroll
    for (int i = k - L + 1; i < L; ++i) {
      if (PR(k - i) != 0) qm.push(lo, hi, m[i], PR(k - i));
    }
    qm.flush(lo, hi);
    t[k - L] = (uint32_t)lo;
    col_shift(lo, hi);
  }
  cond_sub_p<C>(r, t, (uint32_t)lo);
}

template <class C>
ECCX_DEV void fe_mul(Fe<C::L>& r, const Fe<C::L>& a, const Fe<C::L>& b) {
  fe_mul_impl<C, false>(r, a, b.v);
}

// multiply by a curve constant kept in SGPRs (b, 3b, 2d, R^2, 1)
template <class C>
ECCX_DEV void fe_mul_k(Fe<C::L>& r, const Fe<C::L>& a, const uint32_t (&k)[C::L]) {
  fe_mul_impl<C, true>(r, a, k);
}

// Montgomery square.  On gfx950 the carry instructions cost as much as the
// multiplies (see header), so the usual "cross products once, then double" form
// saves nothing in saturated limbs: a square is issued as a product.
template <class C>
ECCX_DEV void fe_sqr(Fe<C::L>& r, const Fe<C::L>& a) {
  fe_mul_impl<C, false>(r, a, a.v);
}

template <class C>
ECCX_DEV void fe_to_mont(Fe<C::L>& r, const Fe<C::L>& a) {
  fe_mul_k<C>(r, a, C::R2);
}

// r = a / R mod P (leave the Montgomery domain): multiply by 1
template <class C>
ECCX_DEV void fe_from_mont(Fe<C::L>& r, const Fe<C::L>& a) {
  constexpr int L = C::L;
  uint32_t one[L];
#pragma unroll
  for (int i = 0; i < L; ++i) one[i] = (i == 0) ? 1u : 0u;
  // a generic multiply by the constant 1 (the compiler drops the zero-limb MACs'
  // operands but keeps the chain); cheap enough, used once per output coordinate
  fe_mul_impl<C, true>(r, a, one);
}

// ---- byte I/O -----------------------------------------------------------------
// Field elements cross the ABI as FB big-endian (Weierstrass) or little-endian (edwards25519)
// bytes, one record per lane.  Where FB is a multiple of 4 and the address allows it the bytes
// move as 16-byte or 4-byte words (a lane's record is contiguous, so a wavefront's accesses
// fill whole cache lines in a few instructions instead of FB byte accesses per lane); any
// other address, and the 66-byte P-521 elements, take the byte path.
template <int W>
ECCX_DEV bool io_words_load(uint32_t (&w)[W], const uint8_t* __restrict__ in) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(in);
  if constexpr (W % 4 == 0) {
    if ((addr & 15u) == 0) {
      const uint4* q = reinterpret_cast<const uint4*>(in);
#pragma unroll
      for (int i = 0; i < W / 4; ++i) {
        const uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
      }
      return true;
    }
  }
  if ((addr & 3u) == 0) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(in);
#pragma unroll
    for (int i = 0; i < W; ++i) w[i] = q[i];
    return true;
  }
  return false;
}
template <int W>
ECCX_DEV bool io_words_store(uint8_t* __restrict__ out, const uint32_t (&w)[W]) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(out);
  if constexpr (W % 4 == 0) {
    if ((addr & 15u) == 0) {
      uint4* q = reinterpret_cast<uint4*>(out);
#pragma unroll
      for (int i = 0; i < W / 4; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
      return true;
    }
  }
  if ((addr & 3u) == 0) {
    uint32_t* q = reinterpret_cast<uint32_t*>(out);
#pragma unroll
    for (int i = 0; i < W; ++i) q[i] = w[i];
    return true;
  }
  return false;
}

// big-endian FB bytes -> integer limbs (no range check, like from_bytes_unchecked_be,
// field_macros.rs:581-596)
template <class C>
ECCX_DEV void fe_load_be(Fe<C::L>& r, const uint8_t* __restrict__ in) {
  if constexpr (C::FB % 4 == 0 && C::FB == 4 * C::L) {
    uint32_t w[C::L];
    if (io_words_load<C::L>(w, in)) {
#pragma unroll
      for (int i = 0; i < C::L; ++i) r.v[i] = __builtin_bswap32(w[C::L - 1 - i]);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int sig = 4 * i + b;  // byte significance
      if (sig < C::FB) w |= (uint32_t)in[C::FB - 1 - sig] << (8 * b);
    }
    r.v[i] = w;
  }
}
template <class C>
ECCX_DEV void fe_store_be(uint8_t* __restrict__ out, const Fe<C::L>& a) {
  if constexpr (C::FB % 4 == 0 && C::FB == 4 * C::L) {
    uint32_t w[C::L];
#pragma unroll
    for (int i = 0; i < C::L; ++i) w[i] = __builtin_bswap32(a.v[C::L - 1 - i]);
    if (io_words_store<C::L>(out, w)) return;
  }
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int sig = 4 * i + b;
      if (sig < C::FB) out[C::FB - 1 - sig] = (uint8_t)(a.v[i] >> (8 * b));
    }
  }
}
template <class C>
ECCX_DEV void fe_load_le(Fe<C::L>& r, const uint8_t* __restrict__ in) {
  if constexpr (C::FB % 4 == 0 && C::FB == 4 * C::L) {
    if (io_words_load<C::L>(r.v, in)) return;
  }
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int sig = 4 * i + b;
      if (sig < C::FB) w |= (uint32_t)in[sig] << (8 * b);
    }
    r.v[i] = w;
  }
}
template <class C>
ECCX_DEV void fe_store_le(uint8_t* __restrict__ out, const Fe<C::L>& a) {
  if constexpr (C::FB % 4 == 0 && C::FB == 4 * C::L) {
    if (io_words_store<C::L>(out, a.v)) return;
  }
#pragma unroll
  for (int i = 0; i < C::L; ++i) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int sig = 4 * i + b;
      if (sig < C::FB) out[sig] = (uint8_t)(a.v[i] >> (8 * b));
    }
  }
}

// value < P ?   (canonical-encoding check, field_macros.rs:604-627)
template <class C>
ECCX_DEV bool fe_is_canonical(const Fe<C::L>& a) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < C::L; ++i) (void)subb(a.v[i], C::P[i], bw);
  return bw != 0;
}

}  // namespace eccx
