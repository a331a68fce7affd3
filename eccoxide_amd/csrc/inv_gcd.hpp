// Modular inversion by division steps (Bernstein-Yang "safegcd") for the batched normalisation.
//
// The reference has both: Fermat chains z^(p-2) (e.g. src/curve/sec2/p256r1.rs:49-66) and a
// generic Bernstein-Yang inverse_safegcd over fiat's divstep (src/curve/fiat/field_macros.rs:
// 677-770), with a unit test that the two agree (:1203-1215); z^-1 mod p is one number.  In
// k_batch_to_affine the inversion is one sequential chain per lane with at most one wavefront
// per SIMD to hide it behind, so its length IS the kernel time.  Division steps need ~25 batches
// of (30 steps on one word + two 2x2-matrix updates of 9-limb numbers) for a 256-bit field:
// about a sixth of a Fermat chain of ~260 field multiplications.
//
// Fixed iteration count, no data-dependent control flow: the delta = 1 division step with
// floor((49 bits + 57) / 17) steps rounded up to batches of 30 (valid for every odd modulus), or -- for
// the moduli of at most 256 bits, C::INV30_HD -- the step started at delta = 1/2 ("hddivsteps"), for which
// 590 steps are proven to suffice below 2^256 (github.com/sipa/safegcd-bounds; libsecp256k1's modinv32
// runs the same 20 batches of 30): 20 batches instead of 25.
// Numbers are little-endian 30-bit limbs, all but the top one in [0, 2^30), the top one signed.
#pragma once
#include "fe.hpp"

namespace eccx {

template <class C>
struct S30 {
  int32_t v[C::INV30_N];
};

struct Trans30 {
  int32_t u, v, q, r;
};

// 30 division steps on the low words of f and g; eta = -delta (HD: eta = -(delta + 1/2), delta starting
// at 1/2).  Afterwards (f', g') = 2^-30 * (u v; q r) * (f, g).
template <bool HD>
ECCX_DEV int32_t divsteps_30(int32_t eta, uint32_t f0, uint32_t g0, Trans30& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(eta >> 31);  // all ones if delta > 0
    const uint32_t c2 = 0u - (g & 1u);    // all ones if g is odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // -f, -u, -v when delta > 0
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;                                            // swap case: delta > 0 and g odd
    if (HD) eta = (int32_t)(((uint32_t)eta ^ c1) - 1u);  // delta -> 1 - delta or 1 + delta on delta + 1/2
    else eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));   // delta -> 1 - delta or 1 + delta
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return eta;
}

// (f, g) <- 2^-30 * t * (f, g)   (exact: the low 30 bits of both combinations are zero)
template <class C>
ECCX_DEV void update_fg_30(S30<C>& f, S30<C>& g, const Trans30& t) {
  constexpr int N = C::INV30_N;
  constexpr int32_t M30 = (int32_t)0x3FFFFFFF;
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f.v[0] + v * g.v[0];
  int64_t cg = q * f.v[0] + r * g.v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < N; ++i) {
    cf += u * f.v[i] + v * g.v[i];
    cg += q * f.v[i] + r * g.v[i];
    f.v[i - 1] = (int32_t)cf & M30;
    g.v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[N - 1] = (int32_t)cf;
  g.v[N - 1] = (int32_t)cg;
}

// (d, e) <- 2^-30 * t * (d, e) mod p, staying in (-2p, p): a multiple of p is added to each
// combination so that its low 30 bits vanish
template <class C>
ECCX_DEV void update_de_30(S30<C>& d, S30<C>& e, const Trans30& t) {
  constexpr int N = C::INV30_N;
  constexpr int32_t M30 = (int32_t)0x3FFFFFFF;
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.v[N - 1] >> 31, se = e.v[N - 1] >> 31;  // all ones if negative
  int32_t md = (t.u & sd) + (t.v & se);
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = u * d.v[0] + v * e.v[0];
  int64_t ce = q * d.v[0] + r * e.v[0];
  md -= (int32_t)((C::P30_INV * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((C::P30_INV * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)C::P30[0] * md;
  ce += (int64_t)C::P30[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < N; ++i) {
    cd += u * d.v[i] + v * e.v[i];
    ce += q * d.v[i] + r * e.v[i];
    if (C::P30[i] != 0) {
      cd += (int64_t)C::P30[i] * md;
      ce += (int64_t)C::P30[i] * me;
    }
    d.v[i - 1] = (int32_t)cd & M30;
    e.v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[N - 1] = (int32_t)cd;
  e.v[N - 1] = (int32_t)ce;
}

// r in (-2p, p), negated if neg, brought to [0, p)
template <class C>
ECCX_DEV void normalize_30(S30<C>& r, int32_t neg_mask) {
  constexpr int N = C::INV30_N;
  constexpr int32_t M30 = (int32_t)0x3FFFFFFF;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int32_t add = r.v[N - 1] >> 31;  // negative: add p
#pragma unroll
    for (int i = 0; i < N; ++i) {
      r.v[i] += C::P30[i] & add;
      if (pass == 0) r.v[i] = (r.v[i] ^ neg_mask) - neg_mask;
    }
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      r.v[i + 1] += r.v[i] >> 30;
      r.v[i] &= M30;
    }
  }
}

// r = a^-1 mod p for plain (non-Montgomery) canonical a; a = 0 gives 0
template <class C>
__device__ __noinline__ void fe_inv_gcd(Fe<C::L>& r, const Fe<C::L>& a) {
  constexpr int N = C::INV30_N;
  constexpr int L = C::L;
  S30<C> d, e, f, g;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    uint64_t lo = 0;
    if (w < L) lo = a.v[w];
    if (w + 1 < L) lo |= (uint64_t)a.v[w + 1] << 32;
    g.v[i] = (int32_t)((uint32_t)(lo >> sh) & 0x3FFFFFFFu);
    f.v[i] = C::P30[i];
    d.v[i] = 0;
    e.v[i] = (i == 0) ? 1 : 0;
  }
  int32_t eta = -1;
  for (int b = 0; b < C::INV30_BATCHES; ++b) {
    Trans30 t;
    eta = divsteps_30<C::INV30_HD>(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    update_de_30<C>(d, e, t);
    update_fg_30<C>(f, g, t);
  }
  // g = 0 and f = +-1 now: d * a = f (mod p)
  normalize_30<C>(d, f.v[N - 1] >> 31);
#pragma unroll
  for (int w = 0; w < L; ++w) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int bit = 30 * i;
      if (bit + 30 > 32 * w && bit < 32 * w + 32) {
        if (bit >= 32 * w) acc |= (uint64_t)(uint32_t)d.v[i] << (bit - 32 * w);
        else acc |= (uint64_t)(uint32_t)d.v[i] >> (32 * w - bit);
      }
    }
    r.v[w] = (uint32_t)acc;
  }
}

// inversion in the field's working representation (Montgomery where the field has one)
template <class C>
ECCX_DEV void fe_inv_fast(Fe<C::L>& r, const Fe<C::L>& a) {
  if constexpr (C::MERSENNE != 0 || C::PM19 != 0) {
    fe_inv_gcd<C>(r, a);  // plain representation, canonical
  } else {
    Fe<C::L> t;
    fe_from_mont<C>(t, a);
    fe_inv_gcd<C>(t, t);
    fe_to_mont<C>(r, t);
  }
}

}  // namespace eccx
