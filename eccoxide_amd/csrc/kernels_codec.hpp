// Point wire formats on the device: compressed encodings in and out of the
// affine x||y records the scalar-multiplication kernels take and produce.
//
//   p256r1 / p384r1 / p521r1   SEC1 compressed, FB + 1 bytes: 0x02 | (y odd) followed by x
//                              big-endian; 0x00 followed by zeros = the point at infinity.
//                              The reference exposes (x, Sign) (src/curve/affine.rs:23-58,
//                              fiat/curve_macros.rs:211-223) with Sign::Negative = low bit of
//                              the canonical y (fiat/field_macros.rs:557-565): the prefix byte
//                              is that sign.
//   bls12_381_g1               zcash compressed, FB bytes: x big-endian, bit 7 of the first
//                              byte = compressed, bit 6 = infinity, bit 5 = y is the larger of
//                              the two roots (src/curve/bls12_381/serialize.rs:52-62,103-127,
//                              143-148,181-202,253-262).
//   edwards25519               RFC 8032, 32 bytes: y little-endian, bit 255 = low bit of x
//                              (src/protocol/ed25519.rs:27-59, curve25519.rs:765-784).
//
// Decompression needs one square root per point: a^((p+1)/4) for the four p = 3 (mod 4) fields
// (sec2/p256r1.rs:68-84, p384r1.rs:71, p521r1.rs:126-131, bls12_381/fp.rs:64-68) and the
// u/v form u v^3 (u v^7)^((p-5)/8) for 2^255 - 19 (curve25519.rs:246-265).  The exponents come
// from curve_consts.inc as runs of ones and zeros: x^(2^k - 1) is built by doubling k, so the
// Solinas and Mersenne exponents cost their bit length in squarings plus a dozen products;
// the irregular BLS12-381 exponent runs through 2-bit windows.  A root is unique up to sign
// and the sign rule is part of each format, so any correct root yields the reference's bytes.
#pragma once
#include "kernels_unsat.hpp"

namespace eccx {

enum : uint8_t { CODEC_OK = 0, CODEC_INFINITY = 1, CODEC_INVALID = 2 };
enum : int { FORMAT_SEC1 = 0, FORMAT_ZCASH = 1, FORMAT_RFC8032 = 2, FORMAT_ZCASH_RAW = 3 };

template <class CU>
using UT = U<CU, 1, 3>;

template <class CU, int K1, int V1, int K2, int V2>
ECCX_DEV UT<CU> ut_mul(const U<CU, K1, V1>& a, const U<CU, K2, V2>& b) {
  return u_fit<1, 3>(u_mul(a, b));
}

// a^(2^n), one multiplier body however large n is
template <class CU>
ECCX_DEV UT<CU> ut_sqr_n(UT<CU> a, int n) {
#pragma nounroll
  for (int i = 0; i < n; ++i) a = u_fit<1, 3>(u_sqr(a));
  return a;
}

// x^(2^K - 1): e(2m) = e(m)^(2^m) e(m), e(2m + 1) = e(2m)^2 x
template <int K, class CU>
ECCX_DEV UT<CU> ut_pow_run(const UT<CU>& x) {
  if constexpr (K == 1) {
    return x;
  } else if constexpr (K % 2 == 0) {
    const UT<CU> h = ut_pow_run<K / 2, CU>(x);
    return ut_mul(ut_sqr_n<CU>(h, K / 2), h);
  } else {
    const UT<CU> h = ut_pow_run<K - 1, CU>(x);
    return ut_mul(ut_sqr_n<CU>(h, 1), x);
  }
}

// x^ROOT over the runs of its binary expansion (segment S onwards)
template <class CU, int S>
ECCX_DEV UT<CU> ut_root_chain(const UT<CU>& x, UT<CU> acc) {
  if constexpr (S == CU::ROOT_SEGS) {
    return acc;
  } else {
    constexpr int ones = CU::ROOT_ONES[S], zeros = CU::ROOT_ZEROS[S];
    const UT<CU> run = ut_pow_run<ones, CU>(x);
    if constexpr (S == 0) acc = run;
    else acc = ut_mul(ut_sqr_n<CU>(acc, ones), run);
    acc = ut_sqr_n<CU>(acc, zeros);
    return ut_root_chain<CU, S + 1>(x, acc);
  }
}

// x^ROOT through 2-bit windows of the exponent words (every lane runs the same exponent, so the
// digit tests are wave-uniform)
template <class CU>
ECCX_DEV UT<CU> ut_root_windows(const UT<CU>& x) {
  constexpr int NWIN = (CU::ROOT_BITS + 1) / 2;
  const UT<CU> x2 = u_fit<1, 3>(u_sqr(x));
  const UT<CU> x3 = ut_mul(x2, x);
  auto digit = [](int w) { return (CU::ROOT_EXP[w >> 4] >> (2 * (w & 15))) & 3u; };
  auto pick = [&](uint32_t d) {
    UT<CU> m;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) m.v[i] = d == 1 ? x.v[i] : (d == 2 ? x2.v[i] : x3.v[i]);
    return m;
  };
  static_assert(((CU::ROOT_EXP[(NWIN - 1) >> 4] >> (2 * ((NWIN - 1) & 15))) & 3u) != 0, "top window holds the top bit");
  UT<CU> acc = pick(digit(NWIN - 1));
#pragma nounroll
  for (int w = NWIN - 2; w >= 0; --w) {
    acc = ut_sqr_n<CU>(acc, 2);
    const uint32_t d = digit(w);
    if (d != 0) acc = ut_mul(acc, pick(d));
  }
  return acc;
}

template <class CU>
ECCX_DEV UT<CU> ut_root_pow(const UT<CU>& x) {
  if constexpr (CU::ROOT_CHAIN != 0) return ut_root_chain<CU, 0>(x, x);
  else return ut_root_windows<CU>(x);
}

// a == b (mod p)
template <class CU, int K1, int V1, int K2, int V2>
ECCX_DEV bool ut_equal(const U<CU, K1, V1>& a, const U<CU, K2, V2>& b) {
  return u_is_zero_mod_p(u_reduce(u_sub(a, b)));
}

// canonical integers: r = (a == 0) ? 0 : p - a
template <class CS>
ECCX_DEV void fe_neg_canonical(Fe<CS::L>& r, const Fe<CS::L>& a) {
  uint32_t bw = 0, any = 0;
#pragma unroll
  for (int i = 0; i < CS::L; ++i) {
    r.v[i] = subb(CS::P[i], a.v[i], bw);
    any |= a.v[i];
  }
  if (any == 0) {
#pragma unroll
    for (int i = 0; i < CS::L; ++i) r.v[i] = 0;
  }
}
// a > b on plain integers
template <int L>
ECCX_DEV bool fe_greater(const Fe<L>& a, const Fe<L>& b) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) (void)subb(b.v[i], a.v[i], bw);
  return bw != 0;
}
template <int L>
ECCX_DEV bool fe_all_zero(const Fe<L>& a) {
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) any |= a.v[i];
  return any == 0;
}

// x^3 + a x + b for a = -3 or a = 0 (working form)
template <class CU>
ECCX_DEV UT<CU> ut_curve_rhs(const UT<CU>& x) {
  UT<CU> cb;
#pragma unroll
  for (int k = 0; k < CU::N; ++k) cb.v[k] = CU::CB[k];
  const UT<CU> x3 = ut_mul(u_sqr(x), x);
  if constexpr (CU::Sat::A0) return u_fit<1, 3>(u_reduce(u_add(x3, cb)));
  else return u_fit<1, 3>(u_reduce(u_add(u_sub(x3, u_add(u_add(x, x), x)), cb)));
}

template <class CS, int FORMAT>
constexpr int enc_bytes() { return FORMAT == FORMAT_SEC1 ? CS::FB + 1 : (FORMAT == FORMAT_ZCASH_RAW ? 2 * CS::FB : CS::FB); }

// enc -> x||y (big-endian), flags: 0 point, 1 infinity encoding, 2 rejected (bad prefix or
// flag bits, x not below p, x^3 + a x + b not a square).  Rejected and infinity records leave
// zeros in out.  affine::Point::decompress (affine.rs:48-58): y = sqrt(x^3 + a x + b), negated
// when its sign is not the requested one; zcash: the root whose is_largest equals the sort flag
// (serialize.rs:181-202), without the subgroup check of from_compressed (the C ABI adds it on
// request: from_compressed_oncurve_only vs from_compressed, serialize.rs:299-335).
template <class CU, int FORMAT>
__global__ void __launch_bounds__(WG) k_point_decompress(size_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out,
                                                         uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int EB = enc_bytes<CS, FORMAT>();
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const uint8_t* __restrict__ e = enc + i * (size_t)EB;
    uint8_t status = CODEC_OK;
    bool want = false;  // SEC1: y odd; zcash: y is the larger root
    Fe<L> rx;
    const uint32_t b0 = e[0];
    if constexpr (FORMAT == FORMAT_SEC1) {
      fe_load_be<CS>(rx, e + 1);
      if (b0 == 2u || b0 == 3u) want = (b0 & 1u) != 0;
      else status = (b0 == 0u && fe_all_zero<L>(rx)) ? CODEC_INFINITY : CODEC_INVALID;
    } else {
      static_assert(8 * FB - CS::PBITS >= 3, "the three flag bits need room above the field");
      fe_load_be<CS>(rx, e);
      rx.v[L - 1] &= ~(0xE0u << ((FB - 1) % 4 * 8));  // the flags share the leading byte with x
      if ((b0 & 0x80u) == 0) status = CODEC_INVALID;  // not the compressed flavour
      else if (b0 & 0x40u) status = ((b0 & 0x20u) == 0 && fe_all_zero<L>(rx)) ? CODEC_INFINITY : CODEC_INVALID;
      else want = (b0 & 0x20u) != 0;
    }
    if (status == CODEC_OK && !fe_is_canonical<CS>(rx)) status = CODEC_INVALID;
    // the arithmetic runs on every lane (a rejected x is some integer below 2^(8 FB))
    const UT<CU> x = u_as<1, 3>(u_to_mont<CU>(rx));
    const UT<CU> rhs = ut_curve_rhs<CU>(x);
    const UT<CU> r = ut_root_pow<CU>(rhs);
    if (status == CODEC_OK && !ut_equal(u_sqr(r), rhs)) status = CODEC_INVALID;
    Fe<L> y, yn;
    u_to_canonical<CU>(y, r);
    fe_neg_canonical<CS>(yn, y);
    bool have;
    if constexpr (FORMAT == FORMAT_SEC1) have = (y.v[0] & 1u) != 0;
    else have = fe_greater<L>(y, yn);  // y > p - y  <=>  y > (p - 1) / 2
    if (have != want) y = yn;
    if (status != CODEC_OK) {
#pragma unroll
      for (int k = 0; k < L; ++k) { rx.v[k] = 0; y.v[k] = 0; }
    }
    fe_store_be<CS>(out + i * (size_t)(2 * FB), rx);
    fe_store_be<CS>(out + i * (size_t)(2 * FB) + FB, y);
    flags[i] = status;
  }
}

// block-wide copy between global memory and LDS: 16 bytes per lane when both sides are aligned
ECCX_DEV void stage_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t bytes) {
  size_t done = 0;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
    const size_t quads = bytes >> 4;
    for (size_t i = threadIdx.x; i < quads; i += WG) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    done = quads << 4;
  }
  for (size_t i = done + threadIdx.x; i < bytes; i += WG) dst[i] = src[i];
}

// x||y (+ optional infinity flags) -> enc.  Coordinates are taken as canonical (what every
// kernel of this library emits and what the reference's types guarantee).  Pure byte movement
// (HBM-bound): a workgroup's records are contiguous on both sides, so they are staged through
// LDS with 16-byte accesses and only the LDS side sees the odd record sizes (33 / 49 / 67 bytes).
template <class CS, int FORMAT>
__global__ void __launch_bounds__(WG) k_point_compress(size_t n, const uint8_t* __restrict__ xy, const uint8_t* __restrict__ inf,
                                                       uint8_t* __restrict__ out) {
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int PB = 2 * FB;
  constexpr int EB = enc_bytes<CS, FORMAT>();
  __shared__ __align__(16) uint8_t s_in[WG * PB];
  __shared__ __align__(16) uint8_t s_out[WG * EB];
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t cnt = n - base < (size_t)WG ? n - base : (size_t)WG;
    stage_copy(s_in, xy + base * PB, cnt * PB);
    __syncthreads();
    if (threadIdx.x < cnt) {
      const uint8_t* p = s_in + threadIdx.x * PB;
      uint8_t* o = s_out + threadIdx.x * EB;
      const bool is_inf = inf != nullptr && inf[base + threadIdx.x] != 0;
      if constexpr (FORMAT == FORMAT_SEC1) {
        o[0] = is_inf ? 0u : (uint8_t)(2u | (p[PB - 1] & 1u));
#pragma unroll
        for (int k = 0; k < FB; ++k) o[1 + k] = is_inf ? 0u : p[k];
      } else if constexpr (FORMAT == FORMAT_ZCASH) {
        Fe<L> y, yn;
        fe_load_be<CS>(y, p + FB);
        fe_neg_canonical<CS>(yn, y);
        const uint8_t fl = is_inf ? 0xC0u : (uint8_t)(0x80u | (fe_greater<L>(y, yn) ? 0x20u : 0u));
        o[0] = (uint8_t)((is_inf ? 0u : p[0]) | fl);
#pragma unroll
        for (int k = 1; k < FB; ++k) o[k] = is_inf ? 0u : p[k];
      } else if constexpr (FORMAT == FORMAT_ZCASH_RAW) {
        // to_uncompressed (serialize.rs:269-277): x||y as they stand, all flags clear; the identity
        // is the infinity flag alone (:98-100)
        o[0] = is_inf ? 0x40u : p[0];
#pragma unroll
        for (int k = 1; k < PB; ++k) o[k] = is_inf ? 0u : p[k];
      } else {
        // y little-endian with the low bit of x in bit 255 (encode_point, ed25519.rs:27-36); the
        // identity is the ordinary point (0, 1), so there is no infinity record
        static_assert(FB % 4 == 0, "word copies");
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(p);
        uint32_t* ow = reinterpret_cast<uint32_t*>(o);
#pragma unroll
        for (int k = 0; k < FB / 4; ++k) ow[k] = pw[FB / 4 + k] | (k == FB / 4 - 1 ? (pw[0] & 1u) << 31 : 0u);
      }
    }
    __syncthreads();
    stage_copy(out + base * EB, s_out, cnt * EB);
    __syncthreads();
  }
}

// zcash uncompressed flavour -> x||y (from_uncompressed_oncurve_only, serialize.rs:371-383:
// read_uncompressed_flags :129-141, read_uncompressed_affine :207-224): the compression and sort
// bits must be clear, bit 6 marks the identity (all else zero), both coordinates below p and on
// the curve.  Flags as for k_point_decompress.
template <class CU>
__global__ void __launch_bounds__(WG) k_point_from_uncompressed(size_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out,
                                                                uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  static_assert(8 * FB - CS::PBITS >= 3, "the three flag bits need room above the field");
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const uint8_t* __restrict__ e = enc + i * (size_t)(2 * FB);
    const uint32_t b0 = e[0];
    Fe<L> rx, ry;
    fe_load_be<CS>(rx, e);
    fe_load_be<CS>(ry, e + FB);
    rx.v[L - 1] &= ~(0xE0u << ((FB - 1) % 4 * 8));
    uint8_t status = CODEC_OK;
    if (b0 & 0xA0u) status = CODEC_INVALID;
    else if (b0 & 0x40u) status = (fe_all_zero<L>(rx) && fe_all_zero<L>(ry)) ? CODEC_INFINITY : CODEC_INVALID;
    if (status == CODEC_OK && !(fe_is_canonical<CS>(rx) && fe_is_canonical<CS>(ry))) status = CODEC_INVALID;
    const UT<CU> x = u_as<1, 3>(u_to_mont<CU>(rx));
    const UT<CU> y = u_as<1, 3>(u_to_mont<CU>(ry));
    if (status == CODEC_OK && !ut_equal(u_sqr(y), ut_curve_rhs<CU>(x))) status = CODEC_INVALID;
    if (status != CODEC_OK) {
#pragma unroll
      for (int k = 0; k < L; ++k) { rx.v[k] = 0; ry.v[k] = 0; }
    }
    fe_store_be<CS>(out + i * (size_t)(2 * FB), rx);
    fe_store_be<CS>(out + i * (size_t)(2 * FB) + FB, ry);
    flags[i] = status;
  }
}

// edwards25519, RFC 8032 (protocol/ed25519.rs:38-59 decode_point, curve25519.rs:772-784
// decompress, :246-265 sqrt_div): flags 0 point, 2 rejected (y not below p, x = 0 with the sign
// bit set, (y^2 - 1) / (d y^2 + 1) not a square).  out = x||y little-endian.
template <class CU>
__global__ void __launch_bounds__(WG) k_ed_point_decompress(size_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out,
                                                            uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  static_assert(CU::KIND == UK_PM19 && L == 8, "written for 2^255 - 19");
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const uint8_t* __restrict__ e = enc + i * 32;
    Fe<L> ry;
    fe_load_le<CS>(ry, e);
    const bool want = (ry.v[L - 1] >> 31) != 0;  // low bit of x
    ry.v[L - 1] &= 0x7FFFFFFFu;
    uint8_t status = fe_is_canonical<CS>(ry) ? CODEC_OK : CODEC_INVALID;
    UT<CU> one, d, im;
#pragma unroll
    for (int k = 0; k < CU::N; ++k) { one.v[k] = CU::ONE[k]; d.v[k] = CU::D[k]; im.v[k] = CU::SQRT_M1[k]; }
    const UT<CU> y = u_as<1, 3>(u_from_sat<CU>(ry));
    const UT<CU> yy = u_fit<1, 3>(u_sqr(y));
    const UT<CU> u = u_fit<1, 3>(u_reduce(u_sub(yy, one)));
    const UT<CU> v = u_fit<1, 3>(u_reduce(u_add(ut_mul(d, yy), one)));
    // x = 0 exactly for y = +-1, i.e. u = 0: that encoding must have a clear sign bit
    if (want && u_is_zero_mod_p(u)) status = CODEC_INVALID;
    const UT<CU> v3 = ut_mul(u_sqr(v), v);
    const UT<CU> v7 = ut_mul(u_sqr(v3), v);
    UT<CU> r = ut_mul(ut_mul(u, v3), ut_root_pow<CU>(ut_mul(u, v7)));
    const UT<CU> check = ut_mul(v, u_sqr(r));
    const bool correct = ut_equal(check, u);
    const bool flipped = u_is_zero_mod_p(u_reduce(u_add(check, u)));
    if (flipped) r = ut_mul(r, im);  // v r^2 = -u: i r is the root
    if (!(correct || flipped)) status = CODEC_INVALID;
    Fe<L> x, xn;
    u_to_canonical<CU>(x, r);
    fe_neg_canonical<CS>(xn, x);
    if (((x.v[0] & 1u) != 0) != want) x = xn;
    if (status != CODEC_OK) {
#pragma unroll
      for (int k = 0; k < L; ++k) { x.v[k] = 0; ry.v[k] = 0; }
    }
    fe_store_le<CS>(out + i * 64, x);
    fe_store_le<CS>(out + i * 64 + 32, ry);
    flags[i] = status;
  }
}

}  // namespace eccx
