// Secret-scalar kernels (ECCX_CT_SCAN) on the unsaturated field: fixed base.
//
// The reference's Point::mul_base is constant-time: select_from_table reads every entry of the
// window's table and keeps one by masks (src/curve/projective.rs:427-434, :965-981;
// src/curve/curve25519.rs:840-869).  These kernels keep that discipline -- no memory address and no
// branch depends on a scalar digit -- on the fast field layer (ufe.hpp):
//
//   table     signed W-bit windows (Booth): entry (w, d) = d * 2^(W w) * G for d = 1 .. 2^(W-1), affine
//             (Weierstrass: x, y; edwards25519: y - x, y + x, 2d x y), one contiguous slice per window.
//             k*G does not depend on the window width: W = 6 gives 43 additions where the reference's
//             4-bit comb makes 64.
//   lookup    the window's slice is the same for every lane, so the WORKGROUP copies it into LDS once
//             (double-buffered, one barrier per window) and every lane reads every entry from there by
//             broadcast ds_read_b128 -- uniform addresses -- and keeps the entry of its own digit with
//             v_cndmask.  The sign of the digit is applied by selects as well.
//   addition  edwards25519: the complete 7-product addition (ued_add_niels), no special cases exist.
//             Weierstrass: mixed addition in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2;
//             madd-2008-s: 8 products + 2 squares -- a comb never doubles, so the Jacobian Z is never
//             needed as such).  The formula is not complete; what it misses is handled by selects only:
//               accumulator at infinity -> the entry;  digit 0 -> keep the accumulator;
//               accumulator == -entry -> infinity;     accumulator == entry -> 2 * entry, computed from the
//               entry's AFFINE coordinates (3 products + 3 squares, only in the windows named below).
//             Windows are added LOW to HIGH, so before window w the accumulator is s*G with
//             |s| < 2^(W w) * 2^(W-1) / (2^W - 1) and the entry is e*G with 2^(W w) <= |e| <= 2^(W w + W - 1):
//             s = +-e (mod n) needs |s| + |e| >= n, impossible while W (w + 1) <= NBITS - 1 (n >= 2^(NBITS-1)).
//             Only the top ct_unsafe_windows() windows run the two collision selects (for ANY scalar
//             string, canonical or not); tests/test_ct_model.py checks the bound by enumeration on
//             small parameters.
//
//   gather    (GATHER = true, ECCX_CT_GATHER) the lookup as a cross-lane move instead of a scan: lane l of every
//             wavefront loads entry l of the window's slice -- an address that depends on the lane number only --
//             and each lane then fetches the words of "its" entry from lane d - 1 with ds_bpermute_b32, which
//             routes registers through the LDS crossbar and touches no memory.  No address and no branch depends
//             on a digit; what the digit steers is the crossbar.  Its time was measured the same for identity,
//             random, all-equal and bank-folding index patterns (97-100 cycles per four permutes either way,
//             profiles/r03_select_rates.jsonl), but that is a measurement, not an architectural guarantee: the
//             strict scan stays the default of ECCX_CT_SCAN and this form is opt-in.  A permute costs no VALU
//             slot and the same whatever the number of entries, so the windows are 7 bits wide (64 entries, one
//             per lane; 37 additions for 256 bits).
//
// Everything a lane does is the same instruction stream whatever its scalar; the only branches are loop
// counters and the bounds of the batch.  tools/isa_histogram.py --branches lists them per kernel.
#pragma once
#include "kernels_unsat.hpp"

namespace eccx {

// Window width of the scanning comb.  Measured per 2^20 units (same box, profiles/r03_ab_ct.txt): p256r1 W = 4 / 5 / 6:
// 3.96 / 3.50 / 3.30 ms, ed25519 2.66 / 2.46 / 2.58 ms -- the scan costs 4.5-5 cycles per word and entry
// (v_cndmask_b32_e64), the Edwards addition is cheaper and its entries are three coordinates wide.
#ifndef ECCX_CT_BASE_BITS
#define ECCX_CT_BASE_BITS 6
#endif
#ifndef ECCX_CT_BASE_BITS_ED
#define ECCX_CT_BASE_BITS_ED 5
#endif
#ifndef ECCX_CT_BASE_OCC9
#define ECCX_CT_BASE_OCC9 4
#endif
#ifndef ECCX_CT_SCAN_SMEM
#define ECCX_CT_SCAN_SMEM 0  // 1: the scan reads the slice through the scalar cache (ct_scan_smem) instead of an LDS copy
#endif
#ifndef ECCX_CT_SCAN_PK
#define ECCX_CT_SCAN_PK 1  // the LDS scan keeps two words per instruction (ct_scan_lds_pk); 0: one v_cndmask_b32 per word
#endif
#ifndef ECCX_CT_GATHER_BITS
#define ECCX_CT_GATHER_BITS 7
#endif
template <class CU, bool GATHER = false>
constexpr int ct_base_bits() { return GATHER ? ECCX_CT_GATHER_BITS : (CU::KIND == UK_PM19 ? ECCX_CT_BASE_BITS_ED : ECCX_CT_BASE_BITS); }
// waves per SIMD the comb is compiled for (the XYZZ accumulator is four coordinates)
template <class CU>
constexpr int ct_base_occupancy() { return CU::N <= 9 ? ECCX_CT_BASE_OCC9 : unsat_occupancy<CU>(); }
template <class CU, bool GATHER = false>
constexpr int ct_base_windows() { return (8 * CU::Sat::SB + 1 + ct_base_bits<CU, GATHER>() - 1) / ct_base_bits<CU, GATHER>(); }
template <class CU, bool GATHER = false>
constexpr int ct_base_entries() { return 1 << (ct_base_bits<CU, GATHER>() - 1); }
// words per table entry, padded to 16 bytes: Weierstrass (x, y), edwards25519 (y - x, y + x, 2d x y)
template <class CU>
constexpr int ct_entry_words() { return (((CU::KIND == UK_PM19 ? 3 : 2) * CU::N + 3) / 4) * 4; }
// windows (counted from the top) in which accumulator == +-entry is possible: those with W (w + 1) >= NBITS
template <class CU, bool GATHER = false>
constexpr int ct_unsafe_windows() {
  constexpr int W = ct_base_bits<CU, GATHER>();
  return ct_base_windows<CU, GATHER>() - ((CU::Sat::NBITS + W - 1) / W) + 1;
}

// The gather form of the lookup: lane l holds entry (l mod ENT) of the window's slice (loaded from an address
// that depends on the lane number only); every lane takes the EW words of entry d from lane d - 1 -- in its own
// half of the wavefront when two copies of a 32-entry slice fit the 64 lanes.  Digit 0 reads some lane's entry;
// the caller discards it.
template <int EW, int ENT>
ECCX_DEV void ct_gather_lanes(uint32_t (&out)[EW], const uint4* __restrict__ slice, uint32_t d) {
  static_assert(EW % 4 == 0 && (ENT == 64 || ENT == 32 || ENT == 16), "one entry per lane, whole copies per wavefront");
  const uint32_t lane = threadIdx.x & 63u;
  uint4 v[EW / 4];
#pragma unroll
  for (int c = 0; c < EW / 4; ++c) v[c] = slice[(lane & (uint32_t)(ENT - 1)) * (EW / 4) + c];
  const int addr = (int)((((d - 1u) & (uint32_t)(ENT - 1)) | (lane & (uint32_t)(64 - ENT) & 63u)) << 2);
#pragma unroll
  for (int c = 0; c < EW / 4; ++c) {
    out[4 * c] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[c].x);
    out[4 * c + 1] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[c].y);
    out[4 * c + 2] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[c].z);
    out[4 * c + 3] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[c].w);
  }
}

// select_from_table over a slice held in LDS: every lane reads entries 1 .. ENT (uniform addresses)
// and keeps entry d; out is untouched for d = 0 (or any d outside the slice)
template <int EW, int ENT>
ECCX_DEV void ct_scan_lds(uint32_t (&out)[EW], const uint4* __restrict__ slice, uint32_t d) {
  static_assert(EW % 4 == 0, "entries are padded to 16 bytes");
#pragma unroll 2
  for (int j = 0; j < ENT; ++j) {
    const uint64_t m = __builtin_amdgcn_uicmp(d, (uint32_t)(j + 1), 32 /* ICMP_EQ */);
    uint4 v[EW / 4];  // the whole entry first: its reads are in flight together
#pragma unroll
    for (int c = 0; c < EW / 4; ++c) v[c] = slice[j * (EW / 4) + c];
#pragma unroll
    for (int c = 0; c < EW / 4; ++c) ct_cmov4(out[4 * c], out[4 * c + 1], out[4 * c + 2], out[4 * c + 3], v[c].x, v[c].y, v[c].z, v[c].w, m);
  }
}

// The same scan at two words per instruction.  A limb (< 2^30) with bit 30 set is the bit pattern of a positive, normal
// binary32 number (exponent field 10xxxxxx), so e * 1.0 + (+0) = e, e * 0.0 + acc = acc EXACTLY, whatever the rounding
// and denormal modes: v_pk_fma_f32 against the lane's (1.0, 1.0) / (0.0, 0.0) keeps or drops two words at once, at the
// issue cost of one v_cndmask_b32.  No NaN, infinity or denormal can arise (every input is normal or +0, one factor
// is 0 or 1), every lane executes every instruction, and the unit's latency does not depend on its operands.
// The slice in LDS holds the words tagged (CtSliceStage::store<true>); the tag is masked off the selected entry.
// Digit 0 leaves +0 everywhere: out = 0.
constexpr uint32_t CT_F32_TAG = 0x40000000u;
ECCX_DEV void ct_fsel2(uint64_t& acc, uint64_t e, uint64_t m) {  // both halves take the LOW word of m (op_sel_hi)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(e), "v"(m));
}
template <int EW, int ENT, int USED, int B>
ECCX_DEV void ct_scan_lds_pk(uint32_t (&out)[EW], const uint4* __restrict__ slice, uint32_t d) {
  // limbs below 2^29, a top limb below 3 * 2^28: bits 29..23 are never all ones, so the tagged word is never infinity / NaN
  static_assert(EW % 4 == 0 && USED <= EW && B <= 29, "entries are padded to 16 bytes; limbs leave bit 30 free");
  constexpr int PAIRS = (USED + 1) / 2;
  uint64_t acc[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) acc[p] = 0;  // (+0.0, +0.0)
#pragma unroll 2
  for (int j = 0; j < ENT; ++j) {
    uint32_t m;  // 1.0f for the lane whose digit is j + 1, else 0.0f
    asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(m) : "s"(__builtin_amdgcn_uicmp(d, (uint32_t)(j + 1), 32 /* ICMP_EQ */)));
    const uint64_t m2 = m;  // the upper word is not read
    uint4 v[EW / 4];
#pragma unroll
    for (int c = 0; c < EW / 4; ++c) v[c] = slice[j * (EW / 4) + c];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const uint4 q = v[p / 2];
      ct_fsel2(acc[p], (p & 1) ? (((uint64_t)q.w << 32) | q.z) : (((uint64_t)q.y << 32) | q.x), m2);
    }
  }
#pragma unroll
  for (int k = 0; k < EW; ++k) out[k] = k < 2 * PAIRS ? ((uint32_t)(acc[k / 2] >> (32 * (k & 1))) & ~CT_F32_TAG) : 0u;
}

// select_from_table with the entries in SCALAR registers: the slice address is the same for every lane, so the
// entry words arrive by s_load through the scalar cache and each costs one v_and_or_b32 against the lane's
// all-ones / zero mask -- no copy of the slice in LDS, no barrier per window.  out must be zero on entry.
template <int EW, int ENT>
ECCX_DEV void ct_scan_smem(uint32_t (&out)[EW], const uint4* __restrict__ slice, uint32_t d) {
  static_assert(EW % 4 == 0, "entries are padded to 16 bytes");
#pragma unroll 2
  for (int j = 0; j < ENT; ++j) {
    uint32_t mask = 0;
    ct_cmov1(mask, 0xffffffffu, __builtin_amdgcn_uicmp(d, (uint32_t)(j + 1), 32 /* ICMP_EQ */));
    uint4 v[EW / 4];
#pragma unroll
    for (int c = 0; c < EW / 4; ++c) v[c] = slice[j * (EW / 4) + c];
#pragma unroll
    for (int c = 0; c < EW / 4; ++c) {
      asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(out[4 * c]) : "s"(v[c].x), "v"(mask));
      asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(out[4 * c + 1]) : "s"(v[c].y), "v"(mask));
      asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(out[4 * c + 2]) : "s"(v[c].z), "v"(mask));
      asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(out[4 * c + 3]) : "s"(v[c].w), "v"(mask));
    }
  }
}

// The workgroup's copy of one window slice, global -> registers (issued early) -> LDS (written late)
template <int SLICE4>
struct CtSliceStage {
  static constexpr int PER = (SLICE4 + WG - 1) / WG;
  uint4 r[PER];
  ECCX_DEV void load(const uint4* __restrict__ src) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = i * WG + (int)threadIdx.x;
      r[i] = src[at < SLICE4 ? at : SLICE4 - 1];  // clamped: uniform trip count, no out-of-bounds read
    }
  }
  template <bool TAGGED = false>  // TAGGED: every word | CT_F32_TAG, for ct_scan_lds_pk
  ECCX_DEV void store(uint4* __restrict__ dst) const {
    constexpr uint32_t T = TAGGED ? CT_F32_TAG : 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int at = i * WG + (int)threadIdx.x;
      if (at < SLICE4) dst[at] = make_uint4(r[i].x | T, r[i].y | T, r[i].z | T, r[i].w | T);
    }
  }
};

// ---- Weierstrass: XYZZ accumulator -------------------------------------------------------------------
template <class CU>
struct UXyzz {
  U<CU, 1, 3> x, y, zz, zzz;  // infinity <=> every limb of zz is zero
};

// r = p + (x2, y2) (madd-2008-s); h_zero / r_zero: the two differences vanish (mod p) -- reported only,
// the caller decides what a collision means.  WITH_TESTS = false skips the two comparisons.
template <class CU, bool WITH_TESTS>
ECCX_DEV void uxyzz_madd(UXyzz<CU>& r, bool& h_zero, bool& r_zero, const UXyzz<CU>& p, const U<CU, 1, 3>& x2,
                         const U<CU, 1, 3>& y2) {
  auto u2 = u_mul(x2, p.zz);
  auto s2 = u_mul(y2, p.zzz);
  auto h = u_reduce(u_sub(u2, p.x));   // P
  auto rr = u_reduce(u_sub(s2, p.y));  // R
  if constexpr (WITH_TESTS) {
    h_zero = u_is_zero_mod_p_ct(h);
    r_zero = u_is_zero_mod_p_ct(rr);
  } else {
    h_zero = false;
    r_zero = false;
  }
  auto hh = u_sqr(h);          // PP
  auto hhh = u_mul(h, hh);     // PPP
  auto v = u_mul(p.x, hh);     // Q
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
  r.zz = u_fit<1, 3>(u_mul(p.zz, hh));
  r.zzz = u_fit<1, 3>(u_mul(p.zzz, hhh));
}

// r = 2 * (x, y) from affine coordinates (mdbl-2008-s-1): 3 products + 3 squares
template <class CU>
ECCX_DEV void uxyzz_dbl_affine(UXyzz<CU>& r, const U<CU, 1, 3>& x, const U<CU, 1, 3>& y) {
  auto u = u_reduce(u_add(y, y));   // 2 y
  auto v = u_sqr(u);                // ZZ3
  auto w = u_mul(u, v);             // ZZZ3
  auto s = u_mul(x, v);
  auto xx = u_sqr(x);
  U<CU, 1, 3> m;
  if constexpr (CU::Sat::A0) {
    m = u_reduce(u_add(u_add(xx, xx), xx));  // 3 x^2
  } else {
    U<CU, 1, 2> one;
#pragma unroll
    for (int i = 0; i < CU::N; ++i) one.v[i] = CU::ONE[i];
    const auto xm = u_sub(xx, one);          // a = -3: 3 (x^2 - 1)
    m = u_reduce(u_add(u_add(xm, xm), xm));
  }
  auto x3 = u_reduce(u_sub(u_sub(u_sqr(m), s), s));
  auto y3a = u_mul(m, u_sub(s, x3));
  auto wy = u_mul(w, y);
  r.x = x3;
  r.y = u_reduce(u_sub(y3a, wy));
  r.zz = u_fit<1, 3>(v);
  r.zzz = u_fit<1, 3>(w);
}

// affine points as canonical big-endian bytes x | y -> table entries (x, y) in the field's working form,
// ct_entry_words() words per entry (the padding words are zeroed); a zero record (the engine's encoding of
// infinity: the entry of a digit that cannot occur) stays zero
template <class CU>
__global__ void k_affine_to_cttable(size_t entries, const uint8_t* __restrict__ affine, uint32_t* __restrict__ table) {
  using CS = typename CU::Sat;
  constexpr int L = CS::L;
  constexpr int FB = CS::FB;
  constexpr int EW = ct_entry_words<CU>();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= entries) return;
  Fe<L> px, py;
  fe_load_be<CS>(px, affine + i * (size_t)(2 * FB));
  fe_load_be<CS>(py, affine + i * (size_t)(2 * FB) + FB);
  const auto ux = u_reduce(u_to_mont<CU>(px));  // exact digits: every limb < 2^B (the scan tags bit 30)
  const auto uy = u_reduce(u_to_mont<CU>(py));
  uint32_t* o = table + i * (size_t)EW;
#pragma unroll
  for (int k = 0; k < EW; ++k) o[k] = k < CU::N ? ux.v[k] : (k < 2 * CU::N ? uy.v[k - CU::N] : 0u);
}

// Fixed base, secret scalars (Point::mul_base, src/curve/fiat/curve_macros.rs:55-63 ->
// mul_base_table_{am3,a0}, projective.rs:945-981).  table: ct_base_windows() slices of ct_base_entries()
// entries of ct_entry_words() words; rows (X ZZ, Y ZZZ, ZZ) are Jacobian rows for
// k_batch_to_affine_unsat<NORM_JACOBIAN>.
template <class CU, bool GATHER = false>
__global__ void __launch_bounds__(WG, ct_base_occupancy<CU>()) k_scalarmul_base_ct(size_t n, const uint8_t* __restrict__ scalars,
                                                                               const uint32_t* __restrict__ table,
                                                                               uint32_t* __restrict__ rows_out,
                                                                               uint8_t* __restrict__ flags) {
  using CS = typename CU::Sat;
  constexpr int N = CU::N;
  constexpr int SB = CS::SB;
  constexpr int W = ct_base_bits<CU, GATHER>();
  constexpr int NWIN = ct_base_windows<CU, GATHER>();
  constexpr int ENT = ct_base_entries<CU, GATHER>();
  constexpr int EW = ct_entry_words<CU>();
  constexpr int SLICE4 = ENT * EW / 4;
  constexpr int UNSAFE = ct_unsafe_windows<CU, GATHER>();
  static_assert(UNSAFE >= 1 && UNSAFE <= NWIN, "window bookkeeping");
  using T = U<CU, 1, 3>;
  constexpr bool STAGED = !GATHER && !ECCX_CT_SCAN_SMEM;  // the slice goes through LDS
  __shared__ uint4 lds[STAGED ? 2 : 1][STAGED ? SLICE4 : 1];
  const uint4* __restrict__ gtab = reinterpret_cast<const uint4*>(table);
  T one;
#pragma unroll
  for (int i = 0; i < N; ++i) one.v[i] = CU::ONE[i];
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
    CtSliceStage<SLICE4> stage;
    if constexpr (STAGED) {
      stage.load(gtab);
      __syncthreads();  // the previous batch's last reads of buffer 0 are done
      stage.template store<(ECCX_CT_SCAN_PK != 0)>(lds[0]);
      __syncthreads();
    }
    UXyzz<CU> q;
    q.x = one;
    q.y = one;
    u_set_zero(q.zz);
    u_set_zero(q.zzz);
    for (int w = 0; w < NWIN; ++w) {
      if constexpr (STAGED) {
        if (w + 1 < NWIN) stage.load(gtab + (size_t)(w + 1) * SLICE4);
      }
      uint32_t d;
      bool neg;
      booth_digit<W, SB>(k, w, d, neg);
      uint32_t ew[EW];
      if constexpr (GATHER) {
        ct_gather_lanes<EW, ENT>(ew, gtab + (size_t)w * SLICE4, d);
      } else {
#pragma unroll
        for (int i = 0; i < EW; ++i) ew[i] = 0;
        if constexpr (STAGED && ECCX_CT_SCAN_PK) ct_scan_lds_pk<EW, ENT, 2 * N, CU::B>(ew, lds[w & 1], d);
        else if constexpr (STAGED) ct_scan_lds<EW, ENT>(ew, lds[w & 1], d);
        else ct_scan_smem<EW, ENT>(ew, gtab + (size_t)w * SLICE4, d);
      }
      T x2, y2;
#pragma unroll
      for (int i = 0; i < N; ++i) { x2.v[i] = ew[i]; y2.v[i] = ew[N + i]; }
      {
        U<CU, 2, 4> sy;
        u_select_ct(sy, neg, u_neg(y2), u_as<2, 4>(y2));
        y2 = u_reduce(sy);
      }
      const bool q_inf = u_limbs_all_zero(q.zz);
      const bool skip = d == 0;
      UXyzz<CU> sum;
      if (w == 0) {
        sum = q;  // nothing to add to yet: the accumulator is at infinity and the patch below makes the sum the entry
      } else if (w >= NWIN - UNSAFE) {  // loop counter: the same for every lane and every scalar
        bool hz, rz;
        uxyzz_madd<CU, true>(sum, hz, rz, q, x2, y2);
        UXyzz<CU> dbl;
        uxyzz_dbl_affine<CU>(dbl, x2, y2);
        const bool same_x = hz & !q_inf & !skip;
        const bool twice = same_x & rz;    // accumulator == entry
        const bool cancel = same_x & !rz;  // accumulator == -entry
        const uint64_t mt = ct_mask(twice), mc = ct_mask(cancel);
        u_cmov_ct(sum.x, mt, dbl.x);
        u_cmov_ct(sum.y, mt, dbl.y);
        u_cmov_ct(sum.zz, mt, dbl.zz);
        u_cmov_ct(sum.zzz, mt, dbl.zzz);
        T zero;
        u_set_zero(zero);
        u_cmov_ct(sum.zz, mc, zero);
        u_cmov_ct(sum.zzz, mc, zero);
      } else {
        bool hz, rz;
        uxyzz_madd<CU, false>(sum, hz, rz, q, x2, y2);
      }
      // accumulator at infinity: the sum is the entry itself
      const uint64_t mi = ct_mask(q_inf), mk = ct_mask(!skip);
      u_cmov_ct(sum.x, mi, x2);
      u_cmov_ct(sum.y, mi, y2);
      u_cmov_ct(sum.zz, mi, one);
      u_cmov_ct(sum.zzz, mi, one);
      u_cmov_ct(q.x, mk, sum.x);
      u_cmov_ct(q.y, mk, sum.y);
      u_cmov_ct(q.zz, mk, sum.zz);
      u_cmov_ct(q.zzz, mk, sum.zzz);
      if constexpr (STAGED) {
        if (w + 1 < NWIN) stage.template store<(ECCX_CT_SCAN_PK != 0)>(lds[(w + 1) & 1]);
        __syncthreads();
      }
    }
    if (active) {
      // (X, Y, ZZ, ZZZ) -> the Jacobian triple (X ZZ : Y ZZZ : ZZ): X ZZ / ZZ^2 = x, Y ZZZ / ZZ^3 = Y / ZZZ = y
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), u_fit<1, 3>(u_mul(q.x, q.zz)), u_fit<1, 3>(u_mul(q.y, q.zzz)), q.zz);
      flags[idx] = 0;
    }
  }
}

// ---- edwards25519 fixed base, secret scalars (curve25519.rs:840-869) -----------------------------------
// Complete additions: digit 0 adds the neutral element (1, 1, 0), which is what the scan leaves when no
// entry matches; a negative digit swaps y - x with y + x and negates 2d x y.
template <class CU, bool GATHER = false>
__global__ void __launch_bounds__(WG, 4) k_ed_scalarmul_base_ct(size_t n, const uint8_t* __restrict__ scalars,
                                                                const uint32_t* __restrict__ table,
                                                                uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags) {
  constexpr int N = CU::N;
  constexpr int W = ct_base_bits<CU, GATHER>();
  constexpr int NWIN = ct_base_windows<CU, GATHER>();
  constexpr int ENT = ct_base_entries<CU, GATHER>();
  constexpr int EW = ct_entry_words<CU>();
  constexpr int SLICE4 = ENT * EW / 4;
  using T = U<CU, 1, 3>;
  constexpr bool STAGED = !GATHER && !ECCX_CT_SCAN_SMEM;
  __shared__ uint4 lds[STAGED ? 2 : 1][STAGED ? SLICE4 : 1];
  const uint4* __restrict__ gtab = reinterpret_cast<const uint4*>(table);
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * 32;
    CtSliceStage<SLICE4> stage;
    if constexpr (STAGED) {
      stage.load(gtab);
      __syncthreads();
      stage.template store<(ECCX_CT_SCAN_PK != 0)>(lds[0]);
      __syncthreads();
    }
    T qx, qy, qz, qt;  // the neutral element (0, 1, 1, 0)
    u_set_zero(qx); u_set_zero(qy); u_set_zero(qz); u_set_zero(qt);
    qy.v[0] = 1; qz.v[0] = 1;
    for (int w = 0; w < NWIN; ++w) {
      if constexpr (STAGED) {
        if (w + 1 < NWIN) stage.load(gtab + (size_t)(w + 1) * SLICE4);
      }
      uint32_t d;
      bool neg;
      booth_digit<W, 32>(k, w, d, neg);
      uint32_t ew[EW];
      if constexpr (GATHER) {
        ct_gather_lanes<EW, ENT>(ew, gtab + (size_t)w * SLICE4, d);
        // digit 0: the neutral element (1, 1, 0) instead of whatever lane ENT - 1 holds
        const uint64_t mz = ct_mask(d == 0);
#pragma unroll
        for (int i = 0; i < 3 * N; ++i) ct_cmov1(ew[i], (i == 0 || i == N) ? 1u : 0u, mz);
      } else {
#pragma unroll
        for (int i = 0; i < EW; ++i) ew[i] = 0;
        if constexpr (STAGED && !ECCX_CT_SCAN_PK) {
          ew[0] = 1;  // y - x
          ew[N] = 1;  // y + x
          ct_scan_lds<EW, ENT>(ew, lds[w & 1], d);
        } else {
          if constexpr (STAGED) ct_scan_lds_pk<EW, ENT, 3 * N, CU::B>(ew, lds[w & 1], d);
          else ct_scan_smem<EW, ENT>(ew, gtab + (size_t)w * SLICE4, d);
          const uint64_t mz = ct_mask(d == 0);
          ct_cmov1(ew[0], 1u, mz);
          ct_cmov1(ew[N], 1u, mz);
        }
      }
      T a, b, t2d, ym, yp;
#pragma unroll
      for (int i = 0; i < N; ++i) { a.v[i] = ew[i]; b.v[i] = ew[N + i]; t2d.v[i] = ew[2 * N + i]; }
      u_select_ct(ym, neg, b, a);
      u_select_ct(yp, neg, a, b);
      U<CU, 2, 4> t2;
      u_select_ct(t2, neg, u_neg(t2d), u_as<2, 4>(t2d));
      if (w == 0) {
        // the accumulator is the neutral element: (x, y) given as (ym, yp) = (y - x, y + x) is
        // (2(yp - ym) : 2(yp + ym) : 4 : (yp - ym)(yp + ym)) -- one product instead of seven; digit 0 (1, 1, 0) gives (0 : 4 : 4 : 0)
        const auto dx = u_reduce(u_sub(yp, ym));
        const auto sy = u_reduce(u_add(yp, ym));
        qx = u_reduce(u_add(dx, dx));
        qy = u_reduce(u_add(sy, sy));
        u_set_zero(qz);
        qz.v[0] = 4;
        qt = u_fit<1, 3>(u_mul(dx, sy));
      } else {
        ued_add_niels<CU, 2, 4>(qx, qy, qz, qt, ym, yp, t2);
      }
      if constexpr (STAGED) {
        if (w + 1 < NWIN) stage.template store<(ECCX_CT_SCAN_PK != 0)>(lds[(w + 1) & 1]);
        __syncthreads();
      }
    }
    if (active) {
      u3_store<CU>(rows_out + idx * (size_t)urow3_words<CU>(), qx, qy, qz);
      flags[idx] = 0;
    }
  }
}

}  // namespace eccx
