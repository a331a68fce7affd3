// Default (fast) variable-base kernels: Jacobian ladder + batched normalisation.
//
//   k_scalarmul_var_fast<C>  same contract as k_scalarmul_var<C> (kernels.hpp) -- the
//       reference's &Point * &Scalar, src/curve/fiat/curve_macros.rs:321-327 -- but on
//       Jacobian coordinates (curve_fast.hpp) and with signed 5-bit windows (Booth recoding):
//       table d*P, d = 1..16, then per window (MSB first) 5 doublings + 1 addition of +-d*P.
//       The reference's fixed 4-bit unsigned window (src/curve/projective.rs:871-896) needs
//       2*SB additions, this needs ceil((8*SB + 1)/5); k*P is the same point either way.
//       Writes the un-normalised Jacobian result per unit; flags[i] = 2 for rejected inputs.
//   k_batch_to_affine<C>     Point::to_affine (projective.rs:655-682) for a whole batch: each
//       lane normalises U units with ONE field inversion (Montgomery's trick), so the
//       ~380-multiplication Fermat inversion is paid once per 8 units instead of per unit.
//
// Cases the Jacobian addition does not cover are patched per lane after the generic
// formulas ran: accumulator at infinity -> take the table entry; digit 0 (or an entry at
// infinity) -> keep the accumulator; equal x and opposite y -> infinity; equal points ->
// the lane keeps its accumulator and the wavefront runs one extra doubling step (the loop
// holds a single doubling body and a single addition body, like kernels.hpp).
#pragma once
#include "curve_fast.hpp"
#include "kernels.hpp"

namespace eccx {

template <int L>
constexpr int row5_words() { return ((5 * L + 3) / 4) * 4; }

// Waves per SIMD the ladder kernel is compiled for, by limb count (8: P-256; 12: P-384,
// BLS12-381; 17: P-521).  The MAC chain of a field multiplication is one long dependency
// chain, so a SIMD needs several resident waves to keep its multiplier busy (measured,
// tools/ubench/fe_bench.hip: 1470 / 1129 / 1093 / 1054 / 986 cycles per P-256 product at
// 1 / 2 / 3 / 4 / 16 waves per SIMD); the hint trades a few spilled temporaries for residency.
#ifndef ECCX_OCC_8
#define ECCX_OCC_8 4
#endif
#ifndef ECCX_OCC_12
#define ECCX_OCC_12 3
#endif
#ifndef ECCX_OCC_17
#define ECCX_OCC_17 2
#endif

constexpr int FAST_TABLE_ROWS = 17;  // entries 1..16 of the signed 5-bit window table (+ unused row 0)

template <class C>
ECCX_DEV void entry_store(uint32_t* __restrict__ row, const JacEntry<C>& p) {
  constexpr int L = C::L;
  constexpr int W = row5_words<L>();
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    w[i] = p.x.v[i]; w[L + i] = p.y.v[i]; w[2 * L + i] = p.z.v[i]; w[3 * L + i] = p.zz.v[i]; w[4 * L + i] = p.zzz.v[i];
  }
#pragma unroll
  for (int i = 5 * L; i < W; ++i) w[i] = 0;
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

template <class C>
ECCX_DEV void entry_load(JacEntry<C>& p, const uint32_t* __restrict__ row) {
  constexpr int L = C::L;
  constexpr int W = row5_words<L>();
  uint32_t w[W];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < L; ++i) {
    p.x.v[i] = w[i]; p.y.v[i] = w[L + i]; p.z.v[i] = w[2 * L + i]; p.zz.v[i] = w[3 * L + i]; p.zzz.v[i] = w[4 * L + i];
  }
}

template <class C>
__global__ void __launch_bounds__(WG, (C::L <= 8 ? ECCX_OCC_8 : (C::L <= 12 ? ECCX_OCC_12 : ECCX_OCC_17))) k_scalarmul_var_fast(size_t n, const uint8_t* __restrict__ scalars,
                                                           const uint8_t* __restrict__ points,
                                                           uint32_t* __restrict__ jac_out, uint8_t* __restrict__ flags,
                                                           uint32_t* __restrict__ scratch, uint32_t opts) {
  constexpr int L = C::L;
  constexpr int FB = C::FB;
  constexpr int SB = C::SB;
  constexpr int NWIN = (8 * SB + 1 + 4) / 5;  // signed 5-bit windows covering 8*SB + 1 bits
  constexpr int W5 = row5_words<L>();
  constexpr int W3 = row_words<L>();
  // per-lane window table: [workgroup][entry 1..16][thread][W5 words] (entry 0 unused)
  uint32_t* slab = scratch + ((size_t)blockIdx.x * FAST_TABLE_ROWS * WG + threadIdx.x) * (size_t)W5;
  auto row = [&](uint32_t e) { return slab + (size_t)e * WG * W5; };
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;

    Jac<C> q;
    bool rejected = false;
    if (opts & OPT_BASE_IS_GENERATOR) {
      fe_set<C>(q.x, C::GX);
      fe_set<C>(q.y, C::GY);
    } else {
      Fe<L> rx, ry;
      fe_load_be<C>(rx, points + idx * (size_t)(2 * FB));
      fe_load_be<C>(ry, points + idx * (size_t)(2 * FB) + FB);
      fe_to_mont<C>(q.x, rx);
      fe_to_mont<C>(q.y, ry);
      if (opts & OPT_VALIDATE) {
        rejected = !(fe_is_canonical<C>(rx) && fe_is_canonical<C>(ry) && on_curve<C>(q.x, q.y));
      }
    }
    fe_set<C>(q.z, C::ONE);
    {
      JacEntry<C> e1;
      e1.x = q.x; e1.y = q.y; e1.z = q.z; e1.zz = q.z; e1.zzz = q.z;
      entry_store<C>(row(1), e1);
    }
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;

    int b = 0;                    // table-build step: 0 -> T[2] = 2P, 1..14 -> T[b+2] = T[b+1] + P
    int win = NWIN - 1, sub = 5;  // main loop position; the top window needs no doublings
    bool fix_pending = false, fix_lane = false;
    for (;;) {
      const bool building = b < 15;
      if (!building && win < 0) break;
      const bool do_dbl = fix_pending || (building ? (b == 0) : (sub < 5));
      bool step_done;
      if (do_dbl) {
        Jac<C> t;
        jac_dbl<C>(t, q);
        if (fix_pending) {
          jac_select<C>(q, fix_lane, t, q);
          fix_pending = false;
          fix_lane = false;
        } else {
          q = t;
        }
        step_done = true;
      } else {
        uint32_t d = 1;
        bool neg = false;
        if (!building) {
          // Booth digit of window `win`: bits [5*win - 1, 5*win + 4] of the scalar (bit -1 = 0)
          const int pos = 5 * win - 1 + 8;  // bit position in a string with one extra zero byte below
          const int bi = pos >> 3;          // byte 0 is that extra byte, byte j >= 1 is k[SB - j]
          const uint32_t b0 = (bi >= 1 && bi <= SB) ? k[SB - bi] : 0u;
          const uint32_t b1 = (bi + 1 <= SB) ? k[SB - bi - 1] : 0u;
          const uint32_t w6 = ((b0 | (b1 << 8)) >> (pos & 7)) & 0x3fu;
          const uint32_t s = ~((w6 >> 5) - 1u);  // all ones when the window's top bit is set
          uint32_t m = (((1u << 6) - w6 - 1u) & s) | (w6 & ~s);
          d = (m >> 1) + (m & 1u);               // |digit| in 0..16
          neg = (s & 1u) != 0;
        }
        JacEntry<C> e;
        entry_load<C>(e, row(d ? d : 1));
        {
          Fe<L> ny;
          fe_neg<C>(ny, e.y);
          fe_select<C>(e.y, neg, ny, e.y);
        }
        const bool q_inf = fe_is_zero<C>(q.z);
        const bool e_skip = (d == 0) || fe_is_zero<C>(e.z);
        Jac<C> sum;
        bool hz, rz;
        jac_add_raw<C>(sum, hz, rz, q, e);
        const bool same_x = hz && !q_inf && !e_skip;
        fix_lane = same_x && rz;           // q == e: needs a doubling
        const bool to_inf = same_x && !rz; // q == -e
        if (to_inf) fe_zero<C>(sum.z);
        Jac<C> ej;
        ej.x = e.x; ej.y = e.y; ej.z = e.z;
        jac_select<C>(sum, q_inf, ej, sum);
        jac_select<C>(q, e_skip || fix_lane, q, sum);
        fix_pending = __builtin_amdgcn_ballot_w64(fix_lane) != 0;
        step_done = !fix_pending;
      }
      if (step_done) {
        if (building) {
          JacEntry<C> e;
          e.x = q.x; e.y = q.y; e.z = q.z;
          fe_sqr<C>(e.zz, q.z);
          fe_mul<C>(e.zzz, e.zz, q.z);
          entry_store<C>(row(b + 2), e);
          if (++b == 15) fe_zero<C>(q.z);  // accumulator starts at infinity
        } else if (sub < 5) {
          ++sub;
        } else {
          sub = 0;
          --win;
        }
      }
    }
    if (active) {
      uint32_t* o = jac_out + idx * (size_t)W3;
      Pt<C> res;
      res.x = q.x; res.y = q.y; res.z = q.z;
      row_store<C>(o, res);
      flags[idx] = rejected ? 2 : 0;
    }
  }
}

// Fixed-base comb on Jacobian coordinates: same table and digit order as
// mul_base_table (src/curve/projective.rs:965-981) -- window w <-> byte n[len-1-w/2], even w =
// low nibble -- but each of the NW additions is a mixed addition with the affine table entry
// (11 mul instead of 14), with the same special-case patching as the variable-base kernel.
// Writes un-normalised Jacobian rows for k_batch_to_affine.
template <class C>
__global__ void __launch_bounds__(WG) k_scalarmul_base_fast(size_t n, const uint8_t* __restrict__ scalars,
                                                            const uint32_t* __restrict__ table,
                                                            uint32_t* __restrict__ jac_out, uint8_t* __restrict__ flags) {
  constexpr int L = C::L;
  constexpr int SB = C::SB;
  constexpr int NW = 2 * SB;
  constexpr int W3 = row_words<L>();
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
    Jac<C> q;
    fe_set<C>(q.x, C::ONE);
    fe_set<C>(q.y, C::ONE);
    fe_zero<C>(q.z);  // infinity
    for (int w = 0; w < NW; ++w) {
      uint32_t byte = k[SB - 1 - (w >> 1)];
      uint32_t d = (w & 1) ? (byte >> 4) : (byte & 0x0f);
      const uint32_t* __restrict__ e = table + ((size_t)w * 16 + (d ? d : 1)) * (2 * L);
      Fe<L> x2, y2;
#pragma unroll
      for (int i = 0; i < L; ++i) { x2.v[i] = e[i]; y2.v[i] = e[L + i]; }
      const bool q_inf = fe_is_zero<C>(q.z);
      const bool e_skip = (d == 0);
      Jac<C> sum;
      bool hz, rz;
      jac_madd_raw<C>(sum, hz, rz, q, x2, y2);
      const bool same_x = hz && !q_inf && !e_skip;
      const bool need_dbl = same_x && rz;
      if (same_x && !rz) fe_zero<C>(sum.z);  // q == -entry
      Jac<C> ej;
      ej.x = x2; ej.y = y2;
      fe_set<C>(ej.z, C::ONE);
      jac_select<C>(sum, q_inf, ej, sum);
      if (__builtin_amdgcn_ballot_w64(need_dbl) != 0) {  // q == entry: rare, wave-uniform branch
        Jac<C> t;
        jac_dbl<C>(t, q);
        jac_select<C>(sum, need_dbl, t, sum);
      }
      jac_select<C>(q, e_skip, q, sum);
    }
    if (active) {
      Pt<C> res;
      res.x = q.x; res.y = q.y; res.z = q.z;
      row_store<C>(jac_out + idx * (size_t)W3, res);
      flags[idx] = 0;
    }
  }
}

// edwards25519 fixed base with the whole comb table staged in LDS (BASELINE.json configs[2]:
// "comb table in LDS").  One 1024-thread workgroup per CU shares a 96 KiB image of the table
// (64 windows x 16 digits x {x, y, 2d*x*y}, digit 0 = the neutral element so the loop has no
// branch); lanes read their entry with ds_read_b128.  Same arithmetic as the default path of
// k_ed_scalarmul_base (7-multiplication cached addition), which reads the table through L1/L2.
constexpr int ED_LDS_BLOCK = 1024;
template <class C>
__global__ void __launch_bounds__(ED_LDS_BLOCK) k_ed_scalarmul_base_lds(size_t n, const uint8_t* __restrict__ scalars,
                                                                        const uint32_t* __restrict__ table,
                                                                        uint32_t* __restrict__ rows_out,
                                                                        uint8_t* __restrict__ flags) {
  constexpr int L = C::L;
  constexpr int EW = 3 * L;  // LDS words per entry
  extern __shared__ uint4 lds4[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(lds4);
  for (int i = threadIdx.x; i < 64 * 16 * EW; i += ED_LDS_BLOCK) {
    const int e = i / EW, j = i - e * EW;
    uint32_t v;
    if ((e & 15) == 0) v = (j >= L && j < 2 * L) ? C::ONE[0] * (j == L) : 0u;  // (0, 1, 0)
    else v = table[(size_t)e * (4 * L) + (j < 2 * L ? j : j + L)];              // x, y | skip t | 2dxy
    lds[i] = v;
  }
  __syncthreads();
  constexpr int W3 = row_words<L>();
  for (size_t base = (size_t)blockIdx.x * ED_LDS_BLOCK; base < n; base += (size_t)gridDim.x * ED_LDS_BLOCK) {
    const size_t gid = base + threadIdx.x;
    const bool active = gid < n;
    const size_t idx = active ? gid : n - 1;
    const uint8_t* __restrict__ k = scalars + idx * 32;
    EdPt<C> q;
    ed_set_identity<C>(q);
    for (int w = 0; w < 64; ++w) {
      uint32_t byte = k[31 - (w >> 1)];
      uint32_t d = (w & 1) ? (byte >> 4) : (byte & 0x0f);
      const uint4* e = reinterpret_cast<const uint4*>(lds + (w * 16 + d) * EW);
      uint32_t v[EW];
#pragma unroll
      for (int i = 0; i < EW / 4; ++i) {
        uint4 t = e[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
      }
      Fe<L> x2, y2, t2d;
#pragma unroll
      for (int i = 0; i < L; ++i) { x2.v[i] = v[i]; y2.v[i] = v[L + i]; t2d.v[i] = v[2 * L + i]; }
      ed_add_cached<C>(q, q, x2, y2, t2d);
    }
    if (active) {
      Pt<C> row;
      row.x = q.x; row.y = q.y; row.z = q.z;
      row_store<C>(rows_out + idx * (size_t)W3, row);
      flags[idx] = 0;
    }
  }
}

// Normalise a batch of un-normalised points (rows of W3 words: X, Y, Z Montgomery limbs).
//   MODE 0: homogeneous x = X/Z, y = Y/Z, big-endian bytes (projective.rs:655-682)
//   MODE 1: Jacobian x = X/Z^2, y = Y/Z^3, big-endian bytes
//   MODE 2: edwards25519: x = X/Z, y = Y/Z, little-endian bytes, flag 1 = neutral element
//           (curve25519.rs:663-666; Z is never 0 on a complete Edwards curve)
// flags[i] on entry: 2 marks a rejected input (kept, zero output); on exit 1 marks infinity.
// Thread t of a workgroup handles units tile + u*WG + t, u = 0..U-1, with one inversion.
//   MODE 3: curve25519 x-only: u = X/Z with 0 for Z = 0, 32 little-endian bytes per unit,
//           flag 1 = result is zero (curve25519.rs:529-532; x25519.rs:33-35)
enum { NORM_HOMOGENEOUS = 0, NORM_JACOBIAN = 1, NORM_EDWARDS = 2, NORM_MONTGOMERY_U = 3 };
// PLAIN: the rows hold plain canonical integers (written by the unsaturated kernels) and are
// taken into this kernel's Montgomery domain on load.
template <class C, int MODE, int U, bool PLAIN = false>
__global__ void __launch_bounds__(WG) k_batch_to_affine(size_t n, const uint32_t* __restrict__ pts,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ flags) {
  constexpr int L = C::L;
  constexpr int FB = C::FB;
  constexpr int W3 = row_words<L>();
  const size_t tile_units = (size_t)WG * U;
  for (size_t tile = (size_t)blockIdx.x * tile_units; tile < n; tile += (size_t)gridDim.x * tile_units) {
    Fe<L> pre[U];  // prefix products of the (substituted) Z values
    Fe<L> one;
    fe_set<C>(one, C::ONE);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      Fe<L> z = one;
      if (i < n) {
        const uint32_t* r = pts + i * (size_t)W3 + 2 * L;
#pragma unroll
        for (int j = 0; j < L; ++j) z.v[j] = r[j];
        if constexpr (PLAIN) fe_to_mont<C>(z, z);
        if (fe_is_zero<C>(z)) z = one;  // z_inverse_ct substitutes 1 (projective.rs:655-659)
      }
      if (u == 0) pre[0] = z;
      else fe_mul<C>(pre[u], pre[u - 1], z);
    }
    Fe<L> inv;
    fe_inv<C>(inv, pre[U - 1]);
#pragma unroll
    for (int u = U - 1; u >= 0; --u) {
      const size_t i = tile + (size_t)u * WG + threadIdx.x;
      Fe<L> x = one, y = one, z = one;
      bool present = false;
      if (i < n) {
        const uint32_t* r = pts + i * (size_t)W3;
#pragma unroll
        for (int j = 0; j < L; ++j) { x.v[j] = r[j]; y.v[j] = r[L + j]; z.v[j] = r[2 * L + j]; }
        if constexpr (PLAIN) { fe_to_mont<C>(x, x); fe_to_mont<C>(y, y); fe_to_mont<C>(z, z); }
        present = !fe_is_zero<C>(z);
        if (!present) z = one;
      }
      Fe<L> zi;
      if (u > 0) {
        fe_mul<C>(zi, inv, pre[u - 1]);
        fe_mul<C>(inv, inv, z);
      } else {
        zi = inv;
      }
      Fe<L> ax, ay;
      if constexpr (MODE == NORM_JACOBIAN) {
        Fe<L> zi2;
        fe_sqr<C>(zi2, zi);
        fe_mul<C>(ax, x, zi2);
        fe_mul<C>(zi2, zi2, zi);
        fe_mul<C>(ay, y, zi2);
      } else {
        fe_mul<C>(ax, x, zi);
        fe_mul<C>(ay, y, zi);
      }
      if (i < n) {
        const bool rejected = flags[i] == 2;
        Fe<L> t;
        if constexpr (MODE == NORM_MONTGOMERY_U) {
          // u = X * invert_or_zero(Z) (curve25519.rs:529-532): Z = 0 gives u = 0
          fe_from_mont<C>(t, ax);
          if (!present) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)FB, t);
          flags[i] = fe_is_zero<C>(t) ? 1 : 0;
        } else if constexpr (MODE == NORM_EDWARDS) {
          const bool neutral = fe_is_zero<C>(ax) && fe_eq<C>(ay, one);
          fe_from_mont<C>(t, ax);
          if (rejected) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)(2 * FB), t);
          fe_from_mont<C>(t, ay);
          if (rejected) fe_zero<C>(t);
          fe_store_le<C>(out + i * (size_t)(2 * FB) + FB, t);
          flags[i] = rejected ? 2 : (neutral ? 1 : 0);
        } else {
          const bool ok = present && !rejected;
          fe_from_mont<C>(t, ax);
          if (!ok) fe_zero<C>(t);
          fe_store_be<C>(out + i * (size_t)(2 * FB), t);
          fe_from_mont<C>(t, ay);
          if (!ok) fe_zero<C>(t);
          fe_store_be<C>(out + i * (size_t)(2 * FB) + FB, t);
          flags[i] = rejected ? 2 : (present ? 0 : 1);
        }
      }
    }
  }
}

}  // namespace eccx
