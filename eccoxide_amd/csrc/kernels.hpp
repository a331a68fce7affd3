// Reference-mirroring kernels on saturated canonical limbs (fe.hpp) and the shared result tails:
// one (scalar, base) pair per lane, 256-thread workgroups, no cross-lane communication.
//
//   k_scalarmul_base<C>  Point::mul_base             curve_macros.rs:55-63 / :111-119
//                        -> mul_base_table_{am3,a0}  projective.rs:965-981 / :945-961
//   k_ed_scalarmul_var   curve25519::Point::scale    curve25519.rs:746-762 (double-and-add)
//   k_ed_scalarmul_base  curve25519::Point::mul_base curve25519.rs:840-851
//   k_point_add, k_ed_point_add                      the group law on batches
//   tails                to_affine_ct                projective.rs:655-682 ; curve25519.rs:663-666
//
// The reference-mirroring Weierstrass LADDER (&Point * &Scalar, projective.rs:871-896) is
// k_scalarmul_var_mirror_unsat in kernels_unsat.hpp; it ends in store_result below.  The
// default kernels are in kernels_unsat.hpp as well.
#pragma once
#include "curve.hpp"
#include "inv_gcd.hpp"

namespace eccx {

enum : uint32_t {
  OPT_BASE_IS_GENERATOR = 1u << 0,  // ignore `points`, use the curve generator
  OPT_OUT_TABLE = 1u << 1,          // write affine Montgomery limbs (comb-table rows) instead of bytes
  OPT_VALIDATE = 1u << 2,           // reject non-canonical / off-curve input points (flag 2)
  OPT_OUT_ROWS = 1u << 3,           // write the un-normalised result as a row of Montgomery limbs
                                    // (X, Y, Z) for k_batch_to_affine instead of normalising here
  OPT_ONLY_MARKED = 1u << 7,        // variable base: process only the units whose flag is 0xFE (redo marker)
  OPT_CT_SCAN = 1u << 6,            // table lookups read EVERY entry and keep the wanted one with selects
                                    // (select_from_table, projective.rs:427-434 / curve25519.rs:862-869):
                                    // no memory address and no branch depends on a scalar digit
};

constexpr int WG = 256;
// flag value of a unit the affine-table ladder could not finish (base point of order <= 16, or not a curve
// point at all): it is redone by the kernel launched behind with OPT_ONLY_MARKED, and skipped by the
// normalisation in between
constexpr uint8_t FLAG_REDO = 0xFE;

template <int L>
constexpr int row_words() { return ((3 * L + 3) / 4) * 4; }  // padded to 16 bytes

template <class C>
ECCX_DEV void row_store(uint32_t* __restrict__ row, const Pt<C>& p) {
  constexpr int L = C::L;
  uint32_t w[row_words<L>()];
#pragma unroll
  for (int i = 0; i < L; ++i) { w[i] = p.x.v[i]; w[L + i] = p.y.v[i]; w[2 * L + i] = p.z.v[i]; }
#pragma unroll
  for (int i = 3 * L; i < row_words<L>(); ++i) w[i] = 0;
  uint4* dst = reinterpret_cast<uint4*>(row);
#pragma unroll
  for (int i = 0; i < row_words<L>() / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

template <class C>
ECCX_DEV void row_load(Pt<C>& p, const uint32_t* __restrict__ row) {
  constexpr int L = C::L;
  uint32_t w[row_words<L>()];
  const uint4* src = reinterpret_cast<const uint4*>(row);
#pragma unroll
  for (int i = 0; i < row_words<L>() / 4; ++i) {
    uint4 q = src[i];
    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
  }
#pragma unroll
  for (int i = 0; i < L; ++i) { p.x.v[i] = w[i]; p.y.v[i] = w[L + i]; p.z.v[i] = w[2 * L + i]; }
}

// y^2 == x^3 + a x + b  (affine.rs:103-119), Montgomery inputs
template <class C>
ECCX_DEV bool on_curve(const Fe<C::L>& x, const Fe<C::L>& y) {
  Fe<C::L> y2, x3, t, bb;
  fe_sqr<C>(y2, y);
  fe_sqr<C>(x3, x);
  fe_mul<C>(x3, x3, x);
  if constexpr (!C::A0) {  // a = -3
    fe_add<C>(t, x, x);
    fe_add<C>(t, t, x);
    fe_sub<C>(x3, x3, t);
  }
  fe_set<C>(bb, C::B);
  fe_add<C>(x3, x3, bb);
  return fe_eq<C>(y2, x3);
}

// (X:Y:Z) -> affine x, y (Montgomery) ; returns false for the point at infinity
template <class C>
ECCX_DEV bool to_affine(Fe<C::L>& ax, Fe<C::L>& ay, const Pt<C>& q) {
  bool present = !fe_is_zero<C>(q.z);
  Fe<C::L> z, zi, one;
  fe_set<C>(one, C::ONE);
  fe_select<C>(z, present, q.z, one);  // z_inverse_ct: substitute 1 (projective.rs:655-659)
  fe_inv_fast<C>(zi, z);
  fe_mul<C>(ax, q.x, zi);
  fe_mul<C>(ay, q.y, zi);
  return present;
}

// Common tail: write the result of one unit.
//   out   : n x 2FB canonical big-endian x||y (zeros for infinity)      [or table rows]
//   flags : n bytes, 0 = finite, 1 = infinity, 2 = rejected input
//   proj  : optional n x 3FB canonical big-endian X||Y||Z, the reference's
//           un-normalised result
template <class C>
ECCX_DEV void store_result(size_t idx, const Pt<C>& q, bool rejected, uint8_t* __restrict__ out,
                           uint8_t* __restrict__ flags, uint8_t* __restrict__ proj, uint32_t opts) {
  constexpr int L = C::L;
  constexpr int FB = C::FB;
  if (opts & OPT_OUT_ROWS) {
    row_store<C>(reinterpret_cast<uint32_t*>(out) + idx * (size_t)row_words<L>(), q);
    flags[idx] = rejected ? 2 : 0;
    return;
  }
  Fe<L> ax, ay, t;
  bool present = to_affine<C>(ax, ay, q);
  if (opts & OPT_OUT_TABLE) {
    uint32_t* row = reinterpret_cast<uint32_t*>(out) + idx * (size_t)(2 * L);
#pragma unroll
    for (int i = 0; i < L; ++i) { row[i] = ax.v[i]; row[L + i] = ay.v[i]; }
    return;
  }
  bool ok = present && !rejected;
  fe_from_mont<C>(t, ax);
  if (!ok) fe_zero<C>(t);
  fe_store_be<C>(out + idx * (size_t)(2 * FB), t);
  fe_from_mont<C>(t, ay);
  if (!ok) fe_zero<C>(t);
  fe_store_be<C>(out + idx * (size_t)(2 * FB) + FB, t);
  flags[idx] = rejected ? 2 : (present ? 0 : 1);
  if (proj) {
    uint8_t* pr = proj + idx * (size_t)(3 * FB);
    fe_from_mont<C>(t, q.x); fe_store_be<C>(pr, t);
    fe_from_mont<C>(t, q.y); fe_store_be<C>(pr + FB, t);
    fe_from_mont<C>(t, q.z); fe_store_be<C>(pr + 2 * FB, t);
  }
}

// Fixed-base comb: table[w][d] (d = 1..15) = d * 16^w * G as affine Montgomery limbs,
// 2L words per entry, entry 0 of each window unused (digit 0 selects infinity).
template <class C>
__global__ void __launch_bounds__(WG) k_scalarmul_base(size_t n, const uint8_t* __restrict__ scalars,
                                                       const uint32_t* __restrict__ table,
                                                       uint8_t* __restrict__ out, uint8_t* __restrict__ flags,
                                                       uint8_t* __restrict__ proj, uint32_t opts) {
  constexpr int L = C::L;
  constexpr int SB = C::SB;
  constexpr int NW = 2 * SB;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
  const size_t gid = base + threadIdx.x;
  const bool active = gid < n;
  const size_t idx = active ? gid : n - 1;
  const uint8_t* __restrict__ k = scalars + idx * (size_t)SB;
  Pt<C> q;
  pt_set_inf<C>(q);
  for (int w = 0; w < NW; ++w) {
    uint32_t byte = k[SB - 1 - (w >> 1)];
    uint32_t d = (w & 1) ? (byte >> 4) : (byte & 0x0f);  // low nibble first (projective.rs:974-976)
    Pt<C> sel;
    if (opts & OPT_CT_SCAN) {
      // select_from_table (projective.rs:427-434): every entry of the window is read (the address
      // is the same for all lanes) and kept under a per-lane select; digit 0 keeps (0 : 1 : 0)
      pt_set_inf<C>(sel);
      Fe<L> one;
      fe_set<C>(one, C::ONE);
      for (uint32_t j = 1; j < 16; ++j) {
        const uint32_t* __restrict__ e = table + ((size_t)w * 16 + j) * (2 * L);
        Fe<L> ex, ey;
#pragma unroll
        for (int i = 0; i < L; ++i) { ex.v[i] = e[i]; ey.v[i] = e[L + i]; }
        const bool take = (j == d);
        fe_select<C>(sel.x, take, ex, sel.x);
        fe_select<C>(sel.y, take, ey, sel.y);
        fe_select<C>(sel.z, take, one, sel.z);
      }
    } else {
      const uint32_t* __restrict__ e = table + ((size_t)w * 16 + d) * (2 * L);
#pragma unroll
      for (int i = 0; i < L; ++i) { sel.x.v[i] = e[i]; sel.y.v[i] = e[L + i]; }
      fe_set<C>(sel.z, C::ONE);
      if (d == 0) pt_set_inf<C>(sel);
    }
    pt_add<C>(q, q, sel);
  }
  if (active) store_result<C>(idx, q, false, out, flags, proj, opts);
  }  // grid-stride
}

// ---- edwards25519 -----------------------------------------------------------------
// out: n x 64 affine x||y little-endian (curve25519.rs:138); flags: 1 = neutral element.
// proj (optional): n x 128 little-endian X||Y||Z||T.
template <class C>
ECCX_DEV void ed_store_result(size_t idx, const EdPt<C>& q, bool rejected, uint8_t* __restrict__ out,
                              uint8_t* __restrict__ flags, uint8_t* __restrict__ proj, uint32_t opts) {
  constexpr int L = C::L;
  if (opts & OPT_OUT_ROWS) {
    Pt<C> row;
    row.x = q.x; row.y = q.y; row.z = q.z;
    row_store<C>(reinterpret_cast<uint32_t*>(out) + idx * (size_t)row_words<L>(), row);
    flags[idx] = rejected ? 2 : 0;
    return;
  }
  Fe<L> zi, ax, ay, t, one;
  fe_inv_fast<C>(zi, q.z);  // Z != 0 on a complete Edwards curve (curve25519.rs:663-666)
  fe_mul<C>(ax, q.x, zi);
  fe_mul<C>(ay, q.y, zi);
  if (opts & OPT_OUT_TABLE) {
    // comb rows: x, y, t = x*y (from_affine, curve25519.rs:638-645) and 2d*x*y, the operand
    // the addition would otherwise recompute (the reference's CachedPoint, curve25519.rs:712-729)
    uint32_t* row = reinterpret_cast<uint32_t*>(out) + idx * (size_t)(4 * L);
    Fe<L> t2d;
    fe_mul<C>(t, ax, ay);
    fe_mul_k<C>(t2d, t, C::D2);
#pragma unroll
    for (int i = 0; i < L; ++i) { row[i] = ax.v[i]; row[L + i] = ay.v[i]; row[2 * L + i] = t.v[i]; row[3 * L + i] = t2d.v[i]; }
    return;
  }
  fe_set<C>(one, C::ONE);
  bool neutral = fe_is_zero<C>(ax) && fe_eq<C>(ay, one);
  fe_from_mont<C>(t, ax);
  if (rejected) fe_zero<C>(t);
  fe_store_le<C>(out + idx * 64, t);
  fe_from_mont<C>(t, ay);
  if (rejected) fe_zero<C>(t);
  fe_store_le<C>(out + idx * 64 + 32, t);
  flags[idx] = rejected ? 2 : (neutral ? 1 : 0);
  if (proj) {
    uint8_t* pr = proj + idx * 128;
    fe_from_mont<C>(t, q.x); fe_store_le<C>(pr, t);
    fe_from_mont<C>(t, q.y); fe_store_le<C>(pr + 32, t);
    fe_from_mont<C>(t, q.z); fe_store_le<C>(pr + 64, t);
    fe_from_mont<C>(t, q.t); fe_store_le<C>(pr + 96, t);
  }
}

template <class C>
__global__ void __launch_bounds__(WG) k_ed_scalarmul_var(size_t n, const uint8_t* __restrict__ scalars,
                                                         const uint8_t* __restrict__ points,
                                                         uint8_t* __restrict__ out, uint8_t* __restrict__ flags,
                                                         uint8_t* __restrict__ proj, uint32_t opts) {
  constexpr int L = C::L;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
  const size_t gid = base + threadIdx.x;
  const bool active = gid < n;
  const size_t idx = active ? gid : n - 1;
  EdPt<C> p, q;
  bool rejected = false;
  if (opts & OPT_BASE_IS_GENERATOR) {
    fe_set<C>(p.x, C::GX);
    fe_set<C>(p.y, C::GY);
  } else {
    Fe<L> rx, ry;
    fe_load_le<C>(rx, points + idx * 64);
    fe_load_le<C>(ry, points + idx * 64 + 32);
    fe_to_mont<C>(p.x, rx);
    fe_to_mont<C>(p.y, ry);
    if (opts & OPT_VALIDATE) {
      // -x^2 + y^2 == 1 + d x^2 y^2  <=>  2(y^2 - x^2 - 1) == 2d x^2 y^2  (curve25519.rs:649-660)
      Fe<L> xx, yy, lhs, rhs, one;
      fe_sqr<C>(xx, p.x);
      fe_sqr<C>(yy, p.y);
      fe_set<C>(one, C::ONE);
      fe_sub<C>(lhs, yy, xx);
      fe_sub<C>(lhs, lhs, one);
      fe_add<C>(lhs, lhs, lhs);
      fe_mul<C>(rhs, xx, yy);
      fe_mul_k<C>(rhs, rhs, C::D2);
      rejected = !(fe_is_canonical<C>(rx) && fe_is_canonical<C>(ry) && fe_eq<C>(lhs, rhs));
    }
  }
  fe_set<C>(p.z, C::ONE);
  fe_mul<C>(p.t, p.x, p.y);  // from_affine
  ed_set_identity<C>(q);
  const uint8_t* __restrict__ k = scalars + idx * 32;
  // curve25519.rs:746-757: per bit, MSB first: double, add, keep the sum if the bit is set
  for (int bit = 255; bit >= 0; --bit) {
    uint32_t byte = k[31 - (bit >> 3)];
    bool take = (byte >> (bit & 7)) & 1u;
    ed_dbl<C>(q, q);
    EdPt<C> added;
    ed_add<C>(added, q, p);
    fe_select<C>(q.x, take, added.x, q.x);
    fe_select<C>(q.y, take, added.y, q.y);
    fe_select<C>(q.z, take, added.z, q.z);
    fe_select<C>(q.t, take, added.t, q.t);
  }
  if (active) ed_store_result<C>(idx, q, rejected, out, flags, proj, opts);
  }  // grid-stride
}

// table[w][d] (d = 1..15): x, y, t = x*y, 2d*x*y limbs, 4L words per entry
template <class C>
__global__ void __launch_bounds__(WG) k_ed_scalarmul_base(size_t n, const uint8_t* __restrict__ scalars,
                                                          const uint32_t* __restrict__ table,
                                                          uint8_t* __restrict__ out, uint8_t* __restrict__ flags,
                                                          uint8_t* __restrict__ proj, uint32_t opts) {
  constexpr int L = C::L;
  for (size_t base = (size_t)blockIdx.x * WG; base < n; base += (size_t)gridDim.x * WG) {
  const size_t gid = base + threadIdx.x;
  const bool active = gid < n;
  const size_t idx = active ? gid : n - 1;
  const uint8_t* __restrict__ k = scalars + idx * 32;
  EdPt<C> q;
  ed_set_identity<C>(q);
  for (int w = 0; w < 64; ++w) {
    uint32_t byte = k[31 - (w >> 1)];  // indexes the big-endian scalar (curve25519.rs:842-846)
    uint32_t d = (w & 1) ? (byte >> 4) : (byte & 0x0f);
    EdPt<C> sel;
    if (opts & OPT_CT_SCAN) {
      // select_from_table (curve25519.rs:862-869): read all 16 entries, keep one by select; the
      // general addition (Point::add) follows, as in the reference
      ed_set_identity<C>(sel);
      Fe<L> one;
      fe_set<C>(one, C::ONE);
      for (uint32_t j = 1; j < 16; ++j) {
        const uint32_t* __restrict__ ej = table + ((size_t)w * 16 + j) * (4 * L);
        Fe<L> ex, ey, et;
#pragma unroll
        for (int i = 0; i < L; ++i) { ex.v[i] = ej[i]; ey.v[i] = ej[L + i]; et.v[i] = ej[2 * L + i]; }
        const bool take = (j == d);
        fe_select<C>(sel.x, take, ex, sel.x);
        fe_select<C>(sel.y, take, ey, sel.y);
        fe_select<C>(sel.t, take, et, sel.t);
      }
      ed_add<C>(q, q, sel);
      continue;
    }
    const uint32_t* __restrict__ e = table + ((size_t)w * 16 + d) * (4 * L);
#pragma unroll
    for (int i = 0; i < L; ++i) { sel.x.v[i] = e[i]; sel.y.v[i] = e[L + i]; }
    if (opts & OPT_OUT_ROWS) {
      // default path: Z2 = 1 and 2d*T2 comes from the table -> 7 multiplications
      // (same point as Point::add, curve25519.rs:695-710, in other projective coordinates)
      Fe<L> t2d;
#pragma unroll
      for (int i = 0; i < L; ++i) t2d.v[i] = e[3 * L + i];
      if (d == 0) { fe_zero<C>(sel.x); fe_set<C>(sel.y, C::ONE); fe_zero<C>(t2d); }
      ed_add_cached<C>(q, q, sel.x, sel.y, t2d);
    } else {
#pragma unroll
      for (int i = 0; i < L; ++i) sel.t.v[i] = e[2 * L + i];
      fe_set<C>(sel.z, C::ONE);
      if (d == 0) ed_set_identity<C>(sel);
      ed_add<C>(q, q, sel);
    }
  }
  if (active) ed_store_result<C>(idx, q, false, out, flags, proj, opts);
  }  // grid-stride
}

// ---- batched group law: out[i] = a[i] + b[i]  (or a[i] - b[i]) -----------------------------
//   impl Add/Sub/Neg for Point, CurveGroup::double   src/curve/fiat/curve_macros.rs:297-411,
//   src/curve/group.rs:28-70 -> add_or_double_{am3,a0} (projective.rs:1003-1019) i.e. the
//   complete RCB addition, which also covers a == b (doubling), a == -b and infinity;
//   edwards25519: Point::add (curve25519.rs:695-710).
// Inputs are affine x||y with an optional flag array (1 = point at infinity); results go out
// as un-normalised rows for k_batch_to_affine.
enum : uint32_t { OPT_NEGATE_B = 1u << 5 };

template <class C>
__global__ void __launch_bounds__(WG) k_point_add(size_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ a_inf,
                                                  const uint8_t* __restrict__ b, const uint8_t* __restrict__ b_inf,
                                                  uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags,
                                                  uint32_t opts) {
  constexpr int L = C::L;
  constexpr int FB = C::FB;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    Pt<C> p, q, r;
    Fe<L> raw;
    fe_load_be<C>(raw, a + i * (size_t)(2 * FB)); fe_to_mont<C>(p.x, raw);
    fe_load_be<C>(raw, a + i * (size_t)(2 * FB) + FB); fe_to_mont<C>(p.y, raw);
    fe_set<C>(p.z, C::ONE);
    if (a_inf && a_inf[i] == 1) pt_set_inf<C>(p);
    fe_load_be<C>(raw, b + i * (size_t)(2 * FB)); fe_to_mont<C>(q.x, raw);
    fe_load_be<C>(raw, b + i * (size_t)(2 * FB) + FB); fe_to_mont<C>(q.y, raw);
    fe_set<C>(q.z, C::ONE);
    if (opts & OPT_NEGATE_B) fe_neg<C>(q.y, q.y);
    if (b_inf && b_inf[i] == 1) pt_set_inf<C>(q);
    pt_add<C>(r, p, q);
    row_store<C>(rows_out + i * (size_t)row_words<L>(), r);
    flags[i] = ((a_inf && a_inf[i] == 2) || (b_inf && b_inf[i] == 2)) ? 2 : 0;  // rejected operands stay rejected
  }
}

template <class C>
__global__ void __launch_bounds__(WG) k_ed_point_add(size_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ a_fl,
                                                     const uint8_t* __restrict__ b, const uint8_t* __restrict__ b_fl,
                                                     uint32_t* __restrict__ rows_out, uint8_t* __restrict__ flags,
                                                     uint32_t opts) {
  constexpr int L = C::L;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    EdPt<C> p, q, r;
    Fe<L> raw;
    fe_load_le<C>(raw, a + i * 64); fe_to_mont<C>(p.x, raw);
    fe_load_le<C>(raw, a + i * 64 + 32); fe_to_mont<C>(p.y, raw);
    fe_set<C>(p.z, C::ONE);
    fe_mul<C>(p.t, p.x, p.y);
    fe_load_le<C>(raw, b + i * 64); fe_to_mont<C>(q.x, raw);
    fe_load_le<C>(raw, b + i * 64 + 32); fe_to_mont<C>(q.y, raw);
    if (opts & OPT_NEGATE_B) fe_neg<C>(q.x, q.x);  // -(x, y) = (-x, y) (curve25519.rs:731-738)
    fe_set<C>(q.z, C::ONE);
    fe_mul<C>(q.t, q.x, q.y);
    ed_add<C>(r, p, q);
    Pt<C> row;
    row.x = r.x; row.y = r.y; row.z = r.z;
    row_store<C>(rows_out + i * (size_t)row_words<L>(), row);
    flags[i] = ((a_fl && a_fl[i] == 2) || (b_fl && b_fl[i] == 2)) ? 2 : 0;
  }
}

// ---- curve25519 x-only Montgomery ladder (X25519): k_x25519_ladder_unsat, kernels_unsat.hpp
enum : uint32_t { OPT_X25519_RFC = 1u << 4 };

}  // namespace eccx
