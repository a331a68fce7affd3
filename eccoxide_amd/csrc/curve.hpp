// Point arithmetic for the gfx950 scalar-multiplication kernels: one curve point
// per lane, coordinates in VGPRs.
//
// Mirrors the reference's src/curve/projective.rs operation for operation, so the
// un-normalised (X:Y:Z) a kernel produces is the same residue triple the reference
// computes, not merely a projectively equivalent one:
//   pt_add<am3>  add_different_am3  projective.rs:340-423  (RCB 2016, Alg. 4)
//   pt_dbl<am3>  double_am3         projective.rs:586-646  (Alg. 6)
//   pt_add<a0>   add_different_a0   projective.rs:268-338
//   pt_dbl<a0>   double_a0          projective.rs:544-583
//   ed_add       Point::add         curve25519.rs:695-710  (extended, a = -1)
//   ed_dbl       double_parts+double curve25519.rs:604-619, :669-677
#pragma once
#include "fe.hpp"

namespace eccx {

#include "curve_consts.inc"

template <class C>
struct Pt {
  Fe<C::L> x, y, z;
};

template <class C>
ECCX_DEV void pt_set_inf(Pt<C>& p) {  // (0 : 1 : 0), projective.rs:152-156
  fe_zero<C>(p.x);
  fe_set<C>(p.y, C::ONE);
  fe_zero<C>(p.z);
}

#define M_(r, a, b) fe_mul<C>(r, a, b)
#define A_(r, a, b) fe_add<C>(r, a, b)
#define S_(r, a, b) fe_sub<C>(r, a, b)

template <class C>
ECCX_DEV void pt_add(Pt<C>& r, const Pt<C>& p, const Pt<C>& q) {
  using F = Fe<C::L>;
  F t0, t1, t2, t3, t4, x3, y3, z3;
  M_(t0, p.x, q.x); M_(t1, p.y, q.y); M_(t2, p.z, q.z);
  A_(t3, p.x, p.y); A_(t4, q.x, q.y); M_(t3, t3, t4);
  A_(t4, t0, t1); S_(t3, t3, t4); A_(t4, p.y, p.z);
  A_(x3, q.y, q.z); M_(t4, t4, x3); A_(x3, t1, t2);
  S_(t4, t4, x3); A_(x3, p.x, p.z); A_(y3, q.x, q.z);
  M_(x3, x3, y3); A_(y3, t0, t2); S_(y3, x3, y3);
  if constexpr (C::A0) {
    A_(x3, t0, t0); A_(t0, x3, t0); fe_mul_k<C>(t2, t2, C::B3);
    A_(z3, t1, t2); S_(t1, t1, t2); fe_mul_k<C>(y3, y3, C::B3);
    M_(x3, t4, y3); M_(t2, t3, t1); S_(x3, t2, x3);
    M_(y3, y3, t0); M_(t1, t1, z3); A_(y3, t1, y3);
    M_(t0, t0, t3); M_(z3, z3, t4); A_(z3, z3, t0);
  } else {
    fe_mul_k<C>(z3, t2, C::B); S_(x3, y3, z3); A_(z3, x3, x3);
    A_(x3, x3, z3); S_(z3, t1, x3); A_(x3, t1, x3);
    fe_mul_k<C>(y3, y3, C::B); A_(t1, t2, t2); A_(t2, t1, t2);
    S_(y3, y3, t2); S_(y3, y3, t0); A_(t1, y3, y3);
    A_(y3, t1, y3); A_(t1, t0, t0); A_(t0, t1, t0);
    S_(t0, t0, t2); M_(t1, t4, y3); M_(t2, t0, y3);
    M_(y3, x3, z3); A_(y3, y3, t2); M_(x3, t3, x3);
    S_(x3, x3, t1); M_(z3, t4, z3); M_(t1, t3, t0);
    A_(z3, z3, t1);
  }
  r.x = x3; r.y = y3; r.z = z3;
}

template <class C>
ECCX_DEV void pt_dbl(Pt<C>& r, const Pt<C>& p) {
  using F = Fe<C::L>;
  if constexpr (C::A0) {
    F t0, t1, t2, x3, y3, z3;
    fe_sqr<C>(t0, p.y); A_(z3, t0, t0); A_(z3, z3, z3);
    A_(z3, z3, z3); M_(t1, p.y, p.z); fe_sqr<C>(t2, p.z);
    fe_mul_k<C>(t2, t2, C::B3); M_(x3, t2, z3); A_(y3, t0, t2);
    M_(z3, t1, z3); A_(t1, t2, t2); A_(t2, t1, t2);
    S_(t0, t0, t2); M_(y3, t0, y3); A_(y3, x3, y3);
    M_(t1, p.x, p.y); M_(x3, t0, t1); A_(x3, x3, x3);
    r.x = x3; r.y = y3; r.z = z3;
  } else {
    F t0, t1, t2, t3, x3, y3, z3;
    fe_sqr<C>(t0, p.x); fe_sqr<C>(t1, p.y); fe_sqr<C>(t2, p.z);
    M_(t3, p.x, p.y); A_(t3, t3, t3); M_(z3, p.x, p.z);
    A_(z3, z3, z3); fe_mul_k<C>(y3, t2, C::B); S_(y3, y3, z3);
    A_(x3, y3, y3); A_(y3, x3, y3); S_(x3, t1, y3);
    A_(y3, t1, y3); M_(y3, x3, y3); M_(x3, x3, t3);
    A_(t3, t2, t2); A_(t2, t2, t3); fe_mul_k<C>(z3, z3, C::B);
    S_(z3, z3, t2); S_(z3, z3, t0); A_(t3, z3, z3);
    A_(z3, z3, t3); A_(t3, t0, t0); A_(t0, t3, t0);
    S_(t0, t0, t2); M_(t0, t0, z3); A_(y3, y3, t0);
    M_(t0, p.y, p.z); A_(t0, t0, t0); M_(z3, t0, z3);
    S_(x3, x3, z3); M_(z3, t0, t1); A_(z3, z3, z3);
    A_(z3, z3, z3);
    r.x = x3; r.y = y3; r.z = z3;
  }
}

// ---- twisted Edwards, extended coordinates (edwards25519) ----------------------
template <class C>
struct EdPt {
  Fe<C::L> x, y, z, t;
};

template <class C>
ECCX_DEV void ed_set_identity(EdPt<C>& p) {  // (0, 1, 1, 0), curve25519.rs:623-628
  fe_zero<C>(p.x);
  fe_set<C>(p.y, C::ONE);
  fe_set<C>(p.z, C::ONE);
  fe_zero<C>(p.t);
}

template <class C>
ECCX_DEV void ed_add(EdPt<C>& r, const EdPt<C>& p, const EdPt<C>& q) {
  using F = Fe<C::L>;
  F aa, bb, cc, dd, e, f, g, h, u, v;
  S_(u, p.y, p.x); S_(v, q.y, q.x); M_(aa, u, v);
  A_(u, p.y, p.x); A_(v, q.y, q.x); M_(bb, u, v);
  fe_mul_k<C>(u, p.t, C::D2); M_(cc, u, q.t);
  M_(u, p.z, q.z); A_(dd, u, u);
  S_(e, bb, aa); S_(f, dd, cc); A_(g, dd, cc); A_(h, bb, aa);
  M_(r.x, e, f); M_(r.y, g, h); M_(r.z, f, g); M_(r.t, e, h);
}

// r = p + (x2, y2) for an affine table entry (Z2 = 1) whose 2d*x2*y2 is precomputed:
// 7 multiplications (curve25519.rs:712-729 add_cached, with Z2 = 1 folded in).
template <class C>
ECCX_DEV void ed_add_cached(EdPt<C>& r, const EdPt<C>& p, const Fe<C::L>& x2, const Fe<C::L>& y2,
                            const Fe<C::L>& t2d) {
  using F = Fe<C::L>;
  F aa, bb, cc, dd, e, f, g, h, u, v;
  S_(u, p.y, p.x); S_(v, y2, x2); M_(aa, u, v);
  A_(u, p.y, p.x); A_(v, y2, x2); M_(bb, u, v);
  M_(cc, p.t, t2d);
  A_(dd, p.z, p.z);
  S_(e, bb, aa); S_(f, dd, cc); A_(g, dd, cc); A_(h, bb, aa);
  M_(r.x, e, f); M_(r.y, g, h); M_(r.z, f, g); M_(r.t, e, h);
}

template <class C>
ECCX_DEV void ed_dbl(EdPt<C>& r, const EdPt<C>& p) {
  using F = Fe<C::L>;
  F a, b, c, d, e, f, g, h, xy, ab;
  fe_sqr<C>(a, p.x); fe_sqr<C>(b, p.y); fe_sqr<C>(c, p.z); A_(c, c, c);
  fe_neg<C>(d, a);
  A_(xy, p.x, p.y); fe_sqr<C>(xy, xy); A_(ab, a, b); S_(e, xy, ab);
  A_(g, d, b); S_(f, g, c); S_(h, d, b);
  M_(r.x, e, f); M_(r.y, g, h); M_(r.z, f, g); M_(r.t, e, h);
}

#undef M_
#undef A_
#undef S_

}  // namespace eccx
