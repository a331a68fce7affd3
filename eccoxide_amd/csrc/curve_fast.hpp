// Fast (non-mirroring) point arithmetic: Jacobian coordinates (x = X/Z^2, y = Y/Z^3).
//
// The reference's ladder (src/curve/projective.rs:871-896) uses the complete RCB formulas in
// homogeneous coordinates: 13 field mul per doubling, 14 per addition, and no special
// cases.  The result a caller can observe is the AFFINE point (Point::to_affine,
// projective.rs:666-682), which does not depend on the coordinate system, so the default
// kernels use the cheaper Jacobian formulas -- 8 mul per doubling (a = -3; 7 for a = 0),
// 14 per addition against a table entry that caches Z^2 and Z^3 -- and handle the cases
// those formulas do not cover (either operand at infinity, P + P, P + (-P)) explicitly, so
// every input the reference accepts gives the same affine bytes and infinity flag.  The
// reference-mirroring kernels (kernels.hpp) remain available for un-normalised (X:Y:Z)
// parity.
#pragma once
#include "curve.hpp"

namespace eccx {

template <class C>
struct Jac {
  Fe<C::L> x, y, z;  // infinity <=> z == 0
};

// table entry: a Jacobian point with Z^2 and Z^3 cached
template <class C>
struct JacEntry {
  Fe<C::L> x, y, z, zz, zzz;
};

#define M_(r, a, b) fe_mul<C>(r, a, b)
#define Q_(r, a) fe_sqr<C>(r, a)
#define A_(r, a, b) fe_add<C>(r, a, b)
#define S_(r, a, b) fe_sub<C>(r, a, b)

// 2P.  a = -3: dbl-2001-b with Z3 = 2YZ (8 mul); a = 0: dbl-2009-l (7 mul).
// Z = 0 stays Z = 0.
template <class C>
ECCX_DEV void jac_dbl(Jac<C>& r, const Jac<C>& p) {
  using F = Fe<C::L>;
  if constexpr (C::A0) {
    F a, b, c, d, e, f, t, x3, y3, z3;
    Q_(a, p.x); Q_(b, p.y); Q_(c, b);
    A_(t, p.x, b); Q_(t, t); S_(t, t, a); S_(t, t, c); A_(d, t, t);
    A_(e, a, a); A_(e, e, a);
    Q_(f, e);
    A_(t, d, d); S_(x3, f, t);
    M_(z3, p.y, p.z); A_(z3, z3, z3);
    S_(t, d, x3); M_(y3, e, t);
    A_(c, c, c); A_(c, c, c); A_(c, c, c); S_(y3, y3, c);
    r.x = x3; r.y = y3; r.z = z3;
  } else {
    // small multiples are folded into multiplicands (2*gamma, 4*gamma) so that 4*beta and
    // 8*gamma^2 cost one field addition each instead of two and three: 13 add/sub per doubling
    F delta, gamma, g2, g4, alpha, t1, t2, b4, x3, y3, z3;
    Q_(delta, p.z); Q_(gamma, p.y);
    A_(g2, gamma, gamma); A_(g4, g2, g2);
    M_(b4, p.x, g4);                                    // 4*beta = X * 4*gamma
    S_(t1, p.x, delta); A_(t2, p.x, delta); M_(t1, t1, t2);
    A_(alpha, t1, t1); A_(alpha, alpha, t1);
    Q_(x3, alpha); S_(x3, x3, b4); S_(x3, x3, b4);      // alpha^2 - 8*beta
    M_(z3, p.y, p.z); A_(z3, z3, z3);
    Q_(t2, g2); A_(t2, t2, t2);                         // 8*gamma^2 = 2 * (2*gamma)^2
    S_(t1, b4, x3); M_(y3, alpha, t1); S_(y3, y3, t2);
    r.x = x3; r.y = y3; r.z = z3;
  }
}

// r = p + e with the generic formulas (add-1998-cmo-2 with cached Z2^2, Z2^3; 14 mul).
// Valid only when neither operand is infinity and p != +-e; the caller inspects
// h_zero (same x) / r_zero (same y) and the operands' Z to patch the other cases.
template <class C>
ECCX_DEV void jac_add_raw(Jac<C>& r, bool& h_zero, bool& r_zero, const Jac<C>& p, const JacEntry<C>& e) {
  using F = Fe<C::L>;
  F z1z1, u1, u2, s1, s2, h, rr, hh, hhh, v, t, x3, y3, z3;
  Q_(z1z1, p.z);
  M_(u1, p.x, e.zz);
  M_(u2, e.x, z1z1);
  M_(s1, p.y, e.zzz);
  M_(t, p.z, z1z1); M_(s2, e.y, t);
  S_(h, u2, u1);
  S_(rr, s2, s1);
  h_zero = fe_is_zero<C>(h);
  r_zero = fe_is_zero<C>(rr);
  Q_(hh, h); M_(hhh, h, hh); M_(v, u1, hh);
  Q_(x3, rr); S_(x3, x3, hhh); S_(x3, x3, v); S_(x3, x3, v);
  S_(t, v, x3); M_(y3, rr, t); M_(t, s1, hhh); S_(y3, y3, t);
  M_(z3, p.z, e.z); M_(z3, z3, h);
  r.x = x3; r.y = y3; r.z = z3;
}

// r = p + (x2, y2, 1): mixed addition with an affine operand (11 mul).  Same validity
// conditions and h_zero / r_zero reporting as jac_add_raw.
template <class C>
ECCX_DEV void jac_madd_raw(Jac<C>& r, bool& h_zero, bool& r_zero, const Jac<C>& p, const Fe<C::L>& x2,
                           const Fe<C::L>& y2) {
  using F = Fe<C::L>;
  F z1z1, u2, s2, h, rr, hh, hhh, v, t, x3, y3, z3;
  Q_(z1z1, p.z);
  M_(u2, x2, z1z1);
  M_(t, p.z, z1z1); M_(s2, y2, t);
  S_(h, u2, p.x);
  S_(rr, s2, p.y);
  h_zero = fe_is_zero<C>(h);
  r_zero = fe_is_zero<C>(rr);
  Q_(hh, h); M_(hhh, h, hh); M_(v, p.x, hh);
  Q_(x3, rr); S_(x3, x3, hhh); S_(x3, x3, v); S_(x3, x3, v);
  S_(t, v, x3); M_(y3, rr, t); M_(t, p.y, hhh); S_(y3, y3, t);
  M_(z3, p.z, h);
  r.x = x3; r.y = y3; r.z = z3;
}

template <class C>
ECCX_DEV void jac_select(Jac<C>& r, bool take_a, const Jac<C>& a, const Jac<C>& b) {
  fe_select<C>(r.x, take_a, a.x, b.x);
  fe_select<C>(r.y, take_a, a.y, b.y);
  fe_select<C>(r.z, take_a, a.z, b.z);
}

#undef M_
#undef Q_
#undef A_
#undef S_

}  // namespace eccx
