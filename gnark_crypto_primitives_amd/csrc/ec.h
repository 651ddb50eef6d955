// BN254 G1 (over Fq) and G2 (over Fq2) group law, extended-Jacobian ("XYZZ") accumulators with
// affine addends.  x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.
//
// Affine points use the memory image of gnark-crypto's G1Affine{X,Y fp.Element} (64 B) and
// G2Affine{X,Y E2{A0,A1}} (128 B) with infinity encoded as (0,0).
// Formulas: Explicit-Formulas Database, short Weierstrass a=0, "xyzz" coordinates
// (madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1).  gnark-crypto's multiexp uses the same
// coordinate system for its bucket accumulators (g1JacExtended), SURVEY.md §3.2.
#pragma once
#include "ff.h"

namespace zk {

template <class F>
struct Affine {
  F x, y;
  ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  static ZK_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ZK_HD bool is_inf() const { return zz.is_zero(); }
  static ZK_HD XYZZ inf() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
  static ZK_HD XYZZ from_affine(const Affine<F>& a) {
    if (a.is_inf()) return inf();
    return XYZZ{a.x, a.y, F::one(), F::one()};
  }
};

// 2*(affine)  (mdbl-2008-s-1)
template <class F>
ZK_HD XYZZ<F> dbl_affine(const Affine<F>& a) {
  if (a.is_inf()) return XYZZ<F>::inf();
  F u = dbl(a.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(a.x, v);
  F x2 = sqr(a.x);
  F m = add(dbl(x2), x2);
  XYZZ<F> r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, a.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2*p  (dbl-2008-s-1)
template <class F>
ZK_HD XYZZ<F> dbl(const XYZZ<F>& p) {
  if (p.is_inf()) return p;
  F u = dbl(p.y);
  F v = sqr(u);
  F w = mul(u, v);
  F s = mul(p.x, v);
  F x2 = sqr(p.x);
  F m = add(dbl(x2), x2);
  XYZZ<F> r;
  r.x = sub(sqr(m), dbl(s));
  r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
  r.zz = mul(v, p.zz);
  r.zzz = mul(w, p.zzz);
  return r;
}

// acc += q (affine, not infinity unless flagged)  (madd-2008-s), all special cases handled
template <class F>
ZK_HD void madd(XYZZ<F>& acc, const Affine<F>& q) {
  if (q.is_inf()) return;
  if (acc.is_inf()) {
    acc = XYZZ<F>{q.x, q.y, F::one(), F::one()};
    return;
  }
  F u2 = mul(q.x, acc.zz);
  F s2 = mul(q.y, acc.zzz);
  F p = sub(u2, acc.x);
  F r = sub(s2, acc.y);
  if (p.is_zero()) {
    if (r.is_zero())
      acc = dbl_affine(q);
    else
      acc = XYZZ<F>::inf();
    return;
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F qq = mul(acc.x, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(qq));
  F y3 = sub(mul(r, sub(qq, x3)), mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(acc.zz, pp);
  acc.zzz = mul(acc.zzz, ppp);
}

// acc += b  (add-2008-s)
template <class F>
ZK_HD void padd(XYZZ<F>& acc, const XYZZ<F>& b) {
  if (b.is_inf()) return;
  if (acc.is_inf()) {
    acc = b;
    return;
  }
  F u1 = mul(acc.x, b.zz);
  F u2 = mul(b.x, acc.zz);
  F s1 = mul(acc.y, b.zzz);
  F s2 = mul(b.y, acc.zzz);
  F p = sub(u2, u1);
  F r = sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero())
      acc = dbl(acc);
    else
      acc = XYZZ<F>::inf();
    return;
  }
  F pp = sqr(p);
  F ppp = mul(p, pp);
  F qq = mul(u1, pp);
  F x3 = sub(sub(sqr(r), ppp), dbl(qq));
  F y3 = sub(mul(r, sub(qq, x3)), mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = mul(mul(acc.zz, b.zz), pp);
  acc.zzz = mul(mul(acc.zzz, b.zzz), ppp);
}

template <class F>
ZK_HD Affine<F> neg(const Affine<F>& a) {
  return Affine<F>{a.x, neg(a.y)};
}

template <class F>
ZK_HD Affine<F> to_affine(const XYZZ<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  // one inversion serves both coordinates: (ZZ/ZZZ)^2 = ZZ^2/ZZ^3 = 1/ZZ
  F izzz = inverse(p.zzz);
  F izz = sqr(mul(izzz, p.zz));
  return Affine<F>{mul(p.x, izz), mul(p.y, izzz)};
}

// k * q for a 256-bit scalar in canonical (non-Montgomery) 8x32 form; MSB-first double-and-add.
template <class F>
ZK_HD XYZZ<F> scalar_mul(const Affine<F>& q, const uint32_t k[8]) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int i = 7; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      acc = dbl(acc);
      if ((k[i] >> b) & 1) madd(acc, q);
    }
  }
  return acc;
}

typedef Affine<Fq> G1Affine;
typedef Affine<Fq2> G2Affine;
typedef XYZZ<Fq> G1XYZZ;
typedef XYZZ<Fq2> G2XYZZ;

}  // namespace zk
