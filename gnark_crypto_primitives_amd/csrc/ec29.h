// G1 mixed addition on the 9 x 29-bit lazy representation (ff29.h) for the MSM inner loop.
// Same formulas as ec.h (EFD madd-2008-s / mdbl-2008-s-1); the comments track the limb and value
// bounds that ff29.h's mul/sqr contracts need.
#pragma once
#include "ec.h"
#include "ff29.h"
#if defined(__HIP_DEVICE_COMPILE__)
#include "ff29_asm.h"
#endif

namespace zk {

// Field products of the MSM inner loop: on the device the single-accumulator asm chains of
// ff29_asm.h (no v_lshl_add_u64 merges: 2192 -> ~2010 instructions per G1 mixed addition), on the
// host (tests/native/test_ff29.cpp) the C++ forms they restate.
#if defined(__HIP_DEVICE_COMPILE__)
template <class P> ZK_HD F29<P> mmul(const F29<P>& a, const F29<P>& b) { return mul_asm(a, b); }
template <class P> ZK_HD F29<P> msqr(const F29<P>& a) { return sqr_asm(a); }
template <class P>
ZK_HD F29<P> mmul_add2(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) {
  return mul_add2_asm(a, b, c, d);
}
#else
template <class P> ZK_HD F29<P> mmul(const F29<P>& a, const F29<P>& b) { return mul(a, b); }
template <class P> ZK_HD F29<P> msqr(const F29<P>& a) { return sqr(a); }
template <class P>
ZK_HD F29<P> mmul_add2(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) {
  return mul_add2(a, b, c, d);
}
#endif

// Invariants between calls: x, y normalised with |x| < 5p, |y| < 2p; zz, zzz mul outputs.
struct G1Acc29 {
  Fq29 x, y, zz, zzz;
  bool inf;
  static ZK_HD G1Acc29 infinity() {
    G1Acc29 a;
    a.x = a.y = a.zz = a.zzz = Fq29::zero();
    a.inf = true;
    return a;
  }
};

// 2 * (qx, qy), affine input with |limbs| < 2^29
ZK_HD void mdbl29(G1Acc29& acc, const Fq29& qx, const Fq29& qy) {
  const Fq29 u = norm(add(qy, qy));
  const Fq29 v = sqr(u);
  const Fq29 w = mul(u, v);
  const Fq29 s = mul(qx, v);
  const Fq29 x2 = sqr(qx);
  const Fq29 m = norm(add(add(x2, x2), x2));
  const Fq29 x3 = norm(sub(sqr(m), add(s, s)));
  const Fq29 y3 = norm(sub(mul(m, sub(s, x3)), mul(w, qy)));
  acc.x = x3;
  acc.y = y3;
  acc.zz = v;
  acc.zzz = w;
  acc.inf = false;
}

// acc += (qx, qy); the addend is a finite point, canonical x in [0,p), y possibly negated
// limb-wise (|limbs| < 2^29).
// The accumulator is loop-carried in the MSM kernels.  LLVM proves at IR level that its masked limbs
// are non-negative and rewrites their sign extensions as zero extensions, but instruction selection
// works per basic block and cannot see that proof: a product of such a limb with a signed limb is
// then expanded into two v_mad_u64_u32 plus two moves instead of one v_mad_i64_i32 (96 extra mads
// and 192 moves per G1 mixed addition).  Passing the limbs through an empty asm hides the range, so
// every product stays a signed 32 x 32 -> 64 multiply-add.
ZK_HD void opaque_limbs(Fq29& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 9; i++) asm volatile("" : "+v"(a.v[i]));
#endif
}

ZK_HD void madd29(G1Acc29& acc, const Fq29& qx, const Fq29& qy) {
#if !defined(__HIP_DEVICE_COMPILE__)
  opaque_limbs(acc.x);
  opaque_limbs(acc.y);
  opaque_limbs(acc.zz);
  opaque_limbs(acc.zzz);
#endif
  if (acc.inf) {
    acc.x = qx;
    acc.y = norm(qy);
    acc.zz = acc.zzz = Fq29::one();
    acc.inf = false;
    return;
  }
  const Fq29 u2 = mmul(qx, acc.zz);
  const Fq29 s2 = mmul(qy, acc.zzz);
  const Fq29 p = sub(u2, acc.x);  // |limbs| < 2^29, |value| < 6.5p
  const Fq29 r = sub(s2, acc.y);  // |value| < 3.5p
  const Fq29 pp = msqr(p);
  const Fq29 rr = msqr(r);
  if (is_zero_mulout(pp)) {  // same x: doubling or cancellation (never on honest random data)
    if (is_zero_mulout(rr))
      mdbl29(acc, qx, qy);
    else
      acc.inf = true;
    return;
  }
  const Fq29 ppp = mmul(p, pp);
  const Fq29 q = mmul(acc.x, pp);
  const Fq29 x3 = norm(sub(sub(rr, ppp), add(q, q)));  // (-5p, 3p)
  // y3 = r*(q - x3) - y1*ppp with ONE Montgomery reduction for both products (saves 90 of the
  // 342 mads and the normalisation); |r (q-x3)| + |y1 ppp| < 26 p^2, result in (-p/2, 3p/2)
  const Fq29 y3 = mmul_add2(r, sub(q, x3), neg(acc.y), ppp);
  acc.zz = mmul(acc.zz, pp);
  acc.zzz = mmul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

// ---- G2 (Fq2 components, lazy reduction) -------------------------------------------------------
// ff29.h's Fq2 product and square on the products above
ZK_HD Fq2_29 mmul(const Fq2_29& a, const Fq2_29& b) {
  return {mmul_add2(a.c0, b.c0, neg(a.c1), b.c1), mmul_add2(a.c0, b.c1, a.c1, b.c0)};
}
ZK_HD Fq2_29 msqr(const Fq2_29& a) {
  return {mmul(add(a.c0, a.c1), sub(a.c0, a.c1)), mmul(add(a.c0, a.c0), a.c1)};
}

// Invariants between calls: x, y weakly reduced (|component| < 0.6p); zz, zzz products.
struct G2Acc29 {
  Fq2_29 x, y, zz, zzz;
  bool inf;
  static ZK_HD G2Acc29 infinity() {
    G2Acc29 a;
    a.x = a.y = a.zz = a.zzz = Fq2_29::zero();
    a.inf = true;
    return a;
  }
};

ZK_HD void mdbl29(G2Acc29& acc, const Fq2_29& qx, const Fq2_29& qy) {
  const Fq2_29 u = norm(add(qy, qy));
  const Fq2_29 v = sqr(u);
  const Fq2_29 w = mul(u, v);
  const Fq2_29 s = mul(qx, v);
  const Fq2_29 x2 = sqr(qx);
  const Fq2_29 m = norm(add(add(x2, x2), x2));
  const Fq2_29 x3 = wred(sub(sqr(m), add(s, s)));
  const Fq2_29 y3 = wred(sub(mul(m, sub(s, x3)), mul(w, qy)));
  acc.x = x3;
  acc.y = y3;
  acc.zz = v;
  acc.zzz = w;
  acc.inf = false;
}

// qx canonical components, qy canonical or limb-wise negated
ZK_HD void madd29(G2Acc29& acc, const Fq2_29& qx, const Fq2_29& qy) {
#if !defined(__HIP_DEVICE_COMPILE__)
  opaque_limbs(acc.x.c0);    // see the G1 madd29
  opaque_limbs(acc.x.c1);
  opaque_limbs(acc.y.c0);
  opaque_limbs(acc.y.c1);
  opaque_limbs(acc.zz.c0);
  opaque_limbs(acc.zz.c1);
  opaque_limbs(acc.zzz.c0);
  opaque_limbs(acc.zzz.c1);
#endif
  if (acc.inf) {
    acc.x = qx;
    acc.y = norm(qy);
    acc.zz = acc.zzz = Fq2_29::one();
    acc.inf = false;
    return;
  }
  const Fq2_29 u2 = mmul(qx, acc.zz);
  const Fq2_29 s2 = mmul(qy, acc.zzz);
  const Fq2_29 p = sub(u2, acc.x);  // |component| < 1.7p, |limb| < 2^29
  const Fq2_29 r = sub(s2, acc.y);
  const Fq2_29 pp = msqr(p);
  const Fq2_29 rr = msqr(r);
  if (is_zero_mulout(pp)) {
    if (is_zero_mulout(rr))
      mdbl29(acc, qx, qy);
    else
      acc.inf = true;
    return;
  }
  const Fq2_29 ppp = mmul(p, pp);
  const Fq2_29 q = mmul(acc.x, pp);
  const Fq2_29 x3 = wred(sub(sub(rr, ppp), add(q, q)));
  const Fq2_29 y3 = wred(sub(mmul(r, sub(q, x3)), mmul(acc.y, ppp)));
  acc.zz = mmul(acc.zz, pp);
  acc.zzz = mmul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

ZK_HD G2XYZZ to_std(const G2Acc29& a) {
  if (a.inf) return G2XYZZ::inf();
  return G2XYZZ{to_std(a.x), to_std(a.y), to_std(a.zz), to_std(a.zzz)};
}

ZK_HD G1XYZZ to_std(const G1Acc29& a) {
  if (a.inf) return G1XYZZ::inf();
  return G1XYZZ{to_std(a.x), to_std(a.y), to_std(a.zz), to_std(a.zzz)};
}

}  // namespace zk
