// BN254 base-field arithmetic in a register-friendly internal form for the MSM inner loop.
//
// Why a second representation.  On gfx950 v_mad_u64_u32 issues at half the rate of a 32-bit add
// (profiles/r01_instr_rate.log), but so do the 64-bit adds and carry-chain adds that a radix-2^32
// Montgomery product needs around every mad; the compiler output for ff.h spends ~60 % of its
// issue slots on v_mov_b32 / v_lshl_add_u64 glue.  With 29-bit limbs nine limbs cover 261 bits and
// a column of 9 + 9 partial products fits a signed 64-bit accumulator with room to spare, so a
// product is a plain chain of v_mad_i64_i32 into ONE accumulator per column plus a mask and a
// shift: no carries, no register-pair shuffles.  Additions and subtractions are limb-wise with no
// carry chain at all.
//
// Representation (F29): value = sum v[i] * 2^(29 i), limbs signed 32-bit.
//   "normalised": v[0..7] in [0, 2^29), v[8] a small signed integer carrying the sign.
//   Montgomery domain R' = 2^261: x is held as x * 2^261 mod p, as an integer in (-8p, 8p).
// Contracts:
//   mul(a, b): |a_i * b_j| column sums must stay below 2^63: both normalised or differences of two
//              normalised values (|limb| < 2^29), or one operand with |limb| < 2^30;
//              |a * b| < 64 p^2.  Result normalised, value in (-p/2, 3p/2).
//   add/sub/neg: limb-wise, no normalisation.  norm(): carry propagation.
// The memory image (tables, kernel outputs) stays 8 x u32: canonical value in [0, p).
#pragma once
#include "ff.h"

namespace zk {

struct Fq29Params {
  static ZK_HD int32_t p(int i) {
    constexpr int32_t v[9] = {0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978,
                              0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
    return v[i];
  }
  static constexpr uint32_t inv = 0x04866389u;  // -p^-1 mod 2^29
  static ZK_HD int32_t one(int i) {             // 2^261 mod p
    constexpr int32_t v[9] = {0x157ccc21, 0x141c2758, 0x185230d3, 0x014c0419, 0x0aa36fb9,
                              0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836};
    return v[i];
  }
  static ZK_HD int32_t c256(int i) {  // 2^256 mod p: F29 -> standard Montgomery (R = 2^256)
    constexpr int32_t v[9] = {0x058f0d9d, 0x1aea1c6e, 0x11c2cf74, 0x11d651eb, 0x1462c0a7,
                              0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77};
    return v[i];
  }
  static ZK_HD int32_t c266(int i) {  // 2^266 mod p: standard Montgomery -> F29
    constexpr int32_t v[9] = {0x13349ca1, 0x1a5d84a8, 0x0a3e5cac, 0x100249e0, 0x12b951e8,
                              0x0e92d304, 0x14cb95b3, 0x041b9d3d, 0x00058003};
    return v[i];
  }
  // 2^261 mod p as a plain 8 x u32 integer: ff.h mul(x~, K261) maps x*2^256 -> x*2^261, canonical
  static ZK_HD uint32_t k261(int i) {
    constexpr uint32_t v[8] = {0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u,
                               0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
    return v[i];
  }
  typedef FqParams Std;
};

// The scalar field r in the same form (used by the quotient NTTs, ntt.hip)
struct Fr29Params {
  static ZK_HD int32_t p(int i) {
    constexpr int32_t v[9] = {0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283,
                              0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
    return v[i];
  }
  static constexpr uint32_t inv = 0x0fffffffu;  // -r^-1 mod 2^29
  static ZK_HD int32_t one(int i) {             // 2^261 mod r
    constexpr int32_t v[9] = {0x0fffff57, 0x1ea70ab4, 0x052c068b, 0x17504f49, 0x0aa8075b,
                              0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836};
    return v[i];
  }
  static ZK_HD int32_t c256(int i) {  // 2^256 mod r
    constexpr int32_t v[9] = {0x0ffffffb, 0x04b1a0e2, 0x18334a6b, 0x18ed2b3e, 0x1462e36f,
                              0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77};
    return v[i];
  }
  static ZK_HD int32_t c266(int i) {  // 2^266 mod r
    constexpr int32_t v[9] = {0x0fffead7, 0x1d5444f4, 0x04438aa5, 0x03b4d096, 0x134c84da,
                              0x0e92d304, 0x14cb95b3, 0x041b9d3d, 0x00058003};
    return v[i];
  }
  static ZK_HD uint32_t k261(int i) {  // 2^261 mod r as a plain 8 x u32 integer
    constexpr uint32_t v[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u,
                               0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
    return v[i];
  }
  typedef FrParams Std;
};

template <class P>
struct F29 {
  int32_t v[9];
  static constexpr int32_t MASK = (1 << 29) - 1;

  static ZK_HD F29 zero() {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
  }
  static ZK_HD F29 one() {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = P::one(i);
    return r;
  }
};

template <class P>
ZK_HD F29<P> add(const F29<P>& a, const F29<P>& b) {
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <class P>
ZK_HD F29<P> sub(const F29<P>& a, const F29<P>& b) {
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
template <class P>
ZK_HD F29<P> neg(const F29<P>& a) {
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = -a.v[i];
  return r;
}
// a if !c else -a, branch-free
template <class P>
ZK_HD F29<P> cneg(const F29<P>& a, bool c) {
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c ? -a.v[i] : a.v[i];
  return r;
}

// carry propagation: limbs 0..7 into [0, 2^29), sign stays in limb 8
template <class P>
ZK_HD F29<P> norm(const F29<P>& a) {
  F29<P> r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int32_t t = a.v[i] + c;
    r.v[i] = t & F29<P>::MASK;
    c = t >> 29;  // arithmetic
  }
  r.v[8] = a.v[8] + c;
  return r;
}

// Montgomery product, R' = 2^261, product scanning: one signed 64-bit accumulator per column.
template <class P>
ZK_HD F29<P> mul(const F29<P>& a, const F29<P>& b) {
  int64_t acc = 0;
  int32_t m[9];
  F29<P> r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P::p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * P::inv) & (uint32_t)F29<P>::MASK);
    acc += (int64_t)m[k] * P::p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P::p(k - i);
    r.v[k - 9] = (int32_t)acc & F29<P>::MASK;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// The same product scheduled for a LONE wavefront (witness solver: one wave per SIMD, nothing to
// hide the latency of mul()'s 171 dependent mads): 17 independent column accumulators, then nine
// reduction steps whose nine mads each are independent again.  Same partial products, same column
// bound, bit-identical result; 34 accumulator registers instead of 2.
template <class P>
ZK_HD F29<P> mul_ilp(const F29<P>& a, const F29<P>& b) {
  int64_t col[17];
#pragma unroll
  for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) col[i + j] += (int64_t)a.v[i] * b.v[j];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t m = (int32_t)(((uint32_t)col[k] * P::inv) & (uint32_t)F29<P>::MASK);
#pragma unroll
    for (int j = 0; j < 9; j++) col[k + j] += (int64_t)m * P::p(j);
    col[k + 1] += col[k] >> 29;
  }
  F29<P> r;
#pragma unroll
  for (int k = 9; k < 16; k++) {
    r.v[k - 9] = (int32_t)col[k] & F29<P>::MASK;
    col[k + 1] += col[k] >> 29;
  }
  r.v[7] = (int32_t)col[16] & F29<P>::MASK;
  r.v[8] = (int32_t)(col[16] >> 29);
  return r;
}

// Montgomery square: cross terms once, doubled (45 limb products instead of 81)
template <class P>
ZK_HD F29<P> sqr(const F29<P>& a) {
  int64_t acc = 0;
  int32_t m[9], a2[9];
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.v[i] * 2;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    const int lo = k < 9 ? 0 : k - 8;
    const int hi = k < 9 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (i < j)
        acc += (int64_t)a2[i] * a.v[j];
      else if (i == j)
        acc += (int64_t)a.v[i] * a.v[i];
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P::p(k - i);
      m[k] = (int32_t)(((uint32_t)acc * P::inv) & (uint32_t)F29<P>::MASK);
      acc += (int64_t)m[k] * P::p(0);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P::p(k - i);
      r.v[k - 9] = (int32_t)acc & F29<P>::MASK;
    }
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// value == 0 mod p for a normalised mul/sqr output (value in (-p/2, 3p/2)): it is 0 or p
template <class P>
ZK_HD bool is_zero_mulout(const F29<P>& a) {
  int32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ P::p(i);
  }
  return z == 0 || e == 0;
}

// ---- memory image <-> limbs ------------------------------------------------------------------
// 8 x u32 non-negative integer < 2^256 -> nine 29-bit limbs (no domain change)
template <class P>
ZK_HD F29<P> unpack29(const uint32_t w[8]) {
  F29<P> r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i;
    const int q = bit >> 5, s = bit & 31;
    uint32_t lo = w[q] >> s;
    if (s > 3 && q + 1 < 8) lo |= w[q + 1] << (32 - s);
    r.v[i] = (int32_t)(lo & (uint32_t)F29<P>::MASK);
  }
  return r;
}

// normalised value in (-p, 2p) -> canonical [0, p) packed 8 x u32
template <class P>
ZK_HD void pack_canonical(uint32_t w[8], const F29<P>& a) {
  F29<P> t = a;
  // add p if negative
  if (t.v[8] < 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) t.v[i] += P::p(i);
    t = norm(t);
  }
  // subtract p if >= p
  F29<P> d;
#pragma unroll
  for (int i = 0; i < 9; i++) d.v[i] = t.v[i] - P::p(i);
  d = norm(d);
  if (d.v[8] >= 0) t = d;
  uint64_t acc = 0;
  int have = 0, wi = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    acc |= (uint64_t)(uint32_t)t.v[i] << have;
    have += 29;
    if (have >= 32) {
      w[wi++] = (uint32_t)acc;
      acc >>= 32;
      have -= 32;
    }
  }
  if (wi < 8) w[wi] = (uint32_t)acc;
}

// standard Montgomery element (R = 2^256, canonical) -> F29 domain (R' = 2^261)
template <class P>
ZK_HD F29<P> from_std(const Fp<typename P::Std>& x) {
  F29<P> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.v[i] = P::c266(i);
  return mul(unpack29<P>(x.v), c);
}
// F29 (|value| < 8p) -> standard Montgomery canonical
template <class P>
ZK_HD Fp<typename P::Std> to_std(const F29<P>& a) {
  F29<P> c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.v[i] = P::c256(i);
  Fp<typename P::Std> r;
  pack_canonical<P>(r.v, mul(a, c));
  return r;
}

// (a*b + c*d) / 2^261: two products share one Montgomery reduction ("lazy reduction").
// Column bound: 18 + 9 partial products below 2^58 each -> all four operands need |limb| < 2^29.
// |a*b + c*d| < 64 p^2.  Result normalised, value in (-p/2, 3p/2).
template <class P>
ZK_HD F29<P> mul_add2(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) {
  int64_t acc = 0;
  int32_t m[9];
  F29<P> r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)c.v[i] * d.v[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; i++) acc += (int64_t)m[i] * P::p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * P::inv) & (uint32_t)F29<P>::MASK);
    acc += (int64_t)m[k] * P::p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i < 9; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)c.v[i] * d.v[k - i];
    }
#pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (int64_t)m[i] * P::p(k - i);
    r.v[k - 9] = (int32_t)acc & F29<P>::MASK;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// weak reduction of a value with |v| < 2^7 p (limbs below 2^31 in magnitude): subtract the
// multiple of p estimated from the top limb; result normalised with |v| < 0.6 p.
template <class P>
ZK_HD F29<P> wred(const F29<P>& a) {
  const F29<P> v = norm(a);
  const float ptop = (float)P::p(8) + 0.5f;
  const int32_t q = (int32_t)__builtin_rintf((float)v.v[8] / ptop);
  F29<P> r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (int64_t)v.v[i] - (int64_t)q * P::p(i);
    r.v[i] = (int32_t)c & F29<P>::MASK;
    c >>= 29;
  }
  r.v[8] = (int32_t)(c + (int64_t)v.v[8] - (int64_t)q * P::p(8));
  return r;
}

typedef F29<Fq29Params> Fq29;
typedef F29<Fr29Params> Fr29;

// ---- Fq2 = Fq[u]/(u^2+1) on the lazy representation -----------------------------------------
// Components are kept normalised; products use one shared reduction per component.
struct Fq2_29 {
  Fq29 c0, c1;
  static ZK_HD Fq2_29 one() { return Fq2_29{Fq29::one(), Fq29::zero()}; }
  static ZK_HD Fq2_29 zero() { return Fq2_29{Fq29::zero(), Fq29::zero()}; }
};
ZK_HD Fq2_29 add(const Fq2_29& a, const Fq2_29& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_HD Fq2_29 sub(const Fq2_29& a, const Fq2_29& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZK_HD Fq2_29 cneg(const Fq2_29& a, bool c) { return {cneg(a.c0, c), cneg(a.c1, c)}; }
ZK_HD Fq2_29 norm(const Fq2_29& a) { return {norm(a.c0), norm(a.c1)}; }
ZK_HD Fq2_29 wred(const Fq2_29& a) { return {wred(a.c0), wred(a.c1)}; }
// operands: |limb| < 2^29 (normalised values or differences of two)
ZK_HD Fq2_29 mul(const Fq2_29& a, const Fq2_29& b) {
  return {mul_add2(a.c0, b.c0, neg(a.c1), b.c1), mul_add2(a.c0, b.c1, a.c1, b.c0)};
}
ZK_HD Fq2_29 sqr(const Fq2_29& a) {
  // (a0+a1)(a0-a1) + 2 a0 a1 u ; the sum is the one operand allowed |limb| < 2^30
  return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), mul(add(a.c0, a.c0), a.c1)};
}
ZK_HD bool is_zero_mulout(const Fq2_29& a) { return is_zero_mulout(a.c0) && is_zero_mulout(a.c1); }
ZK_HD Fq2 to_std(const Fq2_29& a) { return Fq2{to_std(a.c0), to_std(a.c1)}; }
ZK_HD Fq2_29 unpack2_29(const Fq2& x) {
  return {unpack29<Fq29Params>(x.c0.v), unpack29<Fq29Params>(x.c1.v)};
}

}  // namespace zk
