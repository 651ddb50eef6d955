// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Representation: 8 x 32-bit little-endian limbs, Montgomery form with R = 2^256, value always
// fully reduced to [0, p).  Byte-for-byte this is the memory image of gnark-crypto's
// fr.Element / fp.Element ([4]uint64 little-endian limbs, Montgomery R = 2^256;
// SURVEY.md §3.2, go.mod:9 pins gnark-crypto v0.19.3-0.20251115174214-022ec58e8c19), so buffers
// cross the C-ABI without conversion.
//
// The vector ALU of CDNA4 has no 64x64 multiplier; the widest integer multiply-add is
// v_mad_u64_u32 (32x32+64 -> 64).  Every product below is written as (u64)a*b + c so that
// hipcc selects it.  No MFMA: this is 254-bit integer work.
#pragma once
#include <stdint.h>

#include "modinv30.h"

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

// ---- moduli -----------------------------------------------------------------------------
// Scalar field r (fr) and base field p (fp) of BN254 (values cross-checked in SURVEY.md §8c K6).
struct FrParams {
  static ZK_HD uint32_t p(int i) {
    constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return v[i];
  }
  static ZK_HD uint32_t one(int i) {  // R mod r
    constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                               0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return v[i];
  }
  static ZK_HD uint32_t r2(int i) {  // R^2 mod r
    constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                               0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    return v[i];
  }
  static constexpr uint32_t inv = 0xefffffffu;  // -r^-1 mod 2^32
};

struct FqParams {
  static ZK_HD uint32_t p(int i) {
    constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    return v[i];
  }
  static ZK_HD uint32_t one(int i) {  // R mod p
    constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                               0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    return v[i];
  }
  static ZK_HD uint32_t r2(int i) {
    constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                               0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    return v[i];
  }
  static constexpr uint32_t inv = 0xe4866389u;  // -p^-1 mod 2^32
};

// ---- element ----------------------------------------------------------------------------
template <class P>
struct alignas(16) Fp {
  uint32_t v[8];

  static ZK_HD Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
  }
  static ZK_HD Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::one(i);
    return r;
  }
  ZK_HD bool is_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i];
    return o == 0;
  }
  ZK_HD bool operator==(const Fp& b) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
  ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - p if a >= p (a < 2p assumed, carry = bit 256 of a)
template <class P>
ZK_HD void reduce_once(uint32_t r[8], const uint32_t a[8], uint32_t carry) {
  uint32_t d[8];
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int64_t t = (int64_t)a[i] - (int64_t)P::p(i) + br;
    d[i] = (uint32_t)t;
    br = t >> 32;  // 0 or -1
  }
  // a >= p  <=>  no borrow out, or the addition overflowed 2^256
  bool ge = (br == 0) || carry;
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = ge ? d[i] : a[i];
}

template <class P>
ZK_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
  uint32_t t[8];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    t[i] = (uint32_t)c;
    c >>= 32;
  }
  Fp<P> r;
  reduce_once<P>(r.v, t, (uint32_t)c);  // p < 2^254 so c is always 0; kept for generality
  return r;
}

template <class P>
ZK_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
  uint32_t t[8];
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int64_t d = (int64_t)a.v[i] - (int64_t)b.v[i] + br;
    t[i] = (uint32_t)d;
    br = d >> 32;
  }
  // if borrow, add p back
  uint32_t mask = (uint32_t)br;  // 0 or 0xffffffff
  Fp<P> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)t[i] + (P::p(i) & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class P>
ZK_HD Fp<P> neg(const Fp<P>& a) {
  if (a.is_zero()) return a;
  Fp<P> r;
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int64_t d = (int64_t)P::p(i) - (int64_t)a.v[i] + br;
    r.v[i] = (uint32_t)d;
    br = d >> 32;
  }
  return r;
}

template <class P>
ZK_HD Fp<P> dbl(const Fp<P>& a) {
  return add(a, a);
}

// Montgomery product a*b*R^-1 mod p.  CIOS over 32-bit limbs; because the top two bits of
// both BN254 moduli are clear the running value never exceeds 2p and the 10th word of the
// textbook CIOS is not needed (the "no-carry" variant gnark-crypto's generator documents).
template <class P>
ZK_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t bi = b.v[i];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c = (uint64_t)a.v[j] * bi + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t t8 = (uint32_t)c;
    const uint32_t m = t[0] * P::inv;
    c = (uint64_t)m * P::p(0) + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c = (uint64_t)m * P::p(j) + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    t[7] = t8 + (uint32_t)c;
  }
  Fp<P> r;
  reduce_once<P>(r.v, t, 0);
  return r;
}

template <class P>
ZK_HD Fp<P> sqr(const Fp<P>& a) {
  return mul(a, a);
}

template <class P>
ZK_HD Fp<P> to_mont(const Fp<P>& a) {
  Fp<P> r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
  return mul(a, r2);
}

template <class P>
ZK_HD Fp<P> from_mont(const Fp<P>& a) {
  Fp<P> o = Fp<P>::zero();
  o.v[0] = 1;
  return mul(a, o);
}

// Field inverse, inverse(0) = 0 (gnark-crypto's Element.Inverse convention): the fixed-count
// divsteps of modinv30.h on the canonical integer a = x 2^256, a^-1 = x^-1 2^-256, brought back into
// the Montgomery image by one product with 2^768 (. 2^-256).  ~13 000 instructions instead of the
// ~210 000 of a^(p-2) on this representation; same canonical result.
template <class P> struct ModInvOf;
template <> struct ModInvOf<FrParams> {
  typedef ModInvFr type;
  static ZK_HD uint32_t k768(int i) {   // 2^768 mod r
    constexpr uint32_t v[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu,
                               0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};
    return v[i];
  }
};
template <> struct ModInvOf<FqParams> {
  typedef ModInvFq type;
  static ZK_HD uint32_t k768(int i) {   // 2^768 mod p
    constexpr uint32_t v[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u,
                               0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u};
    return v[i];
  }
};
template <class P>
ZK_HD Fp<P> inverse(const Fp<P>& a) {
  Fp<P> t, k;
  modinv30<typename ModInvOf<P>::type>(t.v, a.v);
#pragma unroll
  for (int i = 0; i < 8; i++) k.v[i] = ModInvOf<P>::k768(i);
  return mul(t, k);
}

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// ---- Fq2 = Fq[u]/(u^2+1) ------------------------------------------------------------------
// Memory image of gnark-crypto's E2{A0, A1}.
struct Fq2 {
  Fq c0, c1;
  static ZK_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  static ZK_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
  ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  ZK_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
};

ZK_HD Fq2 add(const Fq2& a, const Fq2& b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
ZK_HD Fq2 sub(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
ZK_HD Fq2 neg(const Fq2& a) { return Fq2{neg(a.c0), neg(a.c1)}; }
ZK_HD Fq2 dbl(const Fq2& a) { return Fq2{dbl(a.c0), dbl(a.c1)}; }
ZK_HD Fq2 mul(const Fq2& a, const Fq2& b) {
  // Karatsuba, u^2 = -1
  Fq v0 = mul(a.c0, b.c0);
  Fq v1 = mul(a.c1, b.c1);
  Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  return Fq2{sub(v0, v1), sub(sub(s, v0), v1)};
}
ZK_HD Fq2 sqr(const Fq2& a) {
  // (a0+a1)(a0-a1) + 2 a0 a1 u
  Fq t = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
  Fq m = mul(a.c0, a.c1);
  return Fq2{t, dbl(m)};
}
ZK_HD Fq2 inverse(const Fq2& a) {
  // 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + a1^2)
  Fq n = add(sqr(a.c0), sqr(a.c1));
  Fq ni = inverse(n);
  return Fq2{mul(a.c0, ni), neg(mul(a.c1, ni))};
}

}  // namespace zk
