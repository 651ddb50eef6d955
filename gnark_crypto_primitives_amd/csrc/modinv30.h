// Modular inversion by the Bernstein-Yang "safegcd" divsteps (eprint 2019/266), in the fixed-count
// form that runs the same instructions for every input: 20 batches of 30 divsteps on nine signed
// 30-bit limbs, each batch a 2 x 2 integer matrix (scaled by 2^30) that is then applied to (f, g)
// and, modulo p, to (d, e).  600 divsteps cover every 256-bit modulus (the bound for this
// "zeta" variant is 590).
//
// Why: the witness solver's field inversions are the hints of every twisted-Edwards addition
// (elgamal/*.go, eddsa/verifier.go: two divisions per point addition, hundreds of dependent ones per
// scalar multiplication).  One lane = one inversion on a wavefront that is alone on its SIMD, so
// the cost is the instruction count of the dependency chain: Fermat's a^(p-2) is 254 squarings +
// ~130 products, ~75 000 instructions on the 29-bit form; this is ~13 000, most of them full-rate
// 32-bit ALU operations.  No branches on data: the 64 lanes never diverge.
//
// x = 0 -> 0 (gnark-crypto's Element.Inverse convention, fr/element.go [UPSTREAM-RECALL]).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

struct ModInvFr {   // BN254 scalar field r
  static ZK_HD int32_t m(int i) {
    constexpr int32_t v[9] = {0x30000001, 0x0f87d64f, 0x1b970914, 0x0cfa121e, 0x01585d28,
                              0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
    return v[i];
  }
  static constexpr uint32_t inv30 = 0x10000001u;   // r^-1 mod 2^30
};
struct ModInvFq {   // BN254 base field p
  static ZK_HD int32_t m(int i) {
    constexpr int32_t v[9] = {0x187cfd47, 0x3082305b, 0x071ca8d3, 0x205aa45a, 0x01585d97,
                              0x0116da06, 0x1a029b85, 0x139cb84c, 0x00003064};
    return v[i];
  }
  static constexpr uint32_t inv30 = 0x1b799c77u;   // p^-1 mod 2^30
};

// x: canonical integer below the modulus, 8 x u32 little-endian; out = x^-1 mod m (0 for x = 0)
template <class M>
ZK_HD void modinv30(uint32_t out[8], const uint32_t x[8]) {
  constexpr int32_t M30 = (1 << 30) - 1;
  int32_t f[9], g[9], d[9], e[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    f[i] = M::m(i);
    d[i] = 0;
    e[i] = 0;
    // bits [30 i, 30 i + 30) of x
    const int lo = 30 * i, w = lo >> 5, s = lo & 31;
    uint32_t t = w < 8 ? x[w] >> s : 0u;
    if (s > 2 && w + 1 < 8) t |= x[w + 1] << (32 - s);
    g[i] = (int32_t)(t & (uint32_t)M30);
  }
  e[0] = 1;
  int32_t zeta = -1;   // -(delta + 1/2), delta = 1/2 initially
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    // ---- 30 divsteps on the low limbs: the transition matrix (u v; q r), scaled by 2^30
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t ff = (uint32_t)f[0] | ((uint32_t)f[1] << 30), gg = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
#pragma unroll
    for (int i = 0; i < 30; i++) {
      uint32_t c1 = (uint32_t)(zeta >> 31);     // zeta < 0
      const uint32_t c2 = 0u - (gg & 1u);        // g odd
      const uint32_t xx = (ff ^ c1) - c1, yy = (u ^ c1) - c1, zz = (v ^ c1) - c1;
      gg += xx & c2;
      q += yy & c2;
      r += zz & c2;
      c1 &= c2;
      zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
      ff += gg & c1;
      u += q & c1;
      v += r & c1;
      gg >>= 1;
      u <<= 1;
      v <<= 1;
    }
    const int64_t U = (int32_t)u, V = (int32_t)v, Q = (int32_t)q, R = (int32_t)r;
    // ---- (d, e) <- (u v; q r) (d, e) / 2^30 mod m, limbs kept in (-2m, m)
    {
      const int32_t sd = d[8] >> 31, se = e[8] >> 31;
      int32_t md = ((int32_t)u & sd) + ((int32_t)v & se);
      int32_t me = ((int32_t)q & sd) + ((int32_t)r & se);
      int64_t cd = U * d[0] + V * e[0];
      int64_t ce = Q * d[0] + R * e[0];
      // multiples of m that clear the bottom 30 bits
      md -= (int32_t)((M::inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
      me -= (int32_t)((M::inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
      cd += (int64_t)M::m(0) * md;
      ce += (int64_t)M::m(0) * me;
      cd >>= 30;
      ce >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        const int32_t di = d[i], ei = e[i];
        cd += U * di + V * ei;
        ce += Q * di + R * ei;
        cd += (int64_t)M::m(i) * md;
        ce += (int64_t)M::m(i) * me;
        d[i - 1] = (int32_t)cd & M30;
        cd >>= 30;
        e[i - 1] = (int32_t)ce & M30;
        ce >>= 30;
      }
      d[8] = (int32_t)cd;
      e[8] = (int32_t)ce;
    }
    // ---- (f, g) <- (u v; q r) (f, g) / 2^30 (exact)
    {
      int64_t cf = U * f[0] + V * g[0];
      int64_t cg = Q * f[0] + R * g[0];
      cf >>= 30;
      cg >>= 30;
#pragma unroll
      for (int i = 1; i < 9; i++) {
        const int32_t fi = f[i], gi = g[i];
        cf += U * fi + V * gi;
        cg += Q * fi + R * gi;
        f[i - 1] = (int32_t)cf & M30;
        cf >>= 30;
        g[i - 1] = (int32_t)cg & M30;
        cg >>= 30;
      }
      f[8] = (int32_t)cf;
      g[8] = (int32_t)cg;
    }
  }
  // g = 0 and f = +-1 now (f = +-m for x = 0, where d = 0): the inverse is sign(f) * d, in (-2m, m)
  {
    int32_t add = d[8] >> 31;
    const int32_t ng = f[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      d[i] += M::m(i) & add;
      d[i] = (d[i] ^ ng) - ng;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d[i + 1] += d[i] >> 30;
      d[i] &= M30;
    }
    add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] += M::m(i) & add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      d[i + 1] += d[i] >> 30;
      d[i] &= M30;
    }
  }
  // nine 30-bit limbs -> 8 x u32
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int lo = 32 * w, i = lo / 30, s = lo - 30 * i;   // word w starts at bit s of limb i
    uint32_t t = (uint32_t)d[i] >> s;
    t |= (uint32_t)d[i + 1] << (30 - s);
    if (s > 28 && i + 2 < 9) t |= (uint32_t)d[i + 2] << (60 - s);
    out[w] = t;
  }
}

}  // namespace zk
