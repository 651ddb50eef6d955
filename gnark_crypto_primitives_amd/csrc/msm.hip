// Batched multi-scalar multiplication over fixed bases: out[p] = sum_i s[i][p] * P_i, p < batch.
//
// Replaces G1Jac.MultiExp / G2Jac.MultiExp (gnark-crypto ecc/bn254/multiexp*.go) as groth16.Prove
// calls them - four G1 and one G2 MSM per proof over the proving key's bases
// [UPSTREAM-RECALL, SURVEY.md §3.2 step 5].  The result is the affine group element, which is
// canonical, so it equals gnark's bucket-method result bit for bit.
//
// MI355X-first design.  gnark runs Pippenger per proof because a CPU sees one
// proof at a time.  Here a batch of proofs shares the bases, and the bases are fixed for the
// lifetime of the circuit, so the work is organised the other way round: multiples of the bases are
// precomputed in HBM, one lane = one proof, and every step is ONE mixed addition of a gathered
// table entry into a register-resident XYZZ accumulator -- no buckets, no bucket reduction, no
// sorting, no atomics.  Three table layouts, chosen per key by the additions they need within
// the HBM budget:
//   * comb tables (default for keys of >= 4096 bases): a joint table of all subset sums of every
//     group of k consecutive bases, 254 one-bit windows: 254 / k additions per (base, proof)
//     (k = 18 / 19 for the Arbo-160 key).  Window index = a grid dimension; the 254 window sums of
//     a proof are combined by Horner's rule.
//   * one shared table per base (d * P, d = 1..2^(c-1)), ceil(255 / c) signed-digit windows with
//     their own accumulators, Horner combine (explicit window_bits 100 + c).
//   * per-window tables (d * 2^(shift_j) * P for every window j), all windows into one
//     accumulator: no Horner tail, the right shape for small keys and for the one-base delta
//     multiples of the assembly.
// The bound is the vector integer ALU (v_mad_u64_u32), not HBM; both fractions are reported by
// bench.py.
//
// Algorithmic bytes per launch: n * sizeof(affine) + batch * n * 32.
#include <cstdlib>

#include "zkmi_internal.h"
#include "ec29.h"

// This file is compiled twice (Makefile): ZK_MSM_PART 1 = plans, dispatch and the G1 (Fq) kernels,
// ZK_MSM_PART 2 = the G2 (Fq2) instantiations of the table-build and accumulate paths only -- the
// two halves compile in parallel (each is several minutes of hipcc).
#ifndef ZK_MSM_PART
#define ZK_MSM_PART 1
#endif

namespace zk {

ZK_HD Fq to_r261_domain(const Fq& x) {
  Fq k;
#pragma unroll
  for (int i = 0; i < 8; i++) k.v[i] = Fq29Params::k261(i);
  return mul(x, k);
}
ZK_HD Fq2 to_r261_domain(const Fq2& x) { return Fq2{to_r261_domain(x.c0), to_r261_domain(x.c1)}; }

// ---- table construction ------------------------------------------------------------------------
// One thread per table row: (base) for a shared table, (base, window) for per-window tables, whose
// first entry Q = 2^(shift_j) P is reached by shift_j doublings.  A row is a running sum
// d * Q, d = 1..D, in XYZZ, made affine in segments of at most 512 entries with one Montgomery
// batch inversion per segment (the scratch is interleaved across threads so that lanes touch
// adjacent addresses).  Entries are stored as x*2^261, y*2^261 (canonical): the accumulate kernels
// work in the 2^261 domain of ff29.h and only unpack limbs.
template <class F>
__global__ __launch_bounds__(64) void msm_build_table(const Affine<F>* __restrict__ bases,
                                                      uint64_t r0, uint64_t n_rows, WinPlan plan,
                                                      Affine<F>* __restrict__ table,
                                                      F* __restrict__ scratch, uint32_t T,
                                                      uint32_t seg_len, int to_r261) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = r0 + t;
  if (t >= T || r >= n_rows) return;
  F* szz = scratch + t;
  F* szzz = scratch + (size_t)seg_len * T + t;
  F* spre = scratch + (size_t)2 * seg_len * T + t;
  uint32_t D;
  Affine<F>* row;
  Affine<F> Q;
  if (plan.shared) {
    D = plan.per_base;
    row = table + (size_t)r * plan.per_base;
    Q = bases[r];
  } else {
    const uint32_t i = (uint32_t)(r / (uint32_t)plan.W), j = (uint32_t)(r % (uint32_t)plan.W);
    D = 1u << (plan.bits[j] - 1);
    row = table + (size_t)i * plan.per_base + plan.off[j];
    Q = bases[i];
    if (!Q.is_inf() && j) {
      uint32_t shift = 0;
      for (uint32_t k = 0; k < j; k++) shift += plan.bits[k];
      XYZZ<F> a = XYZZ<F>::from_affine(Q);
      for (uint32_t k = 0; k < shift; k++) a = dbl(a);
      Q = to_affine(a);
    }
  }
  if (Q.is_inf()) {
    for (uint32_t d = 0; d < D; d++) row[d] = Affine<F>::inf();
    return;
  }
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t s0 = 0; s0 < D; s0 += seg_len) {
    Affine<F>* seg = row + s0;
    const uint32_t len = D - s0 < seg_len ? D - s0 : seg_len;
    F pref = F::one();
    for (uint32_t d = 0; d < len; d++) {
      madd(acc, Q);
      seg[d].x = acc.x;
      seg[d].y = acc.y;
      szz[(size_t)d * T] = acc.zz;
      szzz[(size_t)d * T] = acc.zzz;
      spre[(size_t)d * T] = pref;
      pref = mul(pref, acc.zzz);
    }
    F inv = inverse(pref);
    for (uint32_t d = len; d-- > 0;) {
      F zzz = szzz[(size_t)d * T];
      F izzz = mul(inv, spre[(size_t)d * T]);
      inv = mul(inv, zzz);
      F izz = sqr(mul(izzz, szz[(size_t)d * T]));
      F x = mul(seg[d].x, izz), y = mul(seg[d].y, izzz);
      seg[d].x = to_r261 ? to_r261_domain(x) : x;
      seg[d].y = to_r261 ? to_r261_domain(y) : y;
    }
  }
}

// ---- accumulation --------------------------------------------------------------------------------
// grid: x over proofs (Bp / blockDim.x), y over chunks of bases.  c, W wave-uniform.
// The accumulator lives in the 9 x 29-bit lazy representation (ff29.h / ec29.h): every field
// product is a carry-free v_mad_i64_i32 chain; G2 shares one Montgomery reduction per Fq2 component.
template <class F> struct Acc29;
template <> struct Acc29<Fq> {
  typedef G1Acc29 type;
  static __device__ __forceinline__ void add(G1Acc29& acc, const G1Affine& e, bool negd) {
    madd29(acc, unpack29<Fq29Params>(e.x.v), cneg(unpack29<Fq29Params>(e.y.v), negd));
  }
};
template <> struct Acc29<Fq2> {
  typedef G2Acc29 type;
  static __device__ __forceinline__ void add(G2Acc29& acc, const G2Affine& e, bool negd) {
    madd29(acc, unpack2_29(e.x), cneg(unpack2_29(e.y), negd));
  }
};

// Accumulator of the comb kernels.  G1: registers (36 + 18 for the entry: 128 VGPRs, no spills).
// G2 at two waves per SIMD has 256 VGPRs for a 72-limb accumulator, a 36-limb entry and ~90 limbs of
// temporaries: the compiler spilled ~100 registers to scratch (27.9 GB of scratch writes per launch in
// the round-2 PMC pass).  zz and zzz are only touched at the two ends of a mixed addition, so they
// live in LDS instead (36 words x 256 lanes = 36 KB per workgroup, [limb][lane]: conflict-free) and
// are re-read where the addition needs them the second time.
template <class F> struct CombAcc;
template <> struct CombAcc<Fq> {
  static constexpr int LDS_ROWS = 1;   // unused
  typedef G1Acc29 type;
  static __device__ __forceinline__ type init(int32_t (*)[256], uint32_t) { return G1Acc29::infinity(); }
  static __device__ __forceinline__ void add(type& acc, const G1Affine& e, bool negd,
                                             int32_t (*)[256], uint32_t) {
    madd29(acc, unpack29<Fq29Params>(e.x.v), cneg(unpack29<Fq29Params>(e.y.v), negd));
  }
  static __device__ __forceinline__ G1XYZZ result(const type& acc, int32_t (*)[256], uint32_t) {
    return to_std(acc);
  }
};
struct G2AccL {
  Fq2_29 x, y;
  bool inf;
};
static __device__ __forceinline__ Fq2_29 lds_ld2(int32_t (*z)[256], int row0, uint32_t t) {
  Fq2_29 r;
#pragma unroll
  for (int l = 0; l < 9; l++) {
    r.c0.v[l] = z[row0 + l][t];
    r.c1.v[l] = z[row0 + 9 + l][t];
  }
  return r;
}
static __device__ __forceinline__ void lds_st2(int32_t (*z)[256], int row0, uint32_t t,
                                               const Fq2_29& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) {
    z[row0 + l][t] = a.c0.v[l];
    z[row0 + 9 + l][t] = a.c1.v[l];
  }
}
template <> struct CombAcc<Fq2> {
  static constexpr int LDS_ROWS = 36;  // zz: rows 0..17, zzz: rows 18..35
  typedef G2AccL type;
  static __device__ __forceinline__ type init(int32_t (*)[256], uint32_t) {
    G2AccL a;
    a.x = a.y = Fq2_29::zero();
    a.inf = true;
    return a;
  }
  // ec29.h madd29(G2Acc29&, ...) with zz / zzz in LDS
  static __device__ __forceinline__ void add(type& acc, const G2Affine& e, bool negd,
                                             int32_t (*z)[256], uint32_t t) {
    const Fq2_29 qx = unpack2_29(e.x), qy = cneg(unpack2_29(e.y), negd);
    if (acc.inf) {
      acc.x = qx;
      acc.y = norm(qy);
      lds_st2(z, 0, t, Fq2_29::one());
      lds_st2(z, 18, t, Fq2_29::one());
      acc.inf = false;
      return;
    }
    const Fq2_29 u2 = mmul(qx, lds_ld2(z, 0, t));
    const Fq2_29 s2 = mmul(qy, lds_ld2(z, 18, t));
    const Fq2_29 p = sub(u2, acc.x);
    const Fq2_29 r = sub(s2, acc.y);
    const Fq2_29 pp = msqr(p);
    const Fq2_29 rr = msqr(r);
    if (is_zero_mulout(pp)) {
      if (is_zero_mulout(rr)) {
        G2Acc29 d;
        d.inf = false;
        mdbl29(d, qx, qy);
        acc.x = d.x;
        acc.y = d.y;
        lds_st2(z, 0, t, d.zz);
        lds_st2(z, 18, t, d.zzz);
      } else {
        acc.inf = true;
      }
      return;
    }
    const Fq2_29 ppp = mmul(p, pp);
    const Fq2_29 q = mmul(acc.x, pp);
    const Fq2_29 x3 = wred(sub(sub(rr, ppp), zk::add(q, q)));
    const Fq2_29 y3 = wred(sub(mmul(r, sub(q, x3)), mmul(acc.y, ppp)));
    uint32_t t2 = t;                     // opaque copy of the lane index: the second read of zz / zzz
    asm volatile("" : "+v"(t2));         // must be a new LDS read, not the first one kept in registers
    lds_st2(z, 0, t2, mmul(lds_ld2(z, 0, t2), pp));
    lds_st2(z, 18, t2, mmul(lds_ld2(z, 18, t2), ppp));
    acc.x = x3;
    acc.y = y3;
  }
  static __device__ __forceinline__ G2XYZZ result(const type& acc, int32_t (*z)[256], uint32_t t) {
    if (acc.inf) return G2XYZZ::inf();
    return G2XYZZ{to_std(acc.x), to_std(acc.y), to_std(lds_ld2(z, 0, t)), to_std(lds_ld2(z, 18, t))};
  }
};

// ONE_BASE only changes the symbol name: the one-base launches (delta multiples in the assembly,
// zkmi_fixed_base_mul) then show up separately from the proving-key MSMs in rocprofv3 statistics.
template <class F, bool ONE_BASE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void msm_accumulate(const Affine<F>* __restrict__ table,
                                                      const Fr* __restrict__ scalars,
                                                      const uint32_t* __restrict__ row_idx,
                                                      size_t Bp, uint32_t n, uint32_t per_chunk,
                                                      WinPlan plan, XYZZ<F>* __restrict__ partial,
                                                      Fr kmul, const uint8_t* __restrict__ inf) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.y;
  const uint32_t i0 = chunk * per_chunk;
  uint32_t i1 = i0 + per_chunk;
  if (i1 > n) i1 = n;
  const int W = plan.W;
  typename Acc29<F>::type acc = Acc29<F>::type::infinity();
  for (uint32_t i = i0; i < i1; i++) {
    if (inf[i]) continue;   // wave-uniform: the point at infinity contributes nothing
    const uint32_t row = row_idx ? row_idx[i] : i;
    // Montgomery image -> integer: product by the plain constant 1 (gnark's x*2^256) or by
    // 2^-5 (the solver's x*2^261)
    Fr s = mul(bi_ld(scalars, row, b, Bp), kmul);
    if (s.is_zero()) continue;
    const Affine<F>* trow = table + (size_t)i * plan.per_base;
    uint32_t carry = 0;
    for (int j = 0; j < W; j++) {
      const int c = plan.bits[j];                 // wave-uniform
      const uint32_t mask = (1u << c) - 1u;
      const uint32_t half = 1u << (c - 1);
      uint32_t d = (s.v[0] & mask) + carry;
#pragma unroll
      for (int l = 0; l < 7; l++) s.v[l] = (s.v[l] >> c) | (s.v[l + 1] << (32 - c));
      s.v[7] >>= c;
      const bool negd = d > half;
      carry = negd ? 1u : 0u;
      const uint32_t mag = negd ? (mask + 1u - d) : d;
      if (mag) {
        const Affine<F> e = trow[plan.off[j] + (mag - 1)];
        Acc29<F>::add(acc, e, negd);
      }
    }
  }
  partial[(size_t)chunk * Bp + b] = to_std(acc);
}

// ---- shared-table path: one table per base, one accumulator per (window, chunk) -------------------
// The per-window tables above spend HBM on 2^(c_j) multiples of 2^(shift_j) P for EVERY window.
// With a single table of d * P the same HBM holds windows ~5 bits wider (c = 15 instead of ~10 for
// the Arbo-160 key: 17 mixed additions per (base, proof) instead of 25); the price is that the
// windows can no longer share an accumulator.  Lane = proof makes that free: the window index is a
// third grid dimension, every (window, chunk) block keeps its own register accumulator, and the W
// window sums of a proof are combined once per MSM with Horner's rule (255 doublings per proof,
// not per base).
//
// Signed digits without a carry chain: with K = sum_j 2^(pos_j + c_j - 1) and s' = s + K, digit j
// is ((s' >> pos_j) & (2^c_j - 1)) - 2^(c_j - 1), each in [-2^(c_j-1), 2^(c_j-1)).  One pass
// converts the Montgomery scalars to integers and stores the digits as int16, [window][base][proof].
static __global__ __launch_bounds__(256) void msm_digits_kernel(const Fr* __restrict__ scalars,
                                                         const uint32_t* __restrict__ row_idx,
                                                         size_t Bp, uint32_t n, WinPlan plan,
                                                         int32_t kmul32, Fr koff,
                                                         const uint8_t* __restrict__ inf,
                                                         int16_t* __restrict__ digits) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
    if (inf[i]) {   // wave-uniform: the point at infinity contributes nothing
      for (int j = 0; j < plan.W; j++) digits[((size_t)j * n + i) * Bp + b] = 0;
      continue;
    }
    const uint32_t row = row_idx ? row_idx[i] : i;
    // Montgomery image -> integer on the 29-bit form: x*2^256 * 32 / 2^261 (gnark's image) or
    // x*2^261 * 1 / 2^261 (the solver's)
    Fr29 k = Fr29::zero();
    k.v[0] = kmul32;
    const Fr raw = bi_ld(scalars, row, b, Bp);
    Fr s;
    pack_canonical<Fr29Params>(s.v, mul(unpack29<Fr29Params>(raw.v), k));
    uint64_t cy = 0;
#pragma unroll
    for (int l = 0; l < 8; l++) {
      cy += (uint64_t)s.v[l] + koff.v[l];
      s.v[l] = (uint32_t)cy;
      cy >>= 32;
    }
    uint32_t pos = 0;
    for (int j = 0; j < plan.W; j++) {
      const uint32_t c = plan.bits[j];              // wave-uniform
      const uint32_t w = pos >> 5, sh = pos & 31u;
      uint32_t v = s.v[w] >> sh;
      if (sh && w + 1 < 8) v |= s.v[w + 1] << (32u - sh);
      const int32_t d = (int32_t)(v & ((1u << c) - 1u)) - (int32_t)(1u << (c - 1));
      digits[((size_t)j * n + i) * Bp + b] = (int16_t)d;
      pos += c;
    }
  }
}

// grid: x over proofs, y over windows, z over chunks of bases (all windows of a chunk are dispatched
// together, so the blocks that walk the same tables run at the same time)
template <class F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void
msm_accumulate_shared(const Affine<F>* __restrict__ table, const int16_t* __restrict__ digits,
                      size_t Bp, uint32_t n, uint32_t per_chunk, uint32_t per_base,
                      XYZZ<F>* __restrict__ partial) {
  // (an XCD-aware deal of the chunks -- all blocks of a chunk on one L2 -- measured no better)
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.z, j = blockIdx.y;
  const uint32_t i0 = chunk * per_chunk;
  uint32_t i1 = i0 + per_chunk;
  if (i1 > n) i1 = n;
  const int16_t* dj = digits + (size_t)j * n * Bp + b;
  typename Acc29<F>::type acc = Acc29<F>::type::infinity();
  for (uint32_t i = i0; i < i1; i++) {
    const int32_t d = dj[(size_t)i * Bp];
    if (d) {
      const bool negd = d < 0;
      const uint32_t mag = (uint32_t)(negd ? -d : d);
      const Affine<F> e = table[(size_t)i * per_base + (mag - 1)];
      Acc29<F>::add(acc, e, negd);
    }
  }
  partial[((size_t)j * gridDim.z + chunk) * Bp + b] = to_std(acc);
}

// out[b] = sum_j 2^(pos_j) * wsum[j][b]; blockIdx.y selects one of up to four independent sums
template <class F>
struct HornerArgs {
  const XYZZ<F>* wsum[4];
  XYZZ<F>* out[4];
};
template <class F>
__global__ __launch_bounds__(64) void msm_horner(HornerArgs<F> args, size_t Bp, WinPlan plan) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  const XYZZ<F>* __restrict__ wsum = args.wsum[blockIdx.y];
  XYZZ<F>* __restrict__ out = args.out[blockIdx.y];
  XYZZ<F> acc = wsum[(size_t)(plan.W - 1) * Bp + b];
  for (int j = plan.W - 2; j >= 0; j--) {
    const int c = plan.bits[j];
    for (int k = 0; k < c; k++)
      if (!acc.is_inf()) acc = dbl(acc);
    const XYZZ<F> p = wsum[(size_t)j * Bp + b];
    padd(acc, p);
  }
  out[b] = acc;
}

// ---- comb tables: one mixed addition per (group of k bases, bit, proof) ------------------------------
// A table over single bases spends 2^(c-1) entries per base to consume c scalar bits per addition.
// A JOINT table over k bases with one-bit digits, T[g][m] = sum_{i in m} P_{gk+i} for every non-empty
// subset m, spends 2^k / k entries per base and consumes k scalar bits per addition: k = 18 fits
// the HBM that gave c = 15 (14.1 instead of 17 additions per base and proof), and the G2 table
// affords k = 19.  Digits are plain bits -- no signs, no recoding: the index of (group, bit j) is
// bit j of the group's k scalars, 0 = nothing to add.  The 254 window sums go through the same
// chunk reduction and a Horner pass of one doubling per window.
constexpr int COMB_W = 254;   // scalars are below 2^254

// D[g][t] = P_t - (P_0 + ... + P_{t-1}): entry(m + 1) = entry(m) + D[number of trailing ones of m]
// (signed tables: every step is twice that, so 2 D is stored); gsum[g] = sum of the group's bases
template <class F>
__global__ void comb_prep(const Affine<F>* __restrict__ bases, uint32_t n, uint32_t k,
                          uint32_t n_groups, Affine<F>* __restrict__ dpts,
                          Affine<F>* __restrict__ gsum, int twice) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  XYZZ<F> s = XYZZ<F>::inf();
  for (uint32_t t = 0; t < k; t++) {
    const size_t i = (size_t)g * k + t;
    const Affine<F> P = i < n ? bases[i] : Affine<F>::inf();
    XYZZ<F> d = s;
    d.y = neg(d.y);
    madd(d, P);
    if (twice) d = dbl(d);
    dpts[i] = to_affine(d);
    madd(s, P);
  }
  gsum[g] = to_affine(s);
}
// sum of the per-group sums (one thread: a few thousand mixed additions, once per key)
template <class F>
__global__ void comb_total(const Affine<F>* __restrict__ gsum, uint32_t n_groups,
                           Affine<F>* __restrict__ out) {
  XYZZ<F> s = XYZZ<F>::inf();
  for (uint32_t g = 0; g < n_groups; g++) madd(s, gsum[g]);
  *out = to_affine(s);
}

// one thread per (group, segment of seg_len consecutive table indices).
// Unsigned tables: entry m = sum of the bases whose bit is set in m (2^k entries, entry 0 unused).
// SIGNED tables: entry e (k - 1 bits) = P_(k-1) + sum_{i < k-1} (e_i ? +P_i : -P_i): every k-bit
// sign pattern or its complement has its top bit set, and sigma(~M) = -sigma(M), so 2^(k-1) entries
// serve all 2^k patterns -- one more base per group in the same HBM.
template <class F, bool SIGNED>
__global__ __launch_bounds__(64) void comb_build(const Affine<F>* __restrict__ bases,
                                                 const Affine<F>* __restrict__ dpts, uint32_t n,
                                                 uint32_t k, uint64_t r0, uint64_t n_rows,
                                                 Affine<F>* __restrict__ table,
                                                 F* __restrict__ scratch, uint32_t T,
                                                 uint32_t seg_len, int* __restrict__ any_inf) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = r0 + t;
  if (t >= T || r >= n_rows) return;
  const uint32_t idx_bits = SIGNED ? k - 1 : k;
  const uint32_t per_group = 1u << idx_bits, segs = per_group / seg_len;
  const uint32_t g = (uint32_t)(r / segs), m0 = (uint32_t)(r % segs) * seg_len;
  F* szz = scratch + t;
  F* szzz = scratch + (size_t)seg_len * T + t;
  F* spre = scratch + (size_t)2 * seg_len * T + t;
  Affine<F>* seg = table + (size_t)g * per_group + m0;
  // entry of the first index of the segment; `live` = bases that exist and are finite (the
  // unsigned digit pass never sets the bit of any other base, so masks outside `live` are never
  // gathered)
  XYZZ<F> acc = XYZZ<F>::inf();
  uint32_t live = 0;
  for (uint32_t i = 0; i < k; i++) {
    const size_t bi = (size_t)g * k + i;
    if (bi >= n) continue;
    const Affine<F> P = bases[bi];
    if (!P.is_inf()) live |= 1u << i;
    if (SIGNED) {
      if (i == k - 1 || ((m0 >> i) & 1u))
        madd(acc, P);
      else
        madd(acc, neg(P));
    } else if ((m0 >> i) & 1u) {
      madd(acc, P);
    }
  }
  F pref = F::one();
  bool inf_seen = false;
  for (uint32_t d = 0; d < seg_len; d++) {
    const bool is_inf = acc.is_inf();
    const uint32_t m = m0 + d;
    inf_seen = inf_seen || (is_inf && (SIGNED ? live != 0 : (m != 0 && (m & ~live) == 0)));
    seg[d].x = is_inf ? F::zero() : acc.x;
    seg[d].y = is_inf ? F::zero() : acc.y;
    szz[(size_t)d * T] = is_inf ? F::one() : acc.zz;
    szzz[(size_t)d * T] = is_inf ? F::one() : acc.zzz;
    spre[(size_t)d * T] = pref;
    pref = mul(pref, is_inf ? F::one() : acc.zzz);
    const uint32_t tz = (uint32_t)__builtin_ctz(m0 + d + 1);   // trailing ones of the index
    if (tz < idx_bits) madd(acc, dpts[(size_t)g * k + tz]);
  }
  F inv = inverse(pref);
  for (uint32_t d = seg_len; d-- > 0;) {
    const F zzz = szzz[(size_t)d * T];
    const F izzz = mul(inv, spre[(size_t)d * T]);
    inv = mul(inv, zzz);
    const F izz = sqr(mul(izzz, szz[(size_t)d * T]));
    // entries are stored in the 2^261 domain of the accumulate kernels (0 stays 0)
    seg[d].x = to_r261_domain(mul(seg[d].x, izz));
    seg[d].y = to_r261_domain(mul(seg[d].y, izzz));
  }
  if (inf_seen) atomicOr(any_inf, 1);
}

// Montgomery scalars -> plain integers, same planar layout (row i of the output = base i).
//
// SIGNED (sign-pattern tables): every scalar is rewritten as a sum of 254 signed powers of two.
//   t = s / 2 mod r;  e = 1 if t is even;  t' = t + e (odd, <= r);  C = (t' + 2^254 - 1) / 2 < 2^254
//   => t' = sum_j (2 C_j - 1) 2^j, and  s P = 2 (t' P - e P).
// Bit j of C is base i's sign in window j; the parity e rides in bit 255 and becomes the sign
// pattern of one extra window (index 254) whose sum is subtracted once at the end together with
// the sum of all bases:  sum_i s_i P_i = 2 H - W_254 - S,  H = sum_j 2^j W_j  (msm_horner_comb).
template <bool SIGNED>
__global__ __launch_bounds__(256) void comb_scalars_kernel(const Fr* __restrict__ scalars,
                                                           const uint32_t* __restrict__ row_idx,
                                                           size_t Bp, uint32_t n, int32_t kmul32,
                                                           const uint8_t* __restrict__ inf,
                                                           Fr* __restrict__ out) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t i = blockIdx.y; i < n; i += gridDim.y) {
    Fr s = Fr::zero();
    if (!inf[i]) {   // the point at infinity contributes nothing: its bits never matter
      Fr29 kk = Fr29::zero();
      kk.v[0] = kmul32;
      const Fr raw = bi_ld(scalars, row_idx ? row_idx[i] : i, b, Bp);
      pack_canonical<Fr29Params>(s.v, mul(unpack29<Fr29Params>(raw.v), kk));
      if (SIGNED) {
        // t = (s + (s odd ? r : 0)) >> 1      (s + r < 2^255)
        const uint32_t odd = s.v[0] & 1u;
        uint64_t cy = 0;
#pragma unroll
        for (int l = 0; l < 8; l++) {
          cy += (uint64_t)s.v[l] + (odd ? FrParams::p(l) : 0u);
          s.v[l] = (uint32_t)cy;
          cy >>= 32;
        }
#pragma unroll
        for (int l = 0; l < 7; l++) s.v[l] = (s.v[l] >> 1) | (s.v[l + 1] << 31);
        s.v[7] >>= 1;
        const uint32_t e = (s.v[0] & 1u) ^ 1u;
        // C = (t' - 1 + 2^254) >> 1 with t' = t + e = t | 1 (odd), so t' - 1 = t with bit 0 cleared;
        // t' <= r < 2^254: the sum stays below 2^255
        s.v[0] &= ~1u;
        cy = 0;
#pragma unroll
        for (int l = 0; l < 8; l++) {
          cy += (uint64_t)s.v[l] + (l == 7 ? 0x40000000u : 0u);   // + 2^254
          s.v[l] = (uint32_t)cy;
          cy >>= 32;
        }
#pragma unroll
        for (int l = 0; l < 7; l++) s.v[l] = (s.v[l] >> 1) | (s.v[l + 1] << 31);
        s.v[7] = (s.v[7] >> 1) | (e << 31);
      }
    }
    bi_st(out, i, b, Bp, s);
  }
}

// digits[j][g][b] = sum_i bit_j(s[gk+i][b]) << i.  SIGNED: k-bit sign pattern M -> (index, negate):
// top bit set: entry M & (2^(k-1) - 1); clear: entry ~M & (2^(k-1) - 1), negated (bit 31 of the
// digit).  Window 254 takes bit 255 of the rewritten scalars (the parity pattern).
template <int KMAX, bool SIGNED>
__global__ __launch_bounds__(256) void comb_digits_kernel(const Fr* __restrict__ sint, size_t Bp,
                                                          uint32_t n, uint32_t k, uint32_t n_groups,
                                                          uint32_t* __restrict__ digits) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* base = reinterpret_cast<const uint4*>(sint);
  const uint32_t low = (1u << (k - 1)) - 1u;
  for (uint32_t g = blockIdx.y; g < n_groups; g += gridDim.y) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint4 wv[KMAX];
#pragma unroll
      for (int i = 0; i < KMAX; i++) {
        const size_t bi = (size_t)g * k + i;
        wv[i] = (i < (int)k && bi < n) ? base[(bi * 2 + h) * Bp + b] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        // 32 x 32 bit transpose (rows = bases of the group, columns = 32 scalar bits): afterwards
        // m[bit] holds, in bit i, bit `bit` of base i's word -- the window's index.  Five
        // butterfly stages of 16 masked swaps instead of 32 x k single-bit extractions.
        uint32_t m[32];
#pragma unroll
        for (int i = 0; i < 32; i++)
          m[i] = i < KMAX ? (c == 0 ? wv[i].x : c == 1 ? wv[i].y : c == 2 ? wv[i].z : wv[i].w) : 0u;
#pragma unroll
        for (int jj = 16; jj != 0; jj >>= 1) {
          const uint32_t mask = jj == 16 ? 0x0000ffffu : jj == 8 ? 0x00ff00ffu : jj == 4 ? 0x0f0f0f0fu
                                : jj == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
          for (int kk = 0; kk < 32; kk = (kk + jj + 1) & ~jj) {
            const uint32_t t = ((m[kk] >> jj) ^ m[kk + jj]) & mask;
            m[kk] ^= t << jj;
            m[kk + jj] ^= t;
          }
        }
#pragma unroll
        for (int bit = 0; bit < 32; bit++) {
          int j = (h * 4 + c) * 32 + bit;
          if (SIGNED) {
            if (j == 254) continue;     // C < 2^254
            if (j == 255) j = 254;      // parity pattern
          } else if (j >= COMB_W) {
            break;
          }
          uint32_t idx = m[bit];
          if (SIGNED) idx = ((idx >> (k - 1)) & 1u) ? (idx & low) : ((~idx & low) | 0x80000000u);
          digits[((size_t)j * n_groups + g) * Bp + b] = idx;
        }
      }
    }
  }
}

// grid: x over proofs, y over the windows (254, or 255 for signed tables), z over chunks of groups
template <class F, bool CHECK_INF, bool SIGNED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void
msm_accumulate_comb(const Affine<F>* __restrict__ table, const uint32_t* __restrict__ digits,
                    size_t Bp, uint32_t n_groups, uint32_t per_chunk, uint32_t per_group,
                    XYZZ<F>* __restrict__ partial) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.z, j = blockIdx.y;
  const uint32_t g0 = chunk * per_chunk;
  uint32_t g1 = g0 + per_chunk;
  if (g1 > n_groups) g1 = n_groups;
  const uint32_t* dj = digits + (size_t)j * n_groups * Bp + b;
  __shared__ int32_t zl[CombAcc<F>::LDS_ROWS][256];
  const uint32_t t = threadIdx.x;
  typename CombAcc<F>::type acc = CombAcc<F>::init(zl, t);
  for (uint32_t g = g0; g < g1; g++) {
    const uint32_t m = dj[(size_t)g * Bp];
    if (SIGNED) {   // a sign pattern is never "nothing to add"
      const Affine<F> e = table[(size_t)g * per_group + (m & 0x7fffffffu)];
      if (CHECK_INF && e.is_inf()) continue;
      CombAcc<F>::add(acc, e, (m >> 31) != 0, zl, t);
    } else if (m) {
      const Affine<F> e = table[(size_t)g * per_group + m];
      if (CHECK_INF && e.is_inf()) continue;
      CombAcc<F>::add(acc, e, false, zl, t);
    }
  }
  partial[((size_t)j * gridDim.z + chunk) * Bp + b] = CombAcc<F>::result(acc, zl, t);
}

// out[b] = sum_j 2^j * wsum[j][b], j < W, in two levels so that the dependent chain is short:
// comb_fold8 replaces wsum[8q] by sum_{i<8} 2^i wsum[8q+i] (one lane per (proof, q), 7 doublings +
// 7 additions), msm_horner_comb then runs Horner over the folded sums (8 doublings + 1 addition per
// step): 256 doublings + 32 additions on the critical path instead of 254 + 254.
template <class F>
struct HornerArgsRW {
  XYZZ<F>* wsum[4];
  XYZZ<F>* out[4];
  Affine<F> stotal[4];   // signed tables: sum of all bases of the MSM (standard Montgomery image)
};
template <class F>
__global__ __launch_bounds__(64) void comb_fold8(HornerArgsRW<F> args, size_t Bp, int W) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  XYZZ<F>* __restrict__ wsum = args.wsum[blockIdx.y];
  const int j0 = 8 * (int)blockIdx.z;
  int top = j0 + 7;
  if (top > W - 1) top = W - 1;
  XYZZ<F> acc = wsum[(size_t)top * Bp + b];
  for (int j = top - 1; j >= j0; j--) {
    acc = dbl(acc);
    const XYZZ<F> p = wsum[(size_t)j * Bp + b];
    padd(acc, p);
  }
  wsum[(size_t)j0 * Bp + b] = acc;
}
// W = number of power-of-two windows (254).  signed_tail: out = 2 H - wsum[W] - stotal.
template <class F>
__global__ __launch_bounds__(64) void msm_horner_comb(HornerArgsRW<F> args, size_t Bp, int W,
                                                      int signed_tail) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  const XYZZ<F>* __restrict__ wsum = args.wsum[blockIdx.y];
  const int Q = (W + 7) / 8;
  XYZZ<F> acc = wsum[(size_t)(8 * (Q - 1)) * Bp + b];
  for (int q = Q - 2; q >= 0; q--) {
#pragma unroll 1
    for (int i = 0; i < 8; i++) acc = dbl(acc);
    const XYZZ<F> p = wsum[(size_t)(8 * q) * Bp + b];
    padd(acc, p);
  }
  if (signed_tail) {
    acc = dbl(acc);
    XYZZ<F> corr = wsum[(size_t)W * Bp + b];
    corr.y = neg(corr.y);
    padd(acc, corr);
    madd(acc, neg(args.stotal[blockIdx.y]));
  }
  args.out[blockIdx.y][b] = acc;
}
template <class F>
static void launch_comb_horner(hipStream_t stream, const HornerArgsRW<F>& ha, int count, size_t Bp,
                               const WinPlan& plan) {
  const int W = COMB_W;   // plan.W = 254, or 255 with the correction window of signed tables
  hipLaunchKernelGGL((comb_fold8<F>),
                     dim3((unsigned)(Bp / 64), (unsigned)count, (unsigned)((W + 7) / 8)), dim3(64), 0,
                     stream, ha, Bp, W);
  hipLaunchKernelGGL((msm_horner_comb<F>), dim3((unsigned)(Bp / 64), (unsigned)count), dim3(64), 0,
                     stream, ha, Bp, W, (int)plan.comb_signed);
}

// sums groups of `group` consecutive chunk partials: out[g][b] = sum_{k < group} in[g*group + k][b]
// (blockIdx.z selects an independent set: partial += z * in_zstride, out += z * out_zstride)
template <class F>
__global__ __launch_bounds__(64) void msm_reduce(const XYZZ<F>* __restrict__ partial, size_t Bp,
                                                 uint32_t chunks, uint32_t group,
                                                 XYZZ<F>* __restrict__ out, size_t in_zstride,
                                                 size_t out_zstride) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  partial += (size_t)blockIdx.z * in_zstride;
  out += (size_t)blockIdx.z * out_zstride;
  const uint32_t k0 = blockIdx.y * group;
  uint32_t k1 = k0 + group;
  if (k1 > chunks) k1 = chunks;
  XYZZ<F> acc = partial[(size_t)k0 * Bp + b];
  for (uint32_t k = k0 + 1; k < k1; k++) {
    XYZZ<F> p = partial[(size_t)k * Bp + b];
    padd(acc, p);
  }
  out[(size_t)blockIdx.y * Bp + b] = acc;
}

template <class F>
__global__ __launch_bounds__(64) void xyzz_to_affine_kernel(const XYZZ<F>* __restrict__ in,
                                                            Affine<F>* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = to_affine(in[i]);
}

#if ZK_MSM_PART == 1
// ---- host side -------------------------------------------------------------------------------------
// ---- window plans ----------------------------------------------------------------------------------
WinPlan plan_with_windows(int W) {
  WinPlan p;
  if (W < 16) W = 16;     // 16-bit windows at most
  if (W > 64) W = 64;
  p.W = W;
  const int c0 = 255 / W, rem = 255 - c0 * W;
  uint32_t off = 0;
  for (int j = 0; j < W; j++) {
    p.bits[j] = (uint8_t)(j < rem ? c0 + 1 : c0);
    p.off[j] = off;
    off += 1u << (p.bits[j] - 1);
  }
  p.per_base = off;
  return p;
}
WinPlan plan_uniform(int c) {   // ceil(255 / c) windows of c bits (covers >= 255 bits)
  WinPlan p;
  p.W = (255 + c - 1) / c;
  uint32_t off = 0;
  for (int j = 0; j < p.W; j++) {
    p.bits[j] = (uint8_t)c;
    p.off[j] = off;
    off += 1u << (c - 1);
  }
  p.per_base = off;
  return p;
}
WinPlan plan_shared(int c) {   // ceil(255 / c) windows over ONE table of 2^(c-1) multiples
  WinPlan p;
  if (c < 4) c = 4;     // at most 64 windows
  if (c > 16) c = 16;   // digits are stored as int16
  p.shared = 1;
  p.W = (255 + c - 1) / c;
  for (int j = 0; j < p.W; j++) {
    p.bits[j] = (uint8_t)(j + 1 < p.W ? c : 255 - (p.W - 1) * c);
    p.off[j] = 0;
  }
  p.per_base = 1u << (c - 1);
  return p;
}
WinPlan plan_comb(int k, bool signed_tables) {
  WinPlan p;
  if (k < 2) k = 2;
  if (k > 21) k = 21;
  if (!signed_tables && k > 20) k = 20;
  p.comb = (uint8_t)k;
  p.comb_signed = signed_tables ? 1 : 0;
  p.W = COMB_W + (signed_tables ? 1 : 0);   // + the parity-correction window
  p.per_base = 0;
  return p;
}
// Comb plan of a key: group sizes (k1, k2) and table kinds minimising the estimated accumulate
// time among the plans whose tables fit `usable_bytes`.  Sign-pattern tables hold 2^(k-1) entries
// per group and need 255 windows, subset-sum tables 2^k entries and 254 windows.  Relative cost of
// one mixed addition, measured on MI355X (Arbo-160 key, ns per (group, window) at 1024 proofs):
// G1 subset-sum 61.1, G1 sign-pattern 62.8 (the per-lane negation), G2 subset-sum 171.6, G2
// sign-pattern 178.5 (198.8 while the G2 accumulator's zz / zzz lived in scratch: the extra live sign
// spilled; re-measured with them in LDS at the end of round 3: 60.5 ms against 58.0 ms per MSM at
// k = 19, and 57.1 ms with the k = 20 the same HBM affords).
void plan_comb_for_budget(size_t n1, size_t n2, double usable_bytes, int* k1, int* k2, bool* sg1,
                          bool* sg2, bool allow_signed) {
  // allow_signed = false: witnesses of bits / small integers (zkmi_pk_desc.sparse_witness): only
  // subset-sum tables skip their zero digits (measured on the Keccak address circuit: 172 ms with
  // them, 280 ms with sign patterns, for 5 % more additions on paper)
  double best = 1e300;
  *k1 = *k2 = 8;
  *sg1 = *sg2 = allow_signed;
  for (int s1 = 0; s1 < (allow_signed ? 2 : 1); s1++)
    for (int s2 = 0; s2 < (allow_signed ? 2 : 1); s2++)
      for (int a = 8; a <= (s1 ? 21 : 20); a++)
        for (int b = 8; b <= (s2 ? 21 : 20); b++) {
          const double g1 = (double)((n1 + a - 1) / a), g2 = (double)((n2 + b - 1) / b);
          const double bytes = g1 * (double)(1u << (a - s1)) * 64.0 + g2 * (double)(1u << (b - s2)) * 128.0;
          if (bytes > usable_bytes) continue;
          const double cost = g1 * (s1 ? 255 * 62.8 : 254 * 61.1) + g2 * (s2 ? 255 * 178.5 : 254 * 171.6);
          if (cost < best) {
            best = cost;
            *k1 = a;
            *k2 = b;
            *sg1 = s1 != 0;
            *sg2 = s2 != 0;
          }
        }
}
// Widths of the shared tables of a key: the (c1, c2) that minimises n1 * W(c1) + 3 * n2 * W(c2)
// (a G2 mixed addition costs about three G1 ones) among those whose tables fit `usable_bytes`.
void plan_shared_for_budget(size_t n1, size_t n2, double usable_bytes, int* c1, int* c2) {
  double best = 1e300;
  *c1 = *c2 = 4;
  for (int a = 4; a <= 16; a++)
    for (int b = 4; b <= 16; b++) {
      const double bytes = (double)n1 * (1u << (a - 1)) * 64.0 + (double)n2 * (1u << (b - 1)) * 128.0;
      if (bytes > usable_bytes) continue;
      const double cost = (double)n1 * ((255 + a - 1) / a) + 3.0 * (double)n2 * ((255 + b - 1) / b);
      if (cost < best) {
        best = cost;
        *c1 = a;
        *c2 = b;
      }
    }
}
WinPlan plan_windows_for_budget(size_t n_total, int group, double budget_bytes) {
  const double entry = group == 1 ? 64.0 : 128.0;
  for (int W = 16; W <= 64; W++) {
    WinPlan p = plan_with_windows(W);
    if ((double)n_total * p.per_base * entry <= budget_bytes) return p;
  }
  return plan_with_windows(64);
}

#endif  // ZK_MSM_PART == 1

// inf[i] = 1 when base i is the point at infinity: its table rows are zeros and must never reach a
// mixed addition (the digit pass / the accumulate loop skip the base)
template <class F>
__global__ void msm_inf_flags(const Affine<F>* __restrict__ bases, uint32_t n,
                              uint8_t* __restrict__ inf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inf[i] = bases[i].is_inf() ? 1 : 0;
}

template <class F>
int build_comb(zkmi_ctx* ctx, const Affine<F>* bases_dev, size_t n, const WinPlan& plan,
                      Affine<F>* table, int* any_inf_host, Affine<F>* stotal_host) {
  const uint32_t k = plan.comb;
  const bool sg = plan.comb_signed != 0;
  const size_t n_groups = (n + k - 1) / k;
  const uint32_t per_group = 1u << (sg ? k - 1 : k);
  const uint32_t seg_len = per_group < 512u ? per_group : 512u;
  const uint64_t n_rows = (uint64_t)n_groups * (per_group / seg_len);
  const size_t per_thread = (size_t)3 * seg_len * sizeof(F);
  size_t T = (size_t)2e9 / per_thread;
  if (T > n_rows) T = (size_t)n_rows;
  T = round_up(T, 64);
  if (T > 262144) T = 262144;
  void* scratch;
  const size_t dp_bytes = round_up(n_groups * k * sizeof(Affine<F>), 256);
  const size_t gs_bytes = round_up((n_groups + 1) * sizeof(Affine<F>), 256);
  int rc = ensure_scratch(ctx, 7, dp_bytes + gs_bytes + 256 + per_thread * T, &scratch);
  if (rc) return rc;
  Affine<F>* dpts = (Affine<F>*)scratch;
  Affine<F>* gsum = (Affine<F>*)((char*)scratch + dp_bytes);
  int* any_inf = (int*)((char*)scratch + dp_bytes + gs_bytes);
  F* inv_scratch = (F*)((char*)scratch + dp_bytes + gs_bytes + 256);
  ZK_HIP(hipMemsetAsync(any_inf, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL((comb_prep<F>), dim3((unsigned)((n_groups + 63) / 64)), dim3(64), 0,
                     ctx->stream, bases_dev, (uint32_t)n, k, (uint32_t)n_groups, dpts, gsum,
                     sg ? 1 : 0);
  hipLaunchKernelGGL((comb_total<F>), dim3(1), dim3(1), 0, ctx->stream, (const Affine<F>*)gsum,
                     (uint32_t)n_groups, gsum + n_groups);
  for (uint64_t r0 = 0; r0 < n_rows; r0 += T) {
    if (sg)
      hipLaunchKernelGGL((comb_build<F, true>), dim3((unsigned)(T / 64)), dim3(64), 0, ctx->stream,
                         bases_dev, (const Affine<F>*)dpts, (uint32_t)n, k, r0, n_rows, table,
                         inv_scratch, (uint32_t)T, seg_len, any_inf);
    else
      hipLaunchKernelGGL((comb_build<F, false>), dim3((unsigned)(T / 64)), dim3(64), 0, ctx->stream,
                         bases_dev, (const Affine<F>*)dpts, (uint32_t)n, k, r0, n_rows, table,
                         inv_scratch, (uint32_t)T, seg_len, any_inf);
  }
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpyAsync(any_inf_host, any_inf, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ZK_HIP(hipMemcpyAsync(stotal_host, gsum + n_groups, sizeof(Affine<F>), hipMemcpyDeviceToHost,
                        ctx->stream));
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

template <class F>
int build_impl(zkmi_ctx* ctx, const Affine<F>* bases_dev, size_t n, const WinPlan& plan,
                      Affine<F>* table) {
  const uint32_t Dmax = plan.shared ? plan.per_base : 1u << (plan.bits[0] - 1);
  const uint32_t seg_len = Dmax < 512u ? Dmax : 512u;
  const uint64_t n_rows = plan.shared ? (uint64_t)n : (uint64_t)n * (uint64_t)plan.W;
  // slab of threads sized so the inversion scratch stays under ~2 GB
  const size_t per_thread = (size_t)3 * seg_len * sizeof(F);
  size_t T = (size_t)2e9 / per_thread;
  if (T > n_rows) T = (size_t)n_rows;
  T = round_up(T, 64);
  if (T > 262144) T = 262144;
  void* scratch;
  int rc = ensure_scratch(ctx, 7, per_thread * T, &scratch);
  if (rc) return rc;
  for (uint64_t r0 = 0; r0 < n_rows; r0 += T) {
    hipLaunchKernelGGL((msm_build_table<F>), dim3((unsigned)(T / 64)), dim3(64), 0, ctx->stream,
                       bases_dev, r0, n_rows, plan, table, (F*)scratch, (uint32_t)T, seg_len, 1);
  }
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

#if ZK_MSM_PART == 1
extern template int build_comb<Fq2>(zkmi_ctx*, const Affine<Fq2>*, size_t, const WinPlan&,
                                    Affine<Fq2>*, int*, Affine<Fq2>*);
extern template int build_impl<Fq2>(zkmi_ctx*, const Affine<Fq2>*, size_t, const WinPlan&,
                                    Affine<Fq2>*);
int msm_bases_build(zkmi_ctx* ctx, int group, const void* bases_dev, size_t n, const WinPlan& plan,
                    zkmi_msm_bases** out) {
  if (group != 1 && group != 2) {
    ctx->err = "group must be 1 (G1) or 2 (G2)";
    return ZKMI_ERR_ARG;
  }
  auto* b = new zkmi_msm_bases();
  b->group = group;
  b->n = n;
  b->plan = plan;
  const size_t entry = group == 1 ? sizeof(G1Affine) : sizeof(G2Affine);
  b->n_groups = plan.comb ? (n + plan.comb - 1) / plan.comb : 0;
  b->table_bytes = plan.comb ? b->n_groups * ((size_t)1 << (plan.comb - plan.comb_signed)) * entry
                             : n * (size_t)plan.per_base * entry;
  if (n == 0) {
    *out = b;
    return ZKMI_OK;
  }
  hipError_t e = hipMalloc(&b->table, b->table_bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "hipMalloc of MSM window table failed (" + std::to_string(b->table_bytes) +
               " bytes): " + hipGetErrorString(e);
    delete b;
    return ZKMI_ERR_OOM;
  }
  if (hipMalloc(&b->inf, n) != hipSuccess) {
    (void)hipGetLastError();
    hipFree(b->table);
    delete b;
    return ZKMI_ERR_OOM;
  }
  if (group == 1)
    hipLaunchKernelGGL((msm_inf_flags<Fq>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, (const G1Affine*)bases_dev, (uint32_t)n, b->inf);
  else
    hipLaunchKernelGGL((msm_inf_flags<Fq2>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, (const G2Affine*)bases_dev, (uint32_t)n, b->inf);
  int rc;
  if (plan.comb)
    rc = group == 1 ? build_comb<Fq>(ctx, (const G1Affine*)bases_dev, n, plan, (G1Affine*)b->table,
                                     &b->entries_may_be_inf, &b->stotal1)
                    : build_comb<Fq2>(ctx, (const G2Affine*)bases_dev, n, plan,
                                      (G2Affine*)b->table, &b->entries_may_be_inf, &b->stotal2);
  else
    rc = group == 1
             ? build_impl<Fq>(ctx, (const G1Affine*)bases_dev, n, plan, (G1Affine*)b->table)
             : build_impl<Fq2>(ctx, (const G2Affine*)bases_dev, n, plan, (G2Affine*)b->table);
  if (rc) {
    hipFree(b->table);
    hipFree(b->inf);
    delete b;
    return rc;
  }
  *out = b;
  return ZKMI_OK;
}

#endif  // ZK_MSM_PART == 1

template <class F> static Affine<F> stotal_of(const zkmi_msm_bases* b);
template <> Affine<Fq> stotal_of<Fq>(const zkmi_msm_bases* b) { return b->stotal1; }
template <> Affine<Fq2> stotal_of<Fq2>(const zkmi_msm_bases* b) { return b->stotal2; }

template <class F>
int run_impl(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const Fr* scalars,
                    const uint32_t* row_idx, size_t Bp, XYZZ<F>* out, bool scalars_f,
                    XYZZ<F>* wsum_out, hipStream_t finish_stream) {
  Fr kmul = Fr::zero();
  kmul.v[0] = 1;  // plain 1: from_mont
  if (scalars_f) {  // plain 2^-5 mod r
    Fr t = Fr::zero();
    t.v[0] = 32;
    kmul = from_mont(inverse(to_mont(t)));
  }
  const size_t n = bases->n;
  if (bases->plan.comb) {
    const uint32_t k = bases->plan.comb;
    const bool sg = bases->plan.comb_signed != 0;
    const size_t G = bases->n_groups;
    const int W = bases->plan.W;   // 254, + the correction window of signed tables
    // measured at 4 / 8 / 12 / 16 / 24: 277 / 275 / 273.5 / 272 / 273.6 ms per Arbo-160 batch
    const size_t comb_factor = bases->chunk_factor ? bases->chunk_factor : 16;
    size_t chunks = comb_factor * 262144 / Bp / (size_t)W;
    if (chunks < 1) chunks = 1;
    if (chunks > G) chunks = G;
    const uint32_t per_chunk = (uint32_t)((G + chunks - 1) / chunks);
    chunks = (G + per_chunk - 1) / per_chunk;
    uint32_t group = 1;
    while ((size_t)group * group < chunks) group++;
    const uint32_t ngroups = (uint32_t)((chunks + group - 1) / group);
    void *partial, *digits, *sint;
    // deferred tail: the reductions run on finish_stream while the next MSM's accumulate already
    // writes the other partial buffer
    const bool defer = wsum_out && finish_stream && finish_stream != ctx->stream;
    const int pb = defer ? (int)(ctx->part_next++ & 1u) : 0;
    int rc = ensure_scratch(ctx, pb ? 16 : 6,
                            ((chunks + ngroups + 1) * W) * Bp * sizeof(XYZZ<F>), &partial);
    if (rc) return rc;
    if (ctx->part_ev_valid[pb]) {   // the last reduction that read this buffer must be done
      ZK_HIP(hipStreamWaitEvent(ctx->stream, ctx->part_ev[pb], 0));
      ctx->part_ev_valid[pb] = false;
    }
    // (Measured and dropped: the digit pass one MSM ahead on a fourth stream through two digit
    // buffers -- correct, no gain: the pass is ALU work like the accumulate kernel it would hide
    // under, which slowed down by exactly the pass's 7 ms.)
    if ((rc = ensure_scratch(ctx, 12, (size_t)W * G * Bp * sizeof(uint32_t), &digits))) return rc;
    if ((rc = ensure_scratch(ctx, 17, n * Bp * sizeof(Fr), &sint))) return rc;
    hipStream_t dq = ctx->stream;
    XYZZ<F>* mid = (XYZZ<F>*)partial + chunks * W * Bp;
    XYZZ<F>* wsum = wsum_out ? wsum_out : mid + (size_t)ngroups * W * Bp;
    const unsigned bx = (Bp % 256 == 0) ? 256 : 64;
    const dim3 sgrid((unsigned)(Bp / bx), (unsigned)(n < 16384 ? n : 16384));
    const dim3 dgrid((unsigned)(Bp / bx), (unsigned)(G < 8192 ? G : 8192));
    const int32_t km = (int32_t)(scalars_f ? 1 : 32);
    if (sg) {
      hipLaunchKernelGGL((comb_scalars_kernel<true>), sgrid, dim3(bx), 0, dq, scalars, row_idx, Bp,
                         (uint32_t)n, km, (const uint8_t*)bases->inf, (Fr*)sint);
      hipLaunchKernelGGL((comb_digits_kernel<21, true>), dgrid, dim3(bx), 0, dq, (const Fr*)sint, Bp,
                         (uint32_t)n, k, (uint32_t)G, (uint32_t*)digits);
    } else {
      hipLaunchKernelGGL((comb_scalars_kernel<false>), sgrid, dim3(bx), 0, dq, scalars, row_idx, Bp,
                         (uint32_t)n, km, (const uint8_t*)bases->inf, (Fr*)sint);
      hipLaunchKernelGGL((comb_digits_kernel<20, false>), dgrid, dim3(bx), 0, dq, (const Fr*)sint,
                         Bp, (uint32_t)n, k, (uint32_t)G, (uint32_t*)digits);
    }
    zkmi_ctx::ProveSet* es = ctx->msm_ev_set >= 0 ? &ctx->sets[ctx->msm_ev_set] : nullptr;
    const int ev = (es && es->msm_ev_used < 8) ? es->msm_ev_used++ : -1;
    if (ev >= 0) {
      es->msm_ev_group[ev] = bases->group;
      hipEventRecord(es->msm_ev[ev][0], ctx->stream);
    }
    const dim3 grid((unsigned)(Bp / bx), (unsigned)W, (unsigned)chunks);
    const uint32_t per_group = 1u << (sg ? k - 1 : k);
#define ZK_LAUNCH_COMB(CI, SG)                                                                   \
    hipLaunchKernelGGL((msm_accumulate_comb<F, CI, SG>), grid, dim3(bx), 0, ctx->stream,        \
                       (const Affine<F>*)bases->table, (const uint32_t*)digits, Bp, (uint32_t)G, \
                       per_chunk, per_group, (XYZZ<F>*)partial)
    if (bases->entries_may_be_inf) {
      if (sg) ZK_LAUNCH_COMB(true, true); else ZK_LAUNCH_COMB(true, false);
    } else {
      if (sg) ZK_LAUNCH_COMB(false, true); else ZK_LAUNCH_COMB(false, false);
    }
#undef ZK_LAUNCH_COMB
    if (ev >= 0) hipEventRecord(es->msm_ev[ev][1], ctx->stream);
    hipStream_t rq = ctx->stream;
    if (defer) {
      ZK_HIP(hipEventRecord(ctx->acc_ev[pb], ctx->stream));
      ZK_HIP(hipStreamWaitEvent(finish_stream, ctx->acc_ev[pb], 0));
      rq = finish_stream;
    }
    if (ngroups > 1) {
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), ngroups, (unsigned)W), dim3(64),
                         0, rq, (const XYZZ<F>*)partial, Bp, (uint32_t)chunks, group, mid,
                         chunks * Bp, (size_t)ngroups * Bp);
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1, (unsigned)W), dim3(64), 0,
                         rq, (const XYZZ<F>*)mid, Bp, ngroups, ngroups, wsum,
                         (size_t)ngroups * Bp, Bp);
    } else {
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1, (unsigned)W), dim3(64), 0,
                         rq, (const XYZZ<F>*)partial, Bp, (uint32_t)chunks, (uint32_t)chunks, wsum,
                         chunks * Bp, Bp);
    }
    if (defer) {
      ZK_HIP(hipEventRecord(ctx->part_ev[pb], finish_stream));
      ctx->part_ev_valid[pb] = true;
    }
    if (!wsum_out) {
      HornerArgsRW<F> ha{};
      ha.wsum[0] = wsum;
      ha.out[0] = out;
      ha.stotal[0] = stotal_of<F>(bases);
      launch_comb_horner<F>(ctx->stream, ha, 1, Bp, bases->plan);
    }
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
  }
  if (bases->plan.shared) {
    const WinPlan& plan = bases->plan;
    const int W = plan.W;
    Fr koff = Fr::zero();   // K = sum_j 2^(pos_j + c_j - 1)
    {
      uint32_t pos = 0;
      for (int j = 0; j < W; j++) {
        const uint32_t bit = pos + plan.bits[j] - 1;
        koff.v[bit >> 5] |= 1u << (bit & 31);
        pos += plan.bits[j];
      }
    }
    // (window, chunk) blocks: 8 x the wave slots of the chip, as for the per-window tables
    const size_t shared_factor = bases->chunk_factor ? bases->chunk_factor : 8;
    size_t chunks = shared_factor * 262144 / Bp / (size_t)W;
    if (chunks < 1) chunks = 1;
    if (chunks > n) chunks = n;
    const uint32_t per_chunk = (uint32_t)((n + chunks - 1) / chunks);
    chunks = (n + per_chunk - 1) / per_chunk;
    uint32_t group = 1;
    while ((size_t)group * group < chunks) group++;
    const uint32_t ngroups = (uint32_t)((chunks + group - 1) / group);
    void *partial, *digits;
    int rc = ensure_scratch(ctx, 6, ((chunks + ngroups + 1) * W) * Bp * sizeof(XYZZ<F>), &partial);
    if (rc) return rc;
    if (ctx->part_ev_valid[0]) {   // a deferred comb tail may still be reading this buffer
      ZK_HIP(hipStreamWaitEvent(ctx->stream, ctx->part_ev[0], 0));
      ctx->part_ev_valid[0] = false;
    }
    if ((rc = ensure_scratch(ctx, 12, (size_t)W * n * Bp * sizeof(int16_t), &digits))) return rc;
    XYZZ<F>* mid = (XYZZ<F>*)partial + chunks * W * Bp;
    // window sums: into the caller's buffer when the Horner step is deferred (msm_horner_run)
    XYZZ<F>* wsum = wsum_out ? wsum_out : mid + (size_t)ngroups * W * Bp;
    const unsigned bx = (Bp % 256 == 0) ? 256 : 64;
    zkmi_ctx::ProveSet* es = ctx->msm_ev_set >= 0 ? &ctx->sets[ctx->msm_ev_set] : nullptr;
    const int ev = (es && es->msm_ev_used < 8) ? es->msm_ev_used++ : -1;
    hipLaunchKernelGGL(msm_digits_kernel,
                       dim3((unsigned)(Bp / bx), (unsigned)(n < 16384 ? n : 16384)), dim3(bx), 0,
                       ctx->stream, scalars, row_idx, Bp, (uint32_t)n, plan,
                       (int32_t)(scalars_f ? 1 : 32), koff, (const uint8_t*)bases->inf,
                       (int16_t*)digits);
    if (ev >= 0) {   // the event pair brackets the accumulate launch alone (zkmi_last_timings [6], [7])
      es->msm_ev_group[ev] = bases->group;
      hipEventRecord(es->msm_ev[ev][0], ctx->stream);
    }
    hipLaunchKernelGGL((msm_accumulate_shared<F>),
                       dim3((unsigned)(Bp / bx), (unsigned)W, (unsigned)chunks), dim3(bx), 0,
                       ctx->stream, (const Affine<F>*)bases->table, (const int16_t*)digits, Bp,
                       (uint32_t)n, per_chunk, plan.per_base, (XYZZ<F>*)partial);
    if (ev >= 0) hipEventRecord(es->msm_ev[ev][1], ctx->stream);
    if (ngroups > 1) {
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), ngroups, (unsigned)W), dim3(64),
                         0, ctx->stream, (const XYZZ<F>*)partial, Bp, (uint32_t)chunks, group, mid,
                         chunks * Bp, (size_t)ngroups * Bp);
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1, (unsigned)W), dim3(64), 0,
                         ctx->stream, (const XYZZ<F>*)mid, Bp, ngroups, ngroups, wsum,
                         (size_t)ngroups * Bp, Bp);
    } else {
      hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1, (unsigned)W), dim3(64), 0,
                         ctx->stream, (const XYZZ<F>*)partial, Bp, (uint32_t)chunks,
                         (uint32_t)chunks, wsum, chunks * Bp, Bp);
    }
    if (!wsum_out) {
      HornerArgs<F> ha{};
      ha.wsum[0] = wsum;
      ha.out[0] = out;
      hipLaunchKernelGGL((msm_horner<F>), dim3((unsigned)(Bp / 64), 1), dim3(64), 0, ctx->stream,
                         ha, Bp, plan);
    }
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
  }
  if (wsum_out) {
    ctx->err = "msm: deferred window sums need a shared-table plan";
    return ZKMI_ERR_ARG;
  }
  // 8 x as many chunks as it takes to put 4 waves on every SIMD: all blocks of the coarse grid
  // run for the whole kernel, so a few occupied wave slots (the overlapped solve of the next
  // batch) or uneven clocks cost a full extra round; measured per 1024-proof batch, G1 launches:
  // x1 325 ms, x2 315, x4 299, x8 292 (best end to end), x16 289 + dearer reduction.
  const size_t chunk_factor = bases->chunk_factor ? bases->chunk_factor : 8;
  size_t chunks = chunk_factor * (size_t)262144 / Bp;
  if (chunks < 1) chunks = 1;
  if (chunks > n) chunks = n;
  uint32_t per_chunk = (uint32_t)((n + chunks - 1) / chunks);
  chunks = (n + per_chunk - 1) / per_chunk;
  void* partial;
  // partials + room for the intermediate level of the reduction
  // side bases own their scratch slot -- 18: commitment MSMs on the second stream, 21: the delta
  // multiples on the assembly stream --: the main stream may be running an MSM on slot 6 at the
  // same time
  int rc = ensure_scratch(ctx, bases->side == 2 ? 21 : bases->side ? 18 : 6,
                          (chunks + 256) * Bp * sizeof(XYZZ<F>), &partial);
  if (rc) return rc;
  if (!bases->side && ctx->part_ev_valid[0]) {   // a deferred tail of an earlier MSM may still be reading this buffer
    ZK_HIP(hipStreamWaitEvent(ctx->stream, ctx->part_ev[0], 0));
    ctx->part_ev_valid[0] = false;
  }
  const unsigned bx_cfg = 256;
  const unsigned bx = (Bp % bx_cfg == 0) ? bx_cfg : 64;
  zkmi_ctx::ProveSet* es = (ctx->msm_ev_set >= 0 && !bases->side) ? &ctx->sets[ctx->msm_ev_set] : nullptr;
  const int ev = (es && n > 1 && es->msm_ev_used < 8) ? es->msm_ev_used++ : -1;
  if (ev >= 0) {
    es->msm_ev_group[ev] = bases->group;
    hipEventRecord(es->msm_ev[ev][0], ctx->stream);
  }
  if (n == 1)
    hipLaunchKernelGGL((msm_accumulate<F, true>), dim3((unsigned)(Bp / bx), (unsigned)chunks),
                       dim3(bx), 0, ctx->stream, (const Affine<F>*)bases->table, scalars, row_idx,
                       Bp, (uint32_t)n, per_chunk, bases->plan, (XYZZ<F>*)partial, kmul,
                       (const uint8_t*)bases->inf);
  else
    hipLaunchKernelGGL((msm_accumulate<F, false>), dim3((unsigned)(Bp / bx), (unsigned)chunks),
                       dim3(bx), 0, ctx->stream, (const Affine<F>*)bases->table, scalars, row_idx,
                       Bp, (uint32_t)n, per_chunk, bases->plan, (XYZZ<F>*)partial, kmul,
                       (const uint8_t*)bases->inf);
  if (ev >= 0) hipEventRecord(es->msm_ev[ev][1], ctx->stream);
  // two-level sum of the per-chunk partials (sqrt(chunks) groups) keeps the tail parallel
  uint32_t group = 1;
  while ((size_t)group * group < chunks) group++;
  const uint32_t ngroups = (uint32_t)((chunks + group - 1) / group);
  if (ngroups > 1) {
    XYZZ<F>* mid = (XYZZ<F>*)partial + chunks * Bp;
    hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), ngroups), dim3(64), 0,
                       ctx->stream, (const XYZZ<F>*)partial, Bp, (uint32_t)chunks, group, mid, (size_t)0, (size_t)0);
    hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1), dim3(64), 0, ctx->stream,
                       (const XYZZ<F>*)mid, Bp, ngroups, ngroups, out, (size_t)0, (size_t)0);
  } else {
    hipLaunchKernelGGL((msm_reduce<F>), dim3((unsigned)(Bp / 64), 1), dim3(64), 0, ctx->stream,
                       (const XYZZ<F>*)partial, Bp, (uint32_t)chunks, (uint32_t)chunks, out, (size_t)0,
                       (size_t)0);
  }
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

#if ZK_MSM_PART == 2
template int build_comb<Fq2>(zkmi_ctx*, const Affine<Fq2>*, size_t, const WinPlan&, Affine<Fq2>*,
                             int*, Affine<Fq2>*);
template int build_impl<Fq2>(zkmi_ctx*, const Affine<Fq2>*, size_t, const WinPlan&, Affine<Fq2>*);
template int run_impl<Fq2>(zkmi_ctx*, const zkmi_msm_bases*, const Fr*, const uint32_t*, size_t,
                           XYZZ<Fq2>*, bool, XYZZ<Fq2>*, hipStream_t);
#else
extern template int run_impl<Fq2>(zkmi_ctx*, const zkmi_msm_bases*, const Fr*, const uint32_t*,
                                  size_t, XYZZ<Fq2>*, bool, XYZZ<Fq2>*, hipStream_t);

__global__ void fill_inf_g1(G1XYZZ* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = G1XYZZ::inf();
}
__global__ void fill_inf_g2(G2XYZZ* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = G2XYZZ::inf();
}

// Horner step of up to four deferred MSMs of one group (same plan) in one launch on `stream`
int msm_horner_run(zkmi_ctx* ctx, hipStream_t stream, int count,
                   const zkmi_msm_bases* const* bases, void* const* wsums, void* const* outs,
                   size_t Bp) {
  if (count < 1 || count > 4) return ZKMI_ERR_ARG;
  const int group = bases[0]->group;
  const WinPlan& plan = bases[0]->plan;
  if (group == 1) {
    if (plan.comb) {
      HornerArgsRW<Fq> hw{};
      for (int i = 0; i < count; i++) {
        hw.wsum[i] = (G1XYZZ*)wsums[i];
        hw.out[i] = (G1XYZZ*)outs[i];
        hw.stotal[i] = bases[i]->stotal1;
      }
      launch_comb_horner<Fq>(stream, hw, count, Bp, plan);
    } else {
      HornerArgs<Fq> ha{};
      for (int i = 0; i < count; i++) {
        ha.wsum[i] = (const G1XYZZ*)wsums[i];
        ha.out[i] = (G1XYZZ*)outs[i];
      }
      hipLaunchKernelGGL((msm_horner<Fq>), dim3((unsigned)(Bp / 64), (unsigned)count), dim3(64), 0,
                         stream, ha, Bp, plan);
    }
  } else {
    if (plan.comb) {
      HornerArgsRW<Fq2> hw{};
      for (int i = 0; i < count; i++) {
        hw.wsum[i] = (G2XYZZ*)wsums[i];
        hw.out[i] = (G2XYZZ*)outs[i];
        hw.stotal[i] = bases[i]->stotal2;
      }
      launch_comb_horner<Fq2>(stream, hw, count, Bp, plan);
    } else {
      HornerArgs<Fq2> ha{};
      for (int i = 0; i < count; i++) {
        ha.wsum[i] = (const G2XYZZ*)wsums[i];
        ha.out[i] = (G2XYZZ*)outs[i];
      }
      hipLaunchKernelGGL((msm_horner<Fq2>), dim3((unsigned)(Bp / 64), (unsigned)count), dim3(64),
                         0, stream, ha, Bp, plan);
    }
  }
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

int msm_run(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const Fr* scalars, const uint32_t* row_idx,
            size_t Bp, void* out_xyzz, bool scalars_f, void* wsum_out, hipStream_t finish_stream) {
  if (bases->n == 0 && wsum_out && (bases->plan.shared || bases->plan.comb)) {
    // deferred path: every window sum is the identity
    const size_t cnt = (size_t)bases->plan.W * Bp;
    if (bases->group == 1)
      hipLaunchKernelGGL(fill_inf_g1, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, ctx->stream,
                         (G1XYZZ*)wsum_out, cnt);
    else
      hipLaunchKernelGGL(fill_inf_g2, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, ctx->stream,
                         (G2XYZZ*)wsum_out, cnt);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
  }
  if (bases->n == 0) {
    if (bases->group == 1)
      hipLaunchKernelGGL(fill_inf_g1, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream,
                         (G1XYZZ*)out_xyzz, Bp);
    else
      hipLaunchKernelGGL(fill_inf_g2, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream,
                         (G2XYZZ*)out_xyzz, Bp);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
  }
  if (bases->group == 1)
    return run_impl<Fq>(ctx, bases, scalars, row_idx, Bp, (G1XYZZ*)out_xyzz, scalars_f,
                        (G1XYZZ*)wsum_out, finish_stream);
  return run_impl<Fq2>(ctx, bases, scalars, row_idx, Bp, (G2XYZZ*)out_xyzz, scalars_f,
                       (G2XYZZ*)wsum_out, finish_stream);
}

int xyzz_to_affine(zkmi_ctx* ctx, int group, const void* in, void* out, size_t n) {
  if (n == 0) return ZKMI_OK;
  const unsigned g = (unsigned)((n + 63) / 64);
  if (group == 1)
    hipLaunchKernelGGL((xyzz_to_affine_kernel<Fq>), dim3(g), dim3(64), 0, ctx->stream,
                       (const G1XYZZ*)in, (G1Affine*)out, n);
  else
    hipLaunchKernelGGL((xyzz_to_affine_kernel<Fq2>), dim3(g), dim3(64), 0, ctx->stream,
                       (const G2XYZZ*)in, (G2Affine*)out, n);
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

#endif  // ZK_MSM_PART

}  // namespace zk
