// PLONK prover rounds for a batch of independent witnesses (BASELINE config 5 names the PLONK
// backend).  Stands in for plonk.Prove in gnark backend/plonk/bn254/prove.go [UPSTREAM-RECALL,
// SURVEY.md §3.2, §8 a-2 / f-4]: KZG commitments (G1 MSMs over the SRS: the same fixed-base comb
// tables as the Groth16 key), wire polynomials blinded with multiples of Z_H, grand-product
// permutation argument, quotient on the coset 5<w_4n> split in three, linearisation and two
// openings.  The Fiat-Shamir hashing stays on the host between rounds, as in gnark (the five
// zkmi_plonk_round* calls); everything else runs here.  Protocol: DESIGN.md §PLONK; the reference
// holds no PLONK vector: parity unpinned, pinned instead to oracle/plonk_ref.py and to the pairing
// check of the openings.
//
// Layout: as everywhere, batch-inner.  A polynomial of a batch is a matrix [coefficient][proof]
// with the proof index fastest: lane = proof, so the sequential recurrences of the protocol
// (grand product, Horner evaluation, division by X - zeta) are plain per-lane loops that run for
// the whole batch at once, and every key-side quantity (selectors, sigma, coset points) is
// wave-uniform.  Field elements are in gnark's image (x * 2^256, ff.h) throughout.
#include <algorithm>

#include "zkmi_internal.h"
#include "sha256.h"
#include <cstring>
#include "ff29.h"
#if defined(__HIP_DEVICE_COMPILE__)
#include "ff29_asm.h"
#endif

using namespace zk;

struct zkmi_plonk_pk {
  uint32_t log_n = 0, n_public = 0, max_batch = 0;
  Fr *coef = nullptr, *coset = nullptr, *sigma = nullptr, *omega = nullptr, *coset_x = nullptr,
     *l1 = nullptr, *zh_inv = nullptr;
  zkmi_msm_bases* srs = nullptr;
  zkmi_msm_bases* lag[3] = {nullptr, nullptr, nullptr};   // Lagrange points of a, b, c (optional)
  uint32_t* lag_rows[3] = {nullptr, nullptr, nullptr};    // their scalar rows (device)
  uint32_t* chunk_idx = nullptr;   // 3 x (n + 6): scalar rows of t_lo / t_mid / t_hi in the 4n buffer
  // state of the batch in flight
  size_t batch = 0, Bp = 0;
  int round = 0;
  DevBuf cf[4];      // coefficient forms of a, b, c, z: (n + 8) rows
  DevBuf big[6];     // 4n-row work buffers
  DevBuf small[4];   // challenges / per-proof scalars, MSM outputs
};

namespace zk {

#define LANE const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x

// Products of the element-wise kernels on the 9 x 29-bit form (the single-accumulator asm chain of
// ff29_asm.h: ~365 instructions with unpack and canonical pack, ~540 for ff.h's mul).  The chain
// computes a b 2^-261 where ff.h's mul computes a b 2^-256, so exponents are tracked instead of
// converting data: write im_e(x) = x 2^e mod r (gnark's image is e = 256); then
//     fmulp(im_e1(x), im_e2(y)) = im_(e1 + e2 - 261)(x y),
// sums need equal exponents, and every key-side operand is stored (plonk_pk_load) or every per-proof
// scalar converted once (fmulp by the constant 2^(e' - e + 261)) with the exponent that makes the
// result come out at 256 again.  Each kernel's comments carry the exponents.
__device__ __forceinline__ Fr fmulp(const Fr& a, const Fr& b) {
  Fr r;
#if defined(__HIP_DEVICE_COMPILE__)
  pack_canonical<Fr29Params>(r.v, mul_asm(unpack29<Fr29Params>(a.v), unpack29<Fr29Params>(b.v)));
#else
  r = a;   // device-only helper
#endif
  return r;
}
// 2^e mod r as a canonical integer, e >= 256: Fr::one() doubled e - 256 times (host)
// x 2^k by doublings: im_e(x) -> im_(e + k)(x)
__host__ __device__ static inline Fr lift(Fr x, int k) {
  for (int i = 0; i < k; i++) x = add(x, x);
  return x;
}
__global__ void plonk_lift_kernel(Fr* x, size_t count, int k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) x[i] = lift(x[i], k);
}
static Fr pow2_image(int e) {
  Fr x = Fr::one();
  for (int i = 256; i < e; i++) x = add(x, x);
  return x;
}

// cf[j] -= b_j, cf[n + j] += b_j: adds (b_(nb-1) X^(nb-1) + ... + b_0)(X^n - 1); the blinding rows
// of `blind` are ordered highest power first (b1 X + b2 -> rows r0, r0 + 1)
__global__ void plonk_blind(Fr* cf, size_t n, const Fr* blind, int r0, int nb, size_t Bp) {
  LANE;
  for (int j = 0; j < nb; j++) {
    const Fr bj = bi_ld(blind, r0 + nb - 1 - j, lane, Bp);
    bi_st(cf, j, lane, Bp, sub(bi_ld(cf, j, lane, Bp), bj));
    bi_st(cf, n + j, lane, Bp, add(bi_ld(cf, n + j, lane, Bp), bj));
  }
}

__global__ void plonk_zero_rows(Fr* p, size_t r0, size_t r1, size_t Bp) {
  LANE;
  for (size_t r = r0 + blockIdx.y; r < r1; r += gridDim.y) bi_st(p, r, lane, Bp, Fr::zero());
}

// PI Lagrange values: row j = -x_j for the public inputs (rows 0 .. n_pub-1 of `inputs`), else 0
__global__ void plonk_pi(const Fr* inputs, Fr* pi, size_t n_pub, size_t n, size_t Bp) {
  LANE;
  for (size_t r = blockIdx.y; r < n; r += gridDim.y)
    bi_st(pi, r, lane, Bp, r < n_pub ? neg(bi_ld(inputs, r, lane, Bp)) : Fr::zero());
}

// F[i] = prod_c (w_c,i + beta k_c w^i + gamma), G[i] = prod_c (w_c,i + beta sigma_c,i + gamma)
__global__ __launch_bounds__(64) void plonk_fg(const Fr* a, const Fr* b, const Fr* c,
                                               const Fr* sigma, const Fr* omega, const Fr* ch,
                                               Fr* F, Fr* G, size_t n, size_t Bp, Fr k1, Fr k2) {
  LANE;
  const Fr beta = bi_ld(ch, 0, lane, Bp), gamma = bi_ld(ch, 1, lane, Bp);
  for (size_t i = blockIdx.y; i < n; i += gridDim.y) {
    const Fr bx = mul(beta, omega[i]);
    const Fr wa = add(bi_ld(a, i, lane, Bp), gamma), wb = add(bi_ld(b, i, lane, Bp), gamma),
             wc = add(bi_ld(c, i, lane, Bp), gamma);
    const Fr f = mul(mul(add(wa, bx), add(wb, mul(bx, k1))), add(wc, mul(bx, k2)));
    const Fr g = mul(mul(add(wa, mul(beta, sigma[i])), add(wb, mul(beta, sigma[n + i]))),
                     add(wc, mul(beta, sigma[2 * n + i])));
    bi_st(F, i, lane, Bp, f);
    bi_st(G, i, lane, Bp, g);
  }
}

// per (lane, chunk): ratio[i] = F[i] / G[i] with one inversion per chunk (Montgomery's trick; P is
// prefix scratch), then the chunk's product into cp[chunk]
__global__ __launch_bounds__(64) void plonk_ratio(Fr* F, const Fr* G, Fr* P, Fr* cp, size_t n,
                                                  size_t chunk, size_t Bp) {
  LANE;
  const size_t i0 = (size_t)blockIdx.y * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  Fr acc = Fr::one();
  for (size_t i = i0; i < i1; i++) {
    bi_st(P, i, lane, Bp, acc);
    acc = mul(acc, bi_ld(G, i, lane, Bp));
  }
  Fr inv = inverse(acc);
  for (size_t i = i1; i-- > i0;) {
    const Fr gi = mul(inv, bi_ld(P, i, lane, Bp));
    inv = mul(inv, bi_ld(G, i, lane, Bp));
    bi_st(F, i, lane, Bp, mul(bi_ld(F, i, lane, Bp), gi));
  }
  acc = Fr::one();
  for (size_t i = i0; i < i1; i++) acc = mul(acc, bi_ld(F, i, lane, Bp));
  bi_st(cp, blockIdx.y, lane, Bp, acc);
}
// exclusive prefix product of the chunk products, per lane
__global__ void plonk_chunk_scan(Fr* cp, size_t n_chunks, size_t Bp) {
  LANE;
  Fr acc = Fr::one();
  for (size_t k = 0; k < n_chunks; k++) {
    const Fr v = bi_ld(cp, k, lane, Bp);
    bi_st(cp, k, lane, Bp, acc);
    acc = mul(acc, v);
  }
}
// z[i0] = carry-in, z[i + 1] = z[i] * ratio[i]
__global__ __launch_bounds__(64) void plonk_z_fill(const Fr* ratio, const Fr* cp, Fr* z, size_t n,
                                                   size_t chunk, size_t Bp) {
  LANE;
  const size_t i0 = (size_t)blockIdx.y * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  Fr acc = bi_ld(cp, blockIdx.y, lane, Bp);
  for (size_t i = i0; i < i1; i++) {
    bi_st(z, i, lane, Bp, acc);
    acc = mul(acc, bi_ld(ratio, i, lane, Bp));
  }
}

// quotient on the coset: T[j] = (gate + alpha (p1 - p2) + alpha^2 (z - 1) L1) / Z_H
// key-side arrays (`ks`: qL, qR, qO, qM, qC, S1, S2, S3 on the coset, 4n each) are wave-uniform
__global__ __launch_bounds__(64, 2) void plonk_quotient(const Fr* ea, const Fr* eb, const Fr* ec,
                                                     const Fr* ez, const Fr* epi, const Fr* ks,
                                                     const Fr* xs, const Fr* l1, const Fr* zh_inv,
                                                     const Fr* ch, Fr* T, size_t m, size_t Bp,
                                                     Fr k1, Fr k2, int n_pub_direct, Fr c271,
                                                     Fr c281) {
  // exponents (fmulp above): per-proof data a, b, c, z, pub, beta, gamma, alpha: 256.  Key side as
  // stored by zkmi_plonk_pk_load: qL qR qO S1 S2 S3 (ks rows 0 1 2 5 6 7), xs, l1, zh_inv, k1 = 5,
  // k2 = 25: 261; qM (ks row 3): 266; qC (row 4): 256.
  LANE;
  const Fr beta = bi_ld(ch, 0, lane, Bp), gamma = bi_ld(ch, 1, lane, Bp),
           alpha = bi_ld(ch, 2, lane, Bp);
  const Fr al2 = fmulp(fmulp(alpha, alpha), c271);   // alpha^2: 251 -> 261
  const Fr al_x = fmulp(alpha, c281);                 // alpha: 276, for the 241 of p1 - p2
  // n_pub_direct >= 0: `epi` holds the public inputs x_t themselves (rows 0 .. n_pub - 1) and PI on
  // the coset is evaluated in closed form: L_t(x) = L_0(x / w^t) and x_j / w^t = x_(j - 4t) on the
  // coset 5 <w_4n>, so PI(x_j) = - sum_t x_t L1[j - 4t] -- no transform for a handful of inputs
  Fr pub[8];
  for (int t = 0; t < 8; t++) pub[t] = t < n_pub_direct ? bi_ld(epi, t, lane, Bp) : Fr::zero();
  for (size_t j = blockIdx.y; j < m; j += gridDim.y) {
    const Fr a = bi_ld(ea, j, lane, Bp), b = bi_ld(eb, j, lane, Bp), c = bi_ld(ec, j, lane, Bp),
             z = bi_ld(ez, j, lane, Bp), zw = bi_ld(ez, (j + 4) & (m - 1), lane, Bp);
    Fr gate = add(add(fmulp(ks[j], a), fmulp(ks[m + j], b)), fmulp(ks[2 * m + j], c));   // 256
    Fr pi;
    if (n_pub_direct >= 0) {
      pi = Fr::zero();
      for (int t = 0; t < n_pub_direct; t++)
        pi = sub(pi, fmulp(pub[t], l1[(j - 4 * (size_t)t) & (m - 1)]));                 // 256
    } else {
      pi = bi_ld(epi, j, lane, Bp);
    }
    gate = add(add(gate, fmulp(ks[3 * m + j], fmulp(a, b))), add(ks[4 * m + j], pi));  // 266 + 251 -> 256
    const Fr bx = fmulp(beta, xs[j]);                                                     // 256
    const Fr ag = add(a, gamma), bg = add(b, gamma), cg = add(c, gamma);
    // three 256 factors and z: 251, 246, 241
    const Fr p1 = fmulp(fmulp(fmulp(add(ag, bx), add(bg, fmulp(bx, k1))), add(cg, fmulp(bx, k2))), z);
    const Fr p2 = fmulp(fmulp(fmulp(add(ag, fmulp(beta, ks[5 * m + j])),
                                    add(bg, fmulp(beta, ks[6 * m + j]))),
                              add(cg, fmulp(beta, ks[7 * m + j]))),
                        zw);
    const Fr t3 = fmulp(fmulp(al2, sub(z, Fr::one())), l1[j]);                           // 261, 256 -> 256
    const Fr num = add(add(gate, fmulp(al_x, sub(p1, p2))), t3);                          // 276 + 241 -> 256
    bi_st(T, j, lane, Bp, fmulp(num, zh_inv[j & 3]));
  }
}

// Evaluation of up to six polynomials at per-lane points, in two levels so that no lane walks a
// whole polynomial alone: plonk_chunk_horner evaluates every chunk of CH coefficients at x
// (blockIdx.y = chunk, blockIdx.z = polynomial), plonk_eval_combine folds the chunk values with
// x^CH.  shared != 0: the polynomial is key-side (one coefficient array for all proofs).
constexpr uint32_t CH = 512;
struct EvalArgs {
  const Fr* poly[6];
  uint32_t len[6];
  uint8_t shared[6];
  uint8_t point_row[6];   // row of `pts` holding the evaluation point
};
__global__ __launch_bounds__(64) void plonk_chunk_horner(EvalArgs args, const Fr* pts, Fr* part,
                                                         uint32_t n_chunks, size_t Bp) {
  LANE;
  const int k = blockIdx.z;
  const uint32_t c = blockIdx.y, len = args.len[k];
  const uint32_t i0 = c * CH, i1 = i0 + CH < len ? i0 + CH : len;
  const Fr x = bi_ld(pts, args.point_row[k], lane, Bp);
  const Fr* p = args.poly[k];
  Fr acc = Fr::zero();
  if (args.shared[k]) {
    for (uint32_t i = i1; i-- > i0;) acc = add(mul(acc, x), p[i]);
  } else {
    for (uint32_t i = i1; i-- > i0;) acc = add(mul(acc, x), bi_ld(p, i, lane, Bp));
  }
  bi_st(part, (size_t)k * n_chunks + c, lane, Bp, acc);   // 0 for chunks past the end
}
__global__ __launch_bounds__(64) void plonk_eval_combine(EvalArgs args, const Fr* pts,
                                                         const Fr* part, Fr* out,
                                                         uint32_t n_chunks, size_t Bp) {
  LANE;
  const int k = blockIdx.y;
  Fr xl = bi_ld(pts, args.point_row[k], lane, Bp);
  for (uint32_t t = 1; t < CH; t <<= 1) xl = sqr(xl);   // x^CH
  Fr acc = Fr::zero();
  for (uint32_t c = n_chunks; c-- > 0;)
    acc = add(mul(acc, xl), bi_ld(part, (size_t)k * n_chunks + c, lane, Bp));
  bi_st(out, k, lane, Bp, acc);
}

// numerators of the two openings.  sc rows: 0 qm, 1 ql, 2 qr, 3 qo, 4 s3, 5 z, 6 tlo, 7 tmid, 8 thi,
// 9 c0 (constant term: r0 - sum_i v^i e_i), 10 v, 11 zeta, 12 zeta*w, 13 z(zeta w)
__global__ __launch_bounds__(64, 2) void plonk_lin(const Fr* ca, const Fr* cb, const Fr* cc,
                                                const Fr* cz, const Fr* t, const Fr* kc,
                                                const Fr* sc, Fr* N, Fr* NZ, size_t n, size_t Bp,
                                                Fr c266) {
  // every product is (per-proof scalar) x (coefficient at 256): the scalars are lifted to 261 once
  // per lane (fmulp by 2^266: 256 + 266 - 261), the products then land at 256 (see fmulp)
  LANE;
  const Fr v = bi_ld(sc, 10, lane, Bp);
  const Fr vF = fmulp(v, c266);
  const Fr v2 = fmulp(vF, v), v3 = fmulp(vF, v2), v4 = fmulp(vF, v3), v5 = fmulp(vF, v4);   // 256
  const Fr v2F = fmulp(v2, c266), v3F = fmulp(v3, c266), v4F = fmulp(v4, c266), v5F = fmulp(v5, c266);
#define SCF(row) fmulp(bi_ld(sc, (row), lane, Bp), c266)
  const Fr sqm = SCF(0), sql = SCF(1), sqr_ = SCF(2), sqo = SCF(3), ss3 = SCF(4), sz = SCF(5),
           stl = SCF(6), stm = SCF(7), sth = SCF(8);
#undef SCF
  const size_t L = n + 3;
  for (size_t i = blockIdx.y; i < L; i += gridDim.y) {
    Fr r = Fr::zero();
    if (i < n) {   // key polynomials: qL qR qO qM qC S1 S2 S3 coefficient forms, n each
      r = add(add(fmulp(sql, kc[i]), fmulp(sqr_, kc[n + i])),
              add(fmulp(sqo, kc[2 * n + i]), fmulp(sqm, kc[3 * n + i])));
      r = add(r, add(kc[4 * n + i], fmulp(ss3, kc[7 * n + i])));
      r = add(r, add(fmulp(v4F, kc[5 * n + i]), fmulp(v5F, kc[6 * n + i])));
    }
    const Fr zi = bi_ld(cz, i, lane, Bp);
    r = add(r, fmulp(sz, zi));
    if (i < n + 2) {
      r = add(r, add(fmulp(stl, bi_ld(t, i, lane, Bp)),
                     add(fmulp(stm, bi_ld(t, n + 2 + i, lane, Bp)),
                         fmulp(sth, bi_ld(t, 2 * n + 4 + i, lane, Bp)))));
      r = add(r, add(fmulp(vF, bi_ld(ca, i, lane, Bp)),
                     add(fmulp(v2F, bi_ld(cb, i, lane, Bp)), fmulp(v3F, bi_ld(cc, i, lane, Bp)))));
    }
    if (i == 0) r = add(r, bi_ld(sc, 9, lane, Bp));
    bi_st(N, i, lane, Bp, r);
    bi_st(NZ, i, lane, Bp, i == 0 ? sub(zi, bi_ld(sc, 13, lane, Bp)) : zi);
  }
}

// q = p / (X - x), exact: q[k-1] = p[k] + x q[k], in chunks of CH coefficients: the carry into
// chunk c is the suffix value S_c = sum_{k >= (c+1) CH} p[k] x^(k - (c+1) CH), a scan over the chunk
// values of plonk_chunk_horner (plonk_div_suffix), then every chunk runs its own recurrence
// (plonk_div_local).  `which` = blockIdx.z: 0 = (N, zeta), 1 = (NZ, zeta w).
__global__ __launch_bounds__(64) void plonk_div_suffix(const Fr* pts, Fr* part, uint32_t n_chunks,
                                                       size_t Bp) {
  LANE;
  const int k = blockIdx.y;
  Fr xl = bi_ld(pts, k, lane, Bp);
  for (uint32_t t = 1; t < CH; t <<= 1) xl = sqr(xl);
  Fr s = Fr::zero();   // S for the last chunk
  for (uint32_t c = n_chunks; c-- > 0;) {
    const Fr h = bi_ld(part, (size_t)k * n_chunks + c, lane, Bp);
    bi_st(part, (size_t)k * n_chunks + c, lane, Bp, s);
    s = add(h, mul(xl, s));
  }
}
__global__ __launch_bounds__(64) void plonk_div_local(const Fr* N, const Fr* NZ, const Fr* pts,
                                                      const Fr* part, Fr* W, Fr* WZ, uint32_t len,
                                                      uint32_t n_chunks, size_t Bp) {
  LANE;
  const int k = blockIdx.z;
  const uint32_t c = blockIdx.y;
  const Fr* p = k ? NZ : N;
  Fr* q = k ? WZ : W;
  const Fr x = bi_ld(pts, k, lane, Bp);
  const uint32_t i0 = c * CH, i1 = i0 + CH < len ? i0 + CH : len;
  Fr carry = bi_ld(part, (size_t)k * n_chunks + c, lane, Bp);   // q at index i1 - 1
  for (uint32_t i = i1; i-- > i0;) {
    if (i + 1 < len) bi_st(q, i, lane, Bp, carry);
    else bi_st(q, i, lane, Bp, Fr::zero());
    carry = add(bi_ld(p, i, lane, Bp), mul(x, carry));
  }
}

static int buf(zkmi_ctx* ctx, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes) return ZKMI_OK;
  if (b.p) {
    hipStreamSynchronize(ctx->stream);
    hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
  if (hipMalloc(&b.p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "plonk: hipMalloc(" + std::to_string(bytes) + " B) failed";
    return ZKMI_ERR_OOM;
  }
  b.bytes = bytes;
  return ZKMI_OK;
}

static Fr fr_small(uint32_t v) {
  Fr x = Fr::zero();
  x.v[0] = v;
  return to_mont(x);
}

// commitment of `count` polynomials whose coefficient rows start at scal[k] (row_idx[k] maps SRS
// power i to a row, or null): MSM over the SRS, affine, copied to out (proof-major, 64 B each)
static int commit(zkmi_ctx* ctx, zkmi_plonk_pk* pk, int count, const Fr* const* scal,
                  const uint32_t* const* row_idx, void* out_host_or_dev, size_t batch) {
  const size_t Bp = pk->Bp;
  int rc;
  if ((rc = buf(ctx, pk->small[2], Bp * 128)) || (rc = buf(ctx, pk->small[3], Bp * 64))) return rc;
  for (int k = 0; k < count; k++) {
    if ((rc = msm_run(ctx, pk->srs, scal[k], row_idx ? row_idx[k] : nullptr, Bp, pk->small[2].p)))
      return rc;
    if ((rc = xyzz_to_affine(ctx, 1, pk->small[2].p, pk->small[3].p, Bp))) return rc;
    // [lane] affine -> out[(lane * count + k)]
    ZK_HIP(hipMemcpy2DAsync((char*)out_host_or_dev + (size_t)k * 64, (size_t)count * 64,
                            pk->small[3].p, 64, 64, batch, hipMemcpyDefault, ctx->stream));
  }
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

}  // namespace zk

extern "C" {

void zkmi_plonk_pk_free(zkmi_ctx* ctx, zkmi_plonk_pk* pk) {
  if (!pk) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
  }
  for (Fr* p : {pk->coef, pk->coset, pk->sigma, pk->omega, pk->coset_x, pk->l1, pk->zh_inv})
    if (p) hipFree(p);
  if (pk->chunk_idx) hipFree(pk->chunk_idx);
  zkmi_msm_bases_free(ctx, pk->srs);
  for (int c = 0; c < 3; c++) {
    zkmi_msm_bases_free(ctx, pk->lag[c]);
    if (pk->lag_rows[c]) hipFree(pk->lag_rows[c]);
  }
  for (auto& b : pk->cf)
    if (b.p) hipFree(b.p);
  for (auto& b : pk->big)
    if (b.p) hipFree(b.p);
  for (auto& b : pk->small)
    if (b.p) hipFree(b.p);
  delete pk;
}

int zkmi_plonk_pk_load(zkmi_ctx* ctx, const zkmi_plonk_pk_desc* d, zkmi_plonk_pk** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!d || !out) return ZKMI_ERR_ARG;
  if (d->log_n < 4 || d->log_n > 24) {
    ctx->err = "plonk pk: log_n out of range [4,24]";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << d->log_n, m = 4 * n;
  if (d->n_public >= n) {
    ctx->err = "plonk pk: more public inputs than gates";
    return ZKMI_ERR_ARG;
  }
  auto* pk = new zkmi_plonk_pk();
  pk->log_n = d->log_n;
  pk->n_public = d->n_public;
  pk->max_batch = d->max_batch ? d->max_batch : 64;
  struct { Fr** dst; const void* src; size_t count; } up[7] = {
      {&pk->coef, d->coef, 8 * n},   {&pk->coset, d->coset, 8 * m},   {&pk->sigma, d->sigma, 3 * n},
      {&pk->omega, d->omega, n},     {&pk->coset_x, d->coset_x, m},   {&pk->l1, d->l1_coset, m},
      {&pk->zh_inv, d->zh_inv, 4}};
  for (auto& u : up) {
    if (!u.src || hipMalloc((void**)u.dst, u.count * 32) != hipSuccess ||
        hipMemcpy(*u.dst, u.src, u.count * 32, hipMemcpyDefault) != hipSuccess) {
      ctx->err = "plonk pk: upload failed";
      zkmi_plonk_pk_free(ctx, pk);
      return ZKMI_ERR_HIP;
    }
  }
  // exponents the quotient kernel expects (plonk_quotient): 2^261 images of qL qR qO S1 S2 S3 on the
  // coset, of the coset points, L_1 and 1 / Z_H; 2^266 of qM; qC as uploaded
  {
    struct { Fr* p; size_t count; int k; } lifts[7] = {
        {pk->coset, 3 * m, 5},   {pk->coset + 3 * m, m, 10}, {pk->coset + 5 * m, 3 * m, 5},
        {pk->coset_x, m, 5},     {pk->l1, m, 5},             {pk->zh_inv, 4, 5},
        {nullptr, 0, 0}};
    for (auto& l : lifts)
      if (l.count)
        hipLaunchKernelGGL(plonk_lift_kernel, dim3((unsigned)((l.count + 255) / 256)), dim3(256), 0,
                           ctx->stream, l.p, l.count, l.k);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      ctx->err = "plonk pk: key scaling failed";
      zkmi_plonk_pk_free(ctx, pk);
      return ZKMI_ERR_HIP;
    }
  }
  // scalar-row maps of the three quotient chunks: power i of chunk c -> row c (n + 2) + i of the
  // 4n-row coefficient buffer; powers n+2 .. n+5 -> its last row (zero: deg t < 3n + 6)
  {
    std::vector<uint32_t> idx(3 * (n + 6));
    for (size_t c = 0; c < 3; c++)
      for (size_t i = 0; i < n + 6; i++)
        idx[c * (n + 6) + i] = (uint32_t)(i < n + 2 ? c * (n + 2) + i : m - 1);
    if (hipMalloc((void**)&pk->chunk_idx, idx.size() * 4) != hipSuccess ||
        hipMemcpy(pk->chunk_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      zkmi_plonk_pk_free(ctx, pk);
      return ZKMI_ERR_HIP;
    }
  }
  int rc = zkmi_msm_bases_load(ctx, 1, d->srs_g1, n + 6, (int)d->window_bits, &pk->srs);
  if (rc) {
    zkmi_plonk_pk_free(ctx, pk);
    return rc;
  }
  if (d->lag_k) {   // after the SRS tables: these are small (2^k / k entries per point)
    if (d->lag_k < 4 || d->lag_k > 16) {
      ctx->err = "plonk pk: lag_k must be 0 or in [4,16]";
      zkmi_plonk_pk_free(ctx, pk);
      return ZKMI_ERR_ARG;
    }
    for (int c = 0; c < 3; c++) {
      if (!d->lag_g1[c] || !d->lag_rows[c]) {
        ctx->err = "plonk pk: lag_k set without lag_g1 / lag_rows";
        zkmi_plonk_pk_free(ctx, pk);
        return ZKMI_ERR_ARG;
      }
      std::vector<uint32_t> rows(n + 2);
      if (hipMemcpy(rows.data(), d->lag_rows[c], (n + 2) * 4, hipMemcpyDefault) != hipSuccess) {
        zkmi_plonk_pk_free(ctx, pk);
        return ZKMI_ERR_HIP;
      }
      for (uint32_t r : rows)
        if (r >= n + 2) {
          ctx->err = "plonk pk: lag_rows entry out of range";
          zkmi_plonk_pk_free(ctx, pk);
          return ZKMI_ERR_ARG;
        }
      if (hipMalloc((void**)&pk->lag_rows[c], (n + 2) * 4) != hipSuccess ||
          hipMemcpy(pk->lag_rows[c], rows.data(), (n + 2) * 4, hipMemcpyHostToDevice) != hipSuccess) {
        zkmi_plonk_pk_free(ctx, pk);
        return ZKMI_ERR_HIP;
      }
      // 200 + k: subset-sum comb tables over groups of k points (zero digits are skipped)
      if ((rc = zkmi_msm_bases_load(ctx, 1, d->lag_g1[c], n + 2, 200 + (int)d->lag_k, &pk->lag[c]))) {
        zkmi_plonk_pk_free(ctx, pk);
        return rc;
      }
    }
  }
  *out = pk;
  return ZKMI_OK;
}

// staging buffer of one round call: freed on every exit path, after the stream has drained
struct RoundTmp {
  zkmi_ctx* ctx;
  void* p = nullptr;
  explicit RoundTmp(zkmi_ctx* c) : ctx(c) {}
  int alloc(size_t bytes) {
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      ctx->err = "plonk: hipMalloc(" + std::to_string(bytes) + " B) failed";
      return ZKMI_ERR_OOM;
    }
    return ZKMI_OK;
  }
  ~RoundTmp() {
    if (p) {
      hipStreamSynchronize(ctx->stream);
      hipFree(p);
    }
  }
};

// Round 1: witness solve (the constraint system's rows are the gate columns a, b, c), blinding,
// commitments [a], [b], [c].
int zkmi_plonk_round1(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const zkmi_cs* cs, const void* inputs,
                      size_t batch, const void* blind, void* commits_out, int32_t* status_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted Groth16 batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (batch == 0 || batch > pk->max_batch) {
    ctx->err = "plonk: batch must be in [1, max_batch]";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n, m = 4 * n, Bp = round_up(batch, 64);
  if (cs->n_constraints > n || cs->n_public - 1 != pk->n_public) {
    ctx->err = "plonk: constraint system does not match the key";
    return ZKMI_ERR_ARG;
  }
  pk->batch = batch;
  pk->Bp = Bp;
  pk->round = 0;
  int rc;
  for (auto& b : pk->cf)
    if ((rc = buf(ctx, b, (n + 8) * Bp * 32))) return rc;
  for (auto& b : pk->big)
    if ((rc = buf(ctx, b, m * Bp * 32))) return rc;
  if ((rc = buf(ctx, pk->small[0], 16 * Bp * 32)) || (rc = buf(ctx, pk->small[1], 16 * Bp * 32)))
    return rc;
  const size_t n_in = cs->n_public - 1 + cs->n_secret, nc = cs->n_constraints;
  // stage inputs + blinding scalars (host or device, proof-major)
  // value file and gate columns: big[0] = slots (reused later), big[1..3] = a, b, c; every size is
  // checked before anything is staged
  if ((size_t)cs->n_slots > m || n_in > m) {
    ctx->err = "plonk: value file or input vector larger than the 4n work buffer";
    return ZKMI_ERR_ARG;
  }
  RoundTmp tin(ctx), tbl(ctx);
  if ((rc = tin.alloc(batch * n_in * 32)) || (rc = tbl.alloc(batch * 9 * 32))) return rc;
  void *in_dev = tin.p, *bl_dev = tbl.p;
  ZK_HIP(hipMemcpyAsync(in_dev, inputs, batch * n_in * 32, hipMemcpyDefault, ctx->stream));
  ZK_HIP(hipMemcpyAsync(bl_dev, blind, batch * 9 * 32, hipMemcpyDefault, ctx->stream));
  Fr* slots = (Fr*)pk->big[0].p;
  Fr *A = (Fr*)pk->big[1].p, *B = (Fr*)pk->big[2].p, *C = (Fr*)pk->big[3].p;
  void* st;
  if ((rc = ensure_scratch(ctx, 5, Bp * 4, &st))) return rc;
  rc = transpose_in(ctx, in_dev, slots + Bp, n_in, batch, Bp, 32);
  // the inputs in gnark's image, for PI(X) in round 3: big[5] rows 0 .. n_pub - 1 are the public ones
  if (!rc) rc = transpose_in(ctx, in_dev, pk->big[5].p, n_in, batch, Bp, 32);
  if (!rc) rc = rows_to_f_domain(ctx, slots + Bp, n_in, Bp);
  if (!rc) rc = transpose_in(ctx, bl_dev, pk->small[0].p, 9, batch, Bp, 32);
  if (!rc) rc = solve_bi(ctx, cs, slots, A, B, C, (int32_t*)st, Bp);
  if (!rc) rc = rows_to_std_domain(ctx, A, nc, Bp);
  if (!rc) rc = rows_to_std_domain(ctx, B, nc, Bp);
  if (!rc) rc = rows_to_std_domain(ctx, C, nc, Bp);
  // rows the solver does not write (padding gates) must read as zero in round 2
  if (!rc && nc < n)
    for (Fr* col : {A, B, C})
      hipLaunchKernelGGL(plonk_zero_rows, dim3((unsigned)(Bp / 64), 64), dim3(64), 0, ctx->stream,
                         col, nc, n, Bp);
  hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  ZK_HIP(hipMemcpy(status_out, st, batch * 4, hipMemcpyDefault));
  NttPlan* plan;
  if ((rc = get_plan(ctx, (int)pk->log_n, &plan))) return rc;
  const dim3 gl((unsigned)(Bp / 64));
  // columns (rows >= nc read as zero) -> coefficients -> blinded
  Fr* cols[3] = {A, B, C};
  for (int k = 0; k < 3; k++) {
    Fr* cf = (Fr*)pk->cf[k].p;
    if ((rc = ntt_bi(ctx, plan, cols[k], cf, Bp, true, false, nc))) return rc;
    hipLaunchKernelGGL(plonk_zero_rows, dim3((unsigned)(Bp / 64), 8), dim3(64), 0, ctx->stream, cf, n,
                       n + 8, Bp);
    hipLaunchKernelGGL(plonk_blind, gl, dim3(64), 0, ctx->stream, cf, n, (const Fr*)pk->small[0].p,
                       2 * k, 2, Bp);
  }
  ZK_HIP(hipGetLastError());
  if (pk->lag[0]) {
    // Lagrange basis: the columns' values are the scalars; rows n, n + 1 of a column buffer take
    // its blinding scalars b1, b2 (points [tau^(n+1) - tau], [tau^n - 1])
    if ((rc = buf(ctx, pk->small[2], Bp * 128)) || (rc = buf(ctx, pk->small[3], Bp * 64))) return rc;
    for (int k = 0; k < 3; k++) {
      ZK_HIP(hipMemcpyAsync(cols[k] + n * Bp, (const Fr*)pk->small[0].p + (size_t)(2 * k) * Bp,
                            2 * Bp * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
      if ((rc = msm_run(ctx, pk->lag[k], cols[k], pk->lag_rows[k], Bp, pk->small[2].p))) return rc;
      if ((rc = xyzz_to_affine(ctx, 1, pk->small[2].p, pk->small[3].p, Bp))) return rc;
      ZK_HIP(hipMemcpy2DAsync((char*)commits_out + (size_t)k * 64, (size_t)3 * 64, pk->small[3].p, 64,
                              64, batch, hipMemcpyDefault, ctx->stream));
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    const Fr* sc[3] = {(Fr*)pk->cf[0].p, (Fr*)pk->cf[1].p, (Fr*)pk->cf[2].p};
    if ((rc = commit(ctx, pk, 3, sc, nullptr, commits_out, batch))) return rc;
  }
  pk->round = 1;
  return ZKMI_OK;
}

// Round 2: beta, gamma (batch x 2 fr) -> grand product z, blinded, commitment [z].
// The gate columns are still in big[1..3] (Lagrange values).
int zkmi_plonk_round2(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* beta_gamma, void* commit_z_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (pk->round != 1) {
    ctx->err = "plonk: round 2 out of order";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n, Bp = pk->Bp, batch = pk->batch;
  int rc;
  RoundTmp t2(ctx);
  if ((rc = t2.alloc(batch * 64))) return rc;
  void* tmp = t2.p;
  ZK_HIP(hipMemcpyAsync(tmp, beta_gamma, batch * 64, hipMemcpyDefault, ctx->stream));
  Fr* ch = (Fr*)pk->small[1].p;   // rows: 0 beta, 1 gamma, 2 alpha, 3 zeta ...
  rc = transpose_in(ctx, tmp, ch, 2, batch, Bp, 32);
  if (rc) return rc;
  // the solver leaves rows >= n_constraints of the columns untouched: they must read as zero here
  Fr *A = (Fr*)pk->big[1].p, *B = (Fr*)pk->big[2].p, *C = (Fr*)pk->big[3].p;
  Fr *F = (Fr*)pk->big[0].p, *G = F + n * Bp, *P = G + n * Bp, *cp = P + n * Bp;   // 4n rows in all
  const size_t chunk = 256, n_chunks = (n + chunk - 1) / chunk;
  const unsigned gy = (unsigned)(n < 4096 ? n : 4096);
  hipLaunchKernelGGL(plonk_fg, dim3((unsigned)(Bp / 64), gy), dim3(64), 0, ctx->stream, A, B, C,
                     pk->sigma, pk->omega, ch, F, G, n, Bp, fr_small(5), fr_small(25));
  hipLaunchKernelGGL(plonk_ratio, dim3((unsigned)(Bp / 64), (unsigned)n_chunks), dim3(64), 0,
                     ctx->stream, F, G, P, cp, n, chunk, Bp);
  hipLaunchKernelGGL(plonk_chunk_scan, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, cp,
                     n_chunks, Bp);
  Fr* Z = (Fr*)pk->big[4].p;
  hipLaunchKernelGGL(plonk_z_fill, dim3((unsigned)(Bp / 64), (unsigned)n_chunks), dim3(64), 0,
                     ctx->stream, F, cp, Z, n, chunk, Bp);
  ZK_HIP(hipGetLastError());
  NttPlan* plan;
  if ((rc = get_plan(ctx, (int)pk->log_n, &plan))) return rc;
  Fr* cz = (Fr*)pk->cf[3].p;
  if ((rc = ntt_bi(ctx, plan, Z, cz, Bp, true, false, n))) return rc;
  hipLaunchKernelGGL(plonk_zero_rows, dim3((unsigned)(Bp / 64), 8), dim3(64), 0, ctx->stream, cz, n,
                     n + 8, Bp);
  hipLaunchKernelGGL(plonk_blind, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, cz, n,
                     (const Fr*)pk->small[0].p, 6, 3, Bp);
  ZK_HIP(hipGetLastError());
  const Fr* sc[1] = {cz};
  if ((rc = commit(ctx, pk, 1, sc, nullptr, commit_z_out, batch))) return rc;
  pk->round = 2;
  return ZKMI_OK;
}

// Round 3: alpha (batch fr) -> quotient, commitments [t_lo], [t_mid], [t_hi].
int zkmi_plonk_round3(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* alpha, void* commits_t_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (pk->round != 2) {
    ctx->err = "plonk: round 3 out of order";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n, m = 4 * n, Bp = pk->Bp, batch = pk->batch;
  int rc;
  RoundTmp t3(ctx);
  if ((rc = t3.alloc(batch * 32))) return rc;
  void* tmp = t3.p;
  ZK_HIP(hipMemcpyAsync(tmp, alpha, batch * 32, hipMemcpyDefault, ctx->stream));
  Fr* ch = (Fr*)pk->small[1].p;
  rc = transpose_in(ctx, tmp, ch + 2 * Bp, 1, batch, Bp, 32);   // row 2
  if (rc) return rc;
  NttPlan *plan_n, *plan_m;
  if ((rc = get_plan(ctx, (int)pk->log_n, &plan_n)) ||
      (rc = get_plan(ctx, (int)pk->log_n + 2, &plan_m)))
    return rc;
  // PI(X): Lagrange values (-x_j) from the public-input rows saved in big[5] -> coefficients
  // (big[0]) -> coset evaluations (big[5])
  // (circuits with more than eight public inputs; otherwise plonk_quotient evaluates PI itself)
  const int n_pub_direct = pk->n_public <= 8 ? (int)pk->n_public : -1;
  Fr *EA = (Fr*)pk->big[1].p, *EB = (Fr*)pk->big[2].p, *EC = (Fr*)pk->big[3].p,
     *EZ = (Fr*)pk->big[4].p, *EPI = (Fr*)pk->big[5].p;
  if (n_pub_direct < 0) {
    Fr* pil = (Fr*)pk->big[0].p;
    hipLaunchKernelGGL(plonk_pi, dim3((unsigned)(Bp / 64), (unsigned)(n < 4096 ? n : 4096)),
                       dim3(64), 0, ctx->stream, (const Fr*)pk->big[5].p, pil, (size_t)pk->n_public, n,
                       Bp);
    ZK_HIP(hipGetLastError());
    Fr* pic = pil + n * Bp;
    if ((rc = ntt_bi(ctx, plan_n, pil, pic, Bp, true, false, n)) ||
        (rc = ntt_bi(ctx, plan_m, pic, EPI, Bp, false, true, n)))
      return rc;
  }
  if ((rc = ntt_bi(ctx, plan_m, (Fr*)pk->cf[0].p, EA, Bp, false, true, n + 2)) ||
      (rc = ntt_bi(ctx, plan_m, (Fr*)pk->cf[1].p, EB, Bp, false, true, n + 2)) ||
      (rc = ntt_bi(ctx, plan_m, (Fr*)pk->cf[2].p, EC, Bp, false, true, n + 2)) ||
      (rc = ntt_bi(ctx, plan_m, (Fr*)pk->cf[3].p, EZ, Bp, false, true, n + 3)))
    return rc;
  Fr* T = (Fr*)pk->big[0].p;
  hipLaunchKernelGGL(plonk_quotient, dim3((unsigned)(Bp / 64), (unsigned)(m < 8192 ? m : 8192)),
                     dim3(64), 0, ctx->stream, EA, EB, EC, EZ, EPI, pk->coset, pk->coset_x, pk->l1,
                     pk->zh_inv, ch, T, m, Bp, lift(fr_small(5), 5),
                     lift(fr_small(25), 5), n_pub_direct, pow2_image(271), pow2_image(281));
  ZK_HIP(hipGetLastError());
  // coset interpolation: t coefficients in big[1]
  Fr* ct = (Fr*)pk->big[1].p;
  if ((rc = ntt_bi(ctx, plan_m, T, ct, Bp, true, true, m))) return rc;
  const Fr* sc[3] = {ct, ct, ct};
  const uint32_t* ri[3] = {pk->chunk_idx, pk->chunk_idx + (n + 6), pk->chunk_idx + 2 * (n + 6)};
  if ((rc = commit(ctx, pk, 3, sc, ri, commits_t_out, batch))) return rc;
  pk->round = 3;
  return ZKMI_OK;
}

// Round 4: zeta (batch fr) -> a(zeta), b(zeta), c(zeta), S1(zeta), S2(zeta), z(zeta w)
int zkmi_plonk_round4(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* zeta_zetaw, void* evals_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (pk->round != 3) {
    ctx->err = "plonk: round 4 out of order";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n, Bp = pk->Bp, batch = pk->batch;
  int rc;
  RoundTmp t4(ctx), t4o(ctx);
  if ((rc = t4.alloc(batch * 64)) || (rc = t4o.alloc(batch * 6 * 32))) return rc;
  void* tmp = t4.p;
  ZK_HIP(hipMemcpyAsync(tmp, zeta_zetaw, batch * 64, hipMemcpyDefault, ctx->stream));
  Fr* pts = (Fr*)pk->small[1].p + 4 * Bp;   // rows 4, 5 of the challenge block: zeta, zeta w
  rc = transpose_in(ctx, tmp, pts, 2, batch, Bp, 32);
  if (rc) return rc;
  EvalArgs ea{};
  const Fr* polys[6] = {(Fr*)pk->cf[0].p, (Fr*)pk->cf[1].p, (Fr*)pk->cf[2].p, pk->coef + 5 * n,
                        pk->coef + 6 * n, (Fr*)pk->cf[3].p};
  const uint32_t lens[6] = {(uint32_t)n + 2, (uint32_t)n + 2, (uint32_t)n + 2, (uint32_t)n,
                            (uint32_t)n, (uint32_t)n + 3};
  for (int k = 0; k < 6; k++) {
    ea.poly[k] = polys[k];
    ea.len[k] = lens[k];
    ea.shared[k] = (k == 3 || k == 4) ? 1 : 0;
    ea.point_row[k] = k == 5 ? 1 : 0;
  }
  Fr* ev = (Fr*)pk->small[0].p;   // rows 0 .. 5 (the blinding rows are no longer needed)
  const uint32_t n_chunks = (uint32_t)((n + 3 + CH - 1) / CH);
  Fr* part = (Fr*)pk->big[2].p;   // 6 * n_chunks rows
  hipLaunchKernelGGL(plonk_chunk_horner, dim3((unsigned)(Bp / 64), n_chunks, 6), dim3(64), 0,
                     ctx->stream, ea, (const Fr*)pts, part, n_chunks, Bp);
  hipLaunchKernelGGL(plonk_eval_combine, dim3((unsigned)(Bp / 64), 6), dim3(64), 0, ctx->stream, ea,
                     (const Fr*)pts, (const Fr*)part, ev, n_chunks, Bp);
  ZK_HIP(hipGetLastError());
  void* out_dev = t4o.p;
  rc = transpose_out(ctx, ev, out_dev, 6, batch, Bp, 32);
  if (!rc && hipMemcpyAsync(evals_out, out_dev, batch * 6 * 32, hipMemcpyDefault, ctx->stream) !=
                 hipSuccess)
    rc = ZKMI_ERR_HIP;
  hipStreamSynchronize(ctx->stream);
  if (rc) return rc;
  pk->round = 4;
  return ZKMI_OK;
}

// Round 5: per-proof scalars of the linearisation polynomial and of the openings (batch x 14 fr,
// see plonk_lin) -> commitments [W_zeta], [W_zeta_w]
int zkmi_plonk_round5(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* scalars, void* commits_w_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (pk->round != 4) {
    ctx->err = "plonk: round 5 out of order";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n, Bp = pk->Bp, batch = pk->batch;
  int rc;
  RoundTmp t5(ctx);
  if ((rc = t5.alloc(batch * 14 * 32))) return rc;
  void* tmp = t5.p;
  ZK_HIP(hipMemcpyAsync(tmp, scalars, batch * 14 * 32, hipMemcpyDefault, ctx->stream));
  Fr* sc = (Fr*)pk->small[0].p;
  rc = transpose_in(ctx, tmp, sc, 14, batch, Bp, 32);
  if (rc) return rc;
  const size_t L = n + 3;
  // numerators in big[2], quotients in big[3]: 2 (n + 8) rows each (log_n >= 4)
  Fr* N = (Fr*)pk->big[2].p;
  Fr* NZ = N + (n + 8) * Bp;
  Fr* W = (Fr*)pk->big[3].p;
  Fr* WZ = W + (n + 8) * Bp;
  hipLaunchKernelGGL(plonk_lin, dim3((unsigned)(Bp / 64), (unsigned)(L < 4096 ? L : 4096)),
                     dim3(64), 0, ctx->stream, (const Fr*)pk->cf[0].p, (const Fr*)pk->cf[1].p,
                     (const Fr*)pk->cf[2].p, (const Fr*)pk->cf[3].p, (const Fr*)pk->big[1].p,
                     (const Fr*)pk->coef, (const Fr*)sc, N, NZ, n, Bp, pow2_image(266));
  hipLaunchKernelGGL(plonk_zero_rows, dim3((unsigned)(Bp / 64), 8), dim3(64), 0, ctx->stream, W, L - 1,
                     n + 8, Bp);
  hipLaunchKernelGGL(plonk_zero_rows, dim3((unsigned)(Bp / 64), 8), dim3(64), 0, ctx->stream, WZ,
                     L - 1, n + 8, Bp);
  {
    // chunk values of N at zeta and of NZ at zeta w (rows 11, 12 of sc), suffix scan, local division
    const uint32_t n_chunks = (uint32_t)((L + CH - 1) / CH);
    Fr* part = (Fr*)pk->big[4].p;
    const Fr* pts = sc + 11 * Bp;
    EvalArgs ea{};
    ea.poly[0] = N;
    ea.poly[1] = NZ;
    ea.len[0] = ea.len[1] = (uint32_t)L;
    ea.point_row[0] = 0;
    ea.point_row[1] = 1;
    hipLaunchKernelGGL(plonk_chunk_horner, dim3((unsigned)(Bp / 64), n_chunks, 2), dim3(64), 0,
                       ctx->stream, ea, pts, part, n_chunks, Bp);
    hipLaunchKernelGGL(plonk_div_suffix, dim3((unsigned)(Bp / 64), 2), dim3(64), 0, ctx->stream, pts,
                       part, n_chunks, Bp);
    hipLaunchKernelGGL(plonk_div_local, dim3((unsigned)(Bp / 64), n_chunks, 2), dim3(64), 0,
                       ctx->stream, (const Fr*)N, (const Fr*)NZ, pts, (const Fr*)part, W, WZ,
                       (uint32_t)L, n_chunks, Bp);
  }
  ZK_HIP(hipGetLastError());
  const Fr* scs[2] = {W, WZ};
  if ((rc = commit(ctx, pk, 2, scs, nullptr, commits_w_out, batch))) return rc;
  pk->round = 0;
  return ZKMI_OK;
}

}  // extern "C"

// ---- plonk.Prove in one call: the five rounds with the Fiat-Shamir transcript on the host in C++ --------
// Transcript (the Python twin is plonk.py::challenge / lin_scalars): SHA-256(label ||
// parts) mod r, integers as 32 big-endian bytes, G1 points as x || y big-endian (infinity: 64 zero
// bytes).  gamma = H("gamma", vk digest, public inputs, [a], [b], [c]); beta = H("beta", gamma);
// alpha = H("alpha", beta, [z]); zeta = H("zeta", alpha, [t_lo], [t_mid], [t_hi]);
// v = H("v", zeta, six evaluations).
namespace {
void fr_be(uint8_t out[32], const Fr& plain) {
  for (int i = 0; i < 8; i++) {
    const uint32_t x = plain.v[7 - i];
    out[4 * i] = (uint8_t)(x >> 24);
    out[4 * i + 1] = (uint8_t)(x >> 16);
    out[4 * i + 2] = (uint8_t)(x >> 8);
    out[4 * i + 3] = (uint8_t)x;
  }
}
struct Transcript {
  Sha256 h;
  explicit Transcript(const char* label) { h.update((const uint8_t*)label, strlen(label)); }
  void fr_mont(const Fr& m) {   // a field element given in Montgomery form
    uint8_t b[32];
    fr_be(b, from_mont(m));
    h.update(b, 32);
  }
  void point(const G1Affine& p) {
    uint8_t b[64] = {0};
    if (!(p.x.is_zero() && p.y.is_zero())) {
      const Fq x = from_mont(p.x), y = from_mont(p.y);
      for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++) {
          b[4 * i + k] = (uint8_t)(x.v[7 - i] >> (24 - 8 * k));
          b[32 + 4 * i + k] = (uint8_t)(y.v[7 - i] >> (24 - 8 * k));
        }
    }
    h.update(b, 64);
  }
  Fr challenge() {   // digest as a big-endian integer, reduced mod r, in Montgomery form
    uint8_t d[32];
    h.final(d);
    Fr x = Fr::zero();
    for (int i = 0; i < 32; i++) x.v[7 - i / 4] |= (uint32_t)d[i] << (24 - 8 * (i % 4));
    for (;;) {
      bool ge = true;
      for (int j = 7; j >= 0; j--)
        if (x.v[j] != FrParams::p(j)) {
          ge = x.v[j] > FrParams::p(j);
          break;
        }
      if (!ge) break;
      int64_t br = 0;
      for (int j = 0; j < 8; j++) {
        const int64_t t = (int64_t)x.v[j] - (int64_t)FrParams::p(j) + br;
        x.v[j] = (uint32_t)t;
        br = t >> 32;
      }
    }
    return to_mont(x);
  }
};
Fr fr_pow_u64(Fr base, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = mul(r, base);
    base = sqr(base);
    e >>= 1;
  }
  return r;
}
Fr fr_u64(uint64_t v) {
  Fr x = Fr::zero();
  x.v[0] = (uint32_t)v;
  x.v[1] = (uint32_t)(v >> 32);
  return to_mont(x);
}
}  // namespace

extern "C" {

/* proofs_out: batch x 96 words of 8 bytes = 9 G1 affine points ([a] [b] [c] [z] [t_lo] [t_mid] [t_hi]
 * [W_zeta] [W_zeta_w], Montgomery) then 6 fr (a, b, c, S1, S2 at zeta, z at zeta w; Montgomery). */
int zkmi_plonk_prove(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const zkmi_cs* cs, const void* inputs,
                     size_t batch, const void* blind, const uint8_t* vk_digest /* 32 bytes */,
                     void* proofs_out, int32_t* status_out) {
  if (!ctx || !pk || !cs || !inputs || !blind || !vk_digest || !proofs_out || !status_out || batch == 0)
    return ZKMI_ERR_ARG;
  const size_t n = (size_t)1 << pk->log_n, n_pub = pk->n_public;
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  // public inputs on the host (gnark's image)
  std::vector<Fr> pubs(batch * (n_pub ? n_pub : 1));
  for (size_t p = 0; p < batch && n_pub; p++)
    if (hipMemcpy(pubs.data() + p * n_pub, (const char*)inputs + p * n_in * 32, n_pub * 32,
                  hipMemcpyDefault) != hipSuccess) {
      ctx->err = "plonk_prove: cannot read the public inputs";
      return ZKMI_ERR_HIP;
    }
  std::vector<G1Affine> abc(batch * 3), cz(batch), ct(batch * 3), cw(batch * 2);
  std::vector<Fr> bg(batch * 2), al(batch), zz(batch * 2), ev(batch * 6), sc(batch * 14);
  int rc = zkmi_plonk_round1(ctx, pk, cs, inputs, batch, blind, abc.data(), status_out);
  if (rc) return rc;
  for (size_t p = 0; p < batch; p++) {
    Transcript t("gamma");
    t.h.update(vk_digest, 32);
    for (size_t j = 0; j < n_pub; j++) t.fr_mont(pubs[p * n_pub + j]);
    for (int k = 0; k < 3; k++) t.point(abc[p * 3 + k]);
    const Fr gamma = t.challenge();
    Transcript tb("beta");
    tb.fr_mont(gamma);
    bg[2 * p] = tb.challenge();
    bg[2 * p + 1] = gamma;
  }
  if ((rc = zkmi_plonk_round2(ctx, pk, bg.data(), cz.data()))) return rc;
  for (size_t p = 0; p < batch; p++) {
    Transcript t("alpha");
    t.fr_mont(bg[2 * p]);
    t.point(cz[p]);
    al[p] = t.challenge();
  }
  if ((rc = zkmi_plonk_round3(ctx, pk, al.data(), ct.data()))) return rc;
  // w = 5^((r - 1) / n): root of unity of the domain, from the key's omega table
  Fr w;
  if (hipMemcpy(&w, pk->omega + 1, 32, hipMemcpyDefault) != hipSuccess) return ZKMI_ERR_HIP;
  for (size_t p = 0; p < batch; p++) {
    Transcript t("zeta");
    t.fr_mont(al[p]);
    for (int k = 0; k < 3; k++) t.point(ct[p * 3 + k]);
    zz[2 * p] = t.challenge();
    zz[2 * p + 1] = mul(zz[2 * p], w);
  }
  if ((rc = zkmi_plonk_round4(ctx, pk, zz.data(), ev.data()))) return rc;
  const Fr ninv = inverse(fr_u64(n)), five = fr_u64(5), tf = fr_u64(25);
  // 1 / (zeta - w^j), j = 0 .. max(n_pub, 1) - 1 (j = 0 also serves L_1), for every proof: one
  // inversion in all (Montgomery's trick); zeta = w^j has probability 2^-250
  const size_t nd = n_pub ? n_pub : 1;
  std::vector<Fr> den(batch * nd), pre(batch * nd);
  {
    Fr acc = Fr::one();
    for (size_t p = 0; p < batch; p++) {
      Fr wj = Fr::one();
      for (size_t j = 0; j < nd; j++) {
        den[p * nd + j] = sub(zz[2 * p], wj);
        pre[p * nd + j] = acc;
        if (!den[p * nd + j].is_zero()) acc = mul(acc, den[p * nd + j]);
        wj = mul(wj, w);
      }
    }
    Fr inv = inverse(acc);
    for (size_t i = batch * nd; i-- > 0;) {
      if (den[i].is_zero()) continue;
      const Fr t = mul(inv, pre[i]);
      inv = mul(inv, den[i]);
      den[i] = t;
    }
  }
  for (size_t p = 0; p < batch; p++) {
    const Fr beta = bg[2 * p], gamma = bg[2 * p + 1], alpha = al[p], zeta = zz[2 * p];
    const Fr* e = ev.data() + 6 * p;   // a, b, c, s1, s2, z(zeta w)
    Transcript t("v");
    t.fr_mont(zeta);
    for (int k = 0; k < 6; k++) t.fr_mont(e[k]);
    const Fr v = t.challenge();
    // scalars of the linearisation polynomial (plonk.py::lin_scalars)
    const Fr zh = sub(fr_pow_u64(zeta, n), Fr::one());
    const Fr l1 = mul(mul(zh, ninv), den[p * nd]);
    Fr pi = Fr::zero(), wj = Fr::one();
    for (size_t j = 0; j < n_pub; j++) {   // PI(zeta) = sum_j -x_j L_j(zeta)
      const Fr lj = mul(mul(mul(zh, wj), ninv), den[p * nd + j]);
      pi = sub(pi, mul(pubs[p * n_pub + j], lj));
      wj = mul(wj, w);
    }
    const Fr bz = mul(beta, zeta);
    const Fr a1 = mul(mul(add(add(e[0], bz), gamma), add(add(e[1], mul(five, bz)), gamma)),
                      add(add(e[2], mul(tf, bz)), gamma));
    const Fr a2 = mul(add(add(e[0], mul(beta, e[3])), gamma), add(add(e[1], mul(beta, e[4])), gamma));
    const Fr zn2 = fr_pow_u64(zeta, n + 2), al2 = mul(alpha, alpha);
    Fr* s = sc.data() + 14 * p;
    s[0] = mul(e[0], e[1]);                                        // qm
    s[1] = e[0];                                                   // ql
    s[2] = e[1];                                                   // qr
    s[3] = e[2];                                                   // qo
    s[4] = neg(mul(mul(mul(alpha, a2), beta), e[5]));              // s3
    s[5] = add(mul(alpha, a1), mul(al2, l1));                      // z
    s[6] = neg(zh);                                                // t_lo
    s[7] = neg(mul(zh, zn2));                                      // t_mid
    s[8] = neg(mul(mul(zh, zn2), zn2));                            // t_hi
    Fr c0 = sub(sub(pi, mul(al2, l1)), mul(mul(mul(alpha, a2), add(e[2], gamma)), e[5]));   // r0
    Fr vp = Fr::one();
    for (int k = 0; k < 5; k++) {
      vp = mul(vp, v);
      c0 = sub(c0, mul(vp, e[k]));
    }
    s[9] = c0;
    s[10] = v;
    s[11] = zeta;
    s[12] = zz[2 * p + 1];
    s[13] = e[5];
  }
  if ((rc = zkmi_plonk_round5(ctx, pk, sc.data(), cw.data()))) return rc;
  // assemble the records on the host, one copy out (host or device destination)
  std::vector<uint8_t> rec(batch * 768);
  for (size_t p = 0; p < batch; p++) {
    uint8_t* r = rec.data() + p * 768;
    memcpy(r, &abc[p * 3], 192);
    memcpy(r + 192, &cz[p], 64);
    memcpy(r + 256, &ct[p * 3], 192);
    memcpy(r + 448, &cw[p * 2], 128);
    memcpy(r + 576, &ev[p * 6], 192);
  }
  ZK_HIP(hipMemcpy(proofs_out, rec.data(), rec.size(), hipMemcpyDefault));
  return ZKMI_OK;
}

}  // extern "C"
