// C-ABI of libzkmi.so (include/zkmi.h) and the Groth16 prove pipeline that strings the kernels
// together: solve -> quotient (7 NTTs) -> 4 G1 MSMs + 1 G2 MSM -> assembly.
// Stands in for groth16.Prove in gnark backend/groth16/bn254/prove.go [UPSTREAM-RECALL,
// SURVEY.md §3.2].  There is no CPU fallback anywhere in this file.
#include <algorithm>

#include "zkmi_internal.h"
#include "ff29.h"
#include "ec29.h"

using namespace zk;

namespace zk {

int ensure_scratch(zkmi_ctx* ctx, int slot, size_t bytes, void** out) {
  DevBuf& s = ctx->scratch[slot];
  if (s.bytes < bytes) {
    if (s.p) {
      // nothing queued on any of the context's streams may still use the old buffer
      hipStreamSynchronize(ctx->stream);
      if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
      if (ctx->stream3) hipStreamSynchronize(ctx->stream3);
      hipFree(s.p);
      s.p = nullptr;
      s.bytes = 0;
    }
    hipError_t e = hipMalloc(&s.p, bytes);
    if (e != hipSuccess) {
      ctx->err = "hipMalloc(" + std::to_string(bytes) + " B) failed: " + hipGetErrorString(e) +
                 " (the working set of this batch does not fit beside the MSM tables: load the key"
                 " with zkmi_pk_desc.max_batch >= the batch size, or with a table_budget_bytes cap)";
      s.p = nullptr;
      return ZKMI_ERR_OOM;
    }
    s.bytes = bytes;
  }
  *out = s.p;
  return ZKMI_OK;
}

static bool is_device_ptr(const void* p) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear sticky "invalid value" for plain host memory
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// RAII staging of a caller buffer (host or device) as device memory
struct Staged {
  zkmi_ctx* ctx;
  void* dev = nullptr;
  void* host = nullptr;  // non-null when a copy back is pending
  size_t bytes = 0;
  bool owned = false;
  Staged(zkmi_ctx* c) : ctx(c) {}
  int in(const void* p, size_t n) {
    bytes = n;
    if (n == 0) return ZKMI_OK;
    if (is_device_ptr(p)) {
      dev = const_cast<void*>(p);
      return ZKMI_OK;
    }
    ZK_HIP(hipMalloc(&dev, n));
    owned = true;
    ZK_HIP(hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, ctx->stream));
    return ZKMI_OK;
  }
  int out(void* p, size_t n) {
    bytes = n;
    if (n == 0) return ZKMI_OK;
    if (is_device_ptr(p)) {
      dev = p;
      return ZKMI_OK;
    }
    ZK_HIP(hipMalloc(&dev, n));
    owned = true;
    host = p;
    return ZKMI_OK;
  }
  int finish() {
    if (host && dev) {
      ZK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
      ZK_HIP(hipStreamSynchronize(ctx->stream));
      host = nullptr;
    }
    return ZKMI_OK;
  }
  ~Staged() {
    if (owned && dev) {
      hipStreamSynchronize(ctx->stream);
      hipFree(dev);
    }
  }
};

// ---- elementwise field kernels --------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void field_mul_kernel(const Fp<P>* a, const Fp<P>* b, Fp<P>* r,
                                                        size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) r[i] = mul(a[i], b[i]);
}

// `iters` dependent products per lane, two independent chains to expose ILP
template <class P>
__global__ __launch_bounds__(256) void field_mul_bench_kernel(Fp<P>* io, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fp<P> x = io[i], y = io[i];
  y.v[0] ^= 1;
  for (int k = 0; k < iters; k++) {
    x = mul(x, y);
    y = mul(y, x);
  }
  io[i] = add(x, y);
}

// the same measurement for the 9 x 29-bit representation used inside the G1 MSM (ff29.h)
__global__ __launch_bounds__(256) void field_mul_bench_f29_kernel(Fq* io, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  Fq29 x = unpack29<Fq29Params>(io[i].v), y = x;
  y.v[0] ^= 1;
  for (int k = 0; k < iters; k++) {
    x = mul(x, y);
    y = mul(y, x);
  }
  pack_canonical<Fq29Params>(io[i].v, norm(add(x, y)));
}

// ---- proof assembly ---------------------------------------------------------------------------------
// Ar = sumA + alpha + r*delta ; Bs1 = sumB1 + beta + s*delta ; Bs = sumB2 + beta2 + s*delta2 ;
// Krs = sumK + sumZ - rs*delta + s*Ar + r*Bs1.
// The multiples of the fixed points delta / delta2 come from one-base window tables through the
// MSM kernel itself (26-32 mixed additions instead of a 254-step double-and-add); only
// s*Ar + r*Bs1 has per-proof bases and is done here with a joint double-and-add.
struct PkConsts {
  G1Affine alpha, beta1;
  G2Affine beta2;
};
struct ProofOut {
  G1Affine ar, krs;
  G2Affine bs;
};

// rs rows: [0] r, [1] s (inputs) -> [2] = -r*s
__global__ __launch_bounds__(64) void rs_prep_kernel(Fr* rs, size_t Bp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp) return;
  bi_st(rs, 2, i, Bp, neg(mul(bi_ld(rs, 0, i, Bp), bi_ld(rs, 1, i, Bp))));
}

// 2 * acc on the 29-bit lazy representation (dbl-2008-s-1; same bounds as ec29.h's mdbl29 with the
// accumulator's own x (|x| < 5p) and y (|y| < 2p) in the place of the affine point: every product
// stays below 25 p^2).  G1 has odd order: y != 0 for every finite point.
__device__ __forceinline__ void dbl29(G1Acc29& acc) {
  if (acc.inf) return;
  const Fq29 u = norm(add(acc.y, acc.y));
  const Fq29 v = msqr(u);
  const Fq29 w = mmul(u, v);
  const Fq29 s = mmul(acc.x, v);
  const Fq29 x2 = msqr(acc.x);
  const Fq29 m = norm(add(add(x2, x2), x2));
  const Fq29 x3 = norm(sub(msqr(m), add(s, s)));       // (-3.5p, 2.5p)
  const Fq29 y3 = norm(sub(mmul(m, sub(s, x3)), mmul(w, acc.y)));   // (-2p, 2p)
  acc.zz = mmul(v, acc.zz);
  acc.zzz = mmul(w, acc.zzz);
  acc.x = x3;
  acc.y = y3;
}

__global__ __launch_bounds__(64) void assemble_kernel(const G1XYZZ* sA, const G1XYZZ* sB1,
                                                      const G1XYZZ* sK, const G1XYZZ* sZ,
                                                      const G1XYZZ* tR, const G1XYZZ* tS,
                                                      const G1XYZZ* tNRS, const Fr* rs, size_t Bp,
                                                      PkConsts pk, ProofOut* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp) return;
  const Fr r = from_mont(bi_ld(rs, 0, i, Bp)), s = from_mont(bi_ld(rs, 1, i, Bp));
  G1XYZZ AR = sA[i];
  madd(AR, pk.alpha);
  padd(AR, tR[i]);
  const G1Affine ar = to_affine(AR);
  G1XYZZ BS1 = sB1[i];
  madd(BS1, pk.beta1);
  padd(BS1, tS[i]);
  const G1Affine bs1 = to_affine(BS1);
  // s*ar + r*bs1, one shared doubling chain, on the 29-bit representation of the MSM inner loop
  // (a 16-wavefront latency chain: what counts is the instruction count per bit, 19 products of
  // ~230 instructions instead of 17 of ~550 on ff.h)
  const bool ar_fin = !ar.is_inf(), bs_fin = !bs1.is_inf();
  const Fq29 ax = from_std<Fq29Params>(ar.x), ay = from_std<Fq29Params>(ar.y);
  const Fq29 bx = from_std<Fq29Params>(bs1.x), by = from_std<Fq29Params>(bs1.y);
  G1Acc29 acc29 = G1Acc29::infinity();
  for (int w = 7; w >= 0; w--) {
    const uint32_t sw = s.v[w], rw = r.v[w];
#pragma unroll 1
    for (int b = 31; b >= 0; b--) {
      dbl29(acc29);
      if (((sw >> b) & 1u) && ar_fin) madd29(acc29, ax, ay);
      if (((rw >> b) & 1u) && bs_fin) madd29(acc29, bx, by);
    }
  }
  G1XYZZ acc = to_std(acc29);
  padd(acc, sK[i]);
  padd(acc, sZ[i]);
  padd(acc, tNRS[i]);
  out[i].ar = ar;
  out[i].krs = to_affine(acc);
}

__global__ __launch_bounds__(64) void assemble_g2_kernel(const G2XYZZ* sB2, const G2XYZZ* tS2,
                                                         size_t Bp, G2Affine beta2, ProofOut* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp) return;
  G2XYZZ BS = sB2[i];
  madd(BS, beta2);
  padd(BS, tS2[i]);
  out[i].bs = to_affine(BS);
}

}  // namespace zk

// ===================================================================================================
extern "C" {

int zkmi_init(int device, zkmi_ctx** out) {
  if (!out) return ZKMI_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZKMI_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return ZKMI_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return ZKMI_ERR_HIP;
  auto* ctx = new zkmi_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ZKMI_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) {
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return ZKMI_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) {
    hipStreamDestroy(ctx->stream2);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return ZKMI_ERR_HIP;
  }
  for (auto& st : ctx->sets) {
    hipEventCreate(&st.ev0);
    hipEventCreate(&st.ev1);
    for (auto& e : st.evq) hipEventCreate(&e);
    for (auto& e : st.eva) hipEventCreate(&e);
    for (auto& p : st.msm_ev) {
      hipEventCreate(&p[0]);
      hipEventCreate(&p[1]);
    }
  }
  for (auto& e : ctx->ev) hipEventCreate(&e);
  for (auto& e : ctx->part_ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& e : ctx->acc_ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  ctx->plans.reserve(32);
  *out = ctx;
  return ZKMI_OK;
}

void zkmi_destroy(zkmi_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  hipStreamSynchronize(ctx->stream2);
  hipStreamSynchronize(ctx->stream3);
  for (auto& st : ctx->sets) {
    if (st.ev0) hipEventDestroy(st.ev0);
    if (st.ev1) hipEventDestroy(st.ev1);
    for (auto& e : st.evq)
      if (e) hipEventDestroy(e);
    for (auto& e : st.eva)
      if (e) hipEventDestroy(e);
    for (auto& p : st.msm_ev) {
      if (p[0]) hipEventDestroy(p[0]);
      if (p[1]) hipEventDestroy(p[1]);
    }
  }
  hipStreamDestroy(ctx->stream2);
  hipStreamDestroy(ctx->stream3);
  for (auto& p : ctx->plans) {
    hipFree(p.tw_fwd);
    hipFree(p.tw_inv);
    hipFree(p.coset_fwd);
    hipFree(p.coset_inv);
    hipFree(p.tw29_fwd);
    hipFree(p.tw29_inv);
    hipFree(p.coset29_fwd);
    hipFree(p.coset29n_fwd);
  }
  for (auto& s : ctx->scratch)
    if (s.p) hipFree(s.p);
  witness_ring_free(ctx);
  for (auto& e : ctx->ev)
    if (e) hipEventDestroy(e);
  for (auto& e : ctx->part_ev)
    if (e) hipEventDestroy(e);
  for (auto& e : ctx->acc_ev)
    if (e) hipEventDestroy(e);
  hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* zkmi_last_error(zkmi_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int zkmi_sync(zkmi_ctx* ctx) {
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

void* zkmi_stream(zkmi_ctx* ctx) { return (void*)ctx->stream; }

int zkmi_field_mul(zkmi_ctx* ctx, int which, const void* a, const void* b, void* r, size_t n) {
  ZK_HIP(hipSetDevice(ctx->device));
  Staged sa(ctx), sb(ctx), sr(ctx);
  int rc;
  if ((rc = sa.in(a, n * 32)) || (rc = sb.in(b, n * 32)) || (rc = sr.out(r, n * 32))) return rc;
  if (n == 0) return ZKMI_OK;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 8192) g = 8192;
  if (which == 0)
    hipLaunchKernelGGL((field_mul_kernel<FrParams>), dim3(g), dim3(256), 0, ctx->stream,
                       (const Fr*)sa.dev, (const Fr*)sb.dev, (Fr*)sr.dev, n);
  else
    hipLaunchKernelGGL((field_mul_kernel<FqParams>), dim3(g), dim3(256), 0, ctx->stream,
                       (const Fq*)sa.dev, (const Fq*)sb.dev, (Fq*)sr.dev, n);
  ZK_HIP(hipGetLastError());
  if ((rc = sr.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int zkmi_field_mul_bench(zkmi_ctx* ctx, int which, size_t n_threads, int iters, double* rate) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  n_threads = round_up(n_threads, 256);
  void* buf;
  int rc = ensure_scratch(ctx, 5, n_threads * 32, &buf);
  if (rc) return rc;
  // every byte 0x11: top limb 0x11111111 < 0x30644e72, so the element is < p
  ZK_HIP(hipMemsetAsync(buf, 0x11, n_threads * 32, ctx->stream));
  auto launch = [&]() {
    if (which == 0)
      hipLaunchKernelGGL((field_mul_bench_kernel<FrParams>), dim3((unsigned)(n_threads / 256)),
                         dim3(256), 0, ctx->stream, (Fr*)buf, iters);
    else if (which == 2)
      hipLaunchKernelGGL(field_mul_bench_f29_kernel, dim3((unsigned)(n_threads / 256)), dim3(256),
                         0, ctx->stream, (Fq*)buf, iters);
    else
      hipLaunchKernelGGL((field_mul_bench_kernel<FqParams>), dim3((unsigned)(n_threads / 256)),
                         dim3(256), 0, ctx->stream, (Fq*)buf, iters);
  };
  launch();  // warm-up
  ZK_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
  launch();
  ZK_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
  ZK_HIP(hipEventSynchronize(ctx->ev[7]));
  float ms = 0;
  ZK_HIP(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]));
  *rate = (double)n_threads * iters * 2.0 / (ms * 1e-3);
  return ZKMI_OK;
}

int zkmi_ntt_batch(zkmi_ctx* ctx, void* data, int log_n, size_t batch, int inverse, int coset) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (batch == 0) return ZKMI_OK;
  NttPlan* plan;
  int rc = get_plan(ctx, log_n, &plan);
  if (rc) return rc;
  const size_t n = (size_t)1 << log_n, Bp = round_up(batch, 64);
  Staged sd(ctx), so(ctx);
  if ((rc = sd.in(data, batch * n * 32)) || (rc = so.out(data, batch * n * 32))) return rc;
  void *t0, *t1;
  if ((rc = ensure_scratch(ctx, 1, n * Bp * 32, &t0)) ||
      (rc = ensure_scratch(ctx, 2, n * Bp * 32, &t1)))
    return rc;
  if ((rc = transpose_in(ctx, sd.dev, t0, n, batch, Bp, 32))) return rc;
  if ((rc = ntt_bi(ctx, plan, (const Fr*)t0, (Fr*)t1, Bp, inverse != 0, coset != 0, n))) return rc;
  if ((rc = transpose_out(ctx, t1, so.dev, n, batch, Bp, 32))) return rc;
  if ((rc = so.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int zkmi_h_batch(zkmi_ctx* ctx, const void* a, const void* b, const void* c, void* h_out,
                 int log_n, size_t batch) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (batch == 0) return ZKMI_OK;
  NttPlan* plan;
  int rc = get_plan(ctx, log_n, &plan);
  if (rc) return rc;
  const size_t n = (size_t)1 << log_n, Bp = round_up(batch, 64);
  Staged sa(ctx), sb(ctx), sc(ctx), sh(ctx);
  if ((rc = sa.in(a, batch * n * 32)) || (rc = sb.in(b, batch * n * 32)) ||
      (rc = sc.in(c, batch * n * 32)) || (rc = sh.out(h_out, batch * n * 32)))
    return rc;
  void* t[4];
  for (int i = 0; i < 4; i++)
    if ((rc = ensure_scratch(ctx, 1 + i, n * Bp * 32, &t[i]))) return rc;
  if ((rc = transpose_in(ctx, sa.dev, t[0], n, batch, Bp, 32)) ||
      (rc = transpose_in(ctx, sb.dev, t[1], n, batch, Bp, 32)) ||
      (rc = transpose_in(ctx, sc.dev, t[2], n, batch, Bp, 32)))
    return rc;
  Fr* h;
  if ((rc = compute_h_bi(ctx, plan, (Fr*)t[0], (Fr*)t[1], (Fr*)t[2], (Fr*)t[3], Bp, n, &h)))
    return rc;
  if ((rc = transpose_out(ctx, h, sh.dev, n, batch, Bp, 32))) return rc;
  if ((rc = sh.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

// window_bits: 0 = as many bits per window as the table budget allows (windows of mixed width
// summing to exactly 255 bits); 2..16 = uniform windows of that width.  If the table cannot be
// allocated the plan is relaxed (more, narrower windows) until it fits.
static int bases_load_plan(zkmi_ctx* ctx, int group, const void* bases_dev, size_t n, WinPlan plan,
                           bool may_relax, zkmi_msm_bases** out) {
  for (;;) {
    int rc = msm_bases_build(ctx, group, bases_dev, n, plan, out);
    if (rc != ZKMI_ERR_OOM || !may_relax) return rc;
    if (plan.comb) {
      if (plan.comb <= 8) return rc;
      plan = plan_comb(plan.comb - 1, plan.comb_signed != 0);
    } else if (plan.shared) {
      if (plan.bits[0] <= 4) return rc;
      plan = plan_shared(plan.bits[0] - 1);
    } else {
      if (plan.W >= 64) return rc;
      plan = plan_with_windows(plan.W + 1);
    }
  }
}

// explicit window_bits of the C-ABI: 2..16 = per-window tables, uniform width; 100 + c = one
// shared table of c-bit signed digits per base (c in 4..16)
// 200 + k = comb tables over groups of k bases (k in 2..20); 300 + k = sign-pattern comb tables
// (k in 3..21: 2^(k-1) entries per group, the auto plan's choice)
static bool window_bits_ok(int wb) {
  return wb == 0 || (wb >= 2 && wb <= 16) || (wb >= 104 && wb <= 116) ||
         (wb >= 202 && wb <= 220) || (wb >= 303 && wb <= 321);
}
static WinPlan plan_explicit(int wb) {
  return wb >= 300   ? plan_comb(wb - 300, true)
         : wb >= 200 ? plan_comb(wb - 200, false)
         : wb >= 100 ? plan_shared(wb - 100)
                     : plan_uniform(wb);
}
static const int COMB_WINDOWS = 254;
// HBM the prover needs beside the tables to prove batches of up to `max_batch` with a key of this
// shape: two pipeline sets of {value file, a, b, c, staged inputs}, the NTT scratch,
// the MSM integer scalars + digits + two partial-sum buffers, the per-set sums, the table-build
// scratch that stays allocated, and a margin for the allocator.
static double prove_working_set_bytes(uint32_t log_n, size_t n_slots, size_t n_in, size_t max_batch,
                                      size_t n_msm_max) {
  const double Bp = (double)round_up(max_batch ? max_batch : 1024, 64);
  const double n = (double)((size_t)1 << log_n);
  const double set = (double)n_slots * Bp * 32 + 3 * n * Bp * 32 + Bp * 100 + Bp * (n_in * 32 + 64);
  const double digits = 254.0 * (double)((n_msm_max + 15) / 16) * Bp * 4;   // comb, k >= 16
  const double sint = (double)n_msm_max * Bp * 32;
  const double partials = 2 * 21.0 * 254 * Bp * 256;
  const double sums = 2 * Bp * (7 * 128 + 2 * 256 + 256 + 256.0 * (4 * 128 + 256));
  // device half of the witness entry's staging ring (witness.hip): three chunks of ~64 MB, or of
  // one 64-proof column block of the wire matrix when that is larger (n_in = n_wires here)
  const double ring = 3.0 * std::max(68e6, (double)n_in * 32 * 64);
  return 2 * set + n * Bp * 32 + digits + sint + partials + sums + ring + 2e9 + 1e9;
}
static double free_hbm_bytes() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)64 << 30;
  return (double)free_b;
}
// standalone base sets (zkmi_msm_bases_load): working set of a 1024-wide zkmi_msm_batch
static double usable_table_bytes_standalone(size_t n) {
  const double ws = (double)n * 1024 * 32 * 2 + 254.0 * ((n + 15) / 16) * 1024 * 4 +
                    21.0 * 254 * 1024 * 256 + 3e9;
  const double usable = free_hbm_bytes() - ws;
  return usable > 0 ? usable : 0.0;
}

int zkmi_msm_bases_load(zkmi_ctx* ctx, int group, const void* bases, size_t n, int window_bits,
                        zkmi_msm_bases** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!out) return ZKMI_ERR_ARG;
  if (!window_bits_ok(window_bits)) {
    ctx->err = "window_bits must be 0 (auto), in [2,16], 100 + [4,16], 200 + [2,20] or 300 + [3,21]";
    return ZKMI_ERR_ARG;
  }
  Staged sb(ctx);
  int rc = sb.in(bases, n * (group == 1 ? 64 : 128));
  if (rc) return rc;
  WinPlan plan;
  if (window_bits) {
    plan = plan_explicit(window_bits);
  } else {
    // per-window tables need no Horner tail: keep them when they reach as few windows as the
    // shared table would (small base sets); otherwise one shared table per base, or comb tables
    const double usable = usable_table_bytes_standalone(n);
    plan = plan_windows_for_budget(n, group, (group == 1 ? 0.64 : 0.34) * usable);
    int c1, c2;
    plan_shared_for_budget(group == 1 ? n : 0, group == 2 ? n : 0, 0.9 * usable, &c1, &c2);
    const WinPlan ps = plan_shared(group == 1 ? c1 : c2);
    if (ps.W < plan.W) plan = ps;
    if (n >= 64) {
      int k1, k2;
      bool s1, s2;
      plan_comb_for_budget(group == 1 ? n : 0, group == 2 ? n : 0, 0.9 * usable, &k1, &k2, &s1, &s2);
      const int k = group == 1 ? k1 : k2;
      if ((double)COMB_WINDOWS / k < (double)plan.W) plan = plan_comb(k, group == 1 ? s1 : s2);
    }
  }
  return bases_load_plan(ctx, group, sb.dev, n, plan, window_bits == 0, out);
}

void zkmi_msm_bases_free(zkmi_ctx* ctx, zkmi_msm_bases* b) {
  if (!b) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
  }
  if (b->table) hipFree(b->table);
  if (b->inf) hipFree(b->inf);
  delete b;
}

int zkmi_msm_batch(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const void* scalars, size_t batch,
                   void* out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (batch == 0) return ZKMI_OK;
  const size_t n = bases->n, Bp = round_up(batch, 64);
  const size_t pt = bases->group == 1 ? 64 : 128;
  Staged ss(ctx), so(ctx);
  int rc;
  if ((rc = ss.in(scalars, batch * n * 32)) || (rc = so.out(out, batch * pt))) return rc;
  void *sbi, *acc, *aff;
  if ((rc = ensure_scratch(ctx, 1, (n ? n : 1) * Bp * 32, &sbi)) ||
      (rc = ensure_scratch(ctx, 2, Bp * pt * 2, &acc)) ||
      (rc = ensure_scratch(ctx, 3, Bp * pt, &aff)))
    return rc;
  if ((rc = transpose_in(ctx, ss.dev, sbi, n, batch, Bp, 32))) return rc;
  if ((rc = msm_run(ctx, bases, (const Fr*)sbi, nullptr, Bp, acc))) return rc;
  if ((rc = xyzz_to_affine(ctx, bases->group, acc, aff, Bp))) return rc;
  ZK_HIP(hipMemcpyAsync(so.dev, aff, batch * pt, hipMemcpyDeviceToDevice, ctx->stream));
  if ((rc = so.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

int zkmi_fixed_base_mul(zkmi_ctx* ctx, int group, const void* base, const void* scalars, size_t n,
                        void* out) {
  // an MSM over ONE base with the scalars playing the role of the batch
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (n == 0) return ZKMI_OK;
  zkmi_msm_bases* b = nullptr;
  int rc = zkmi_msm_bases_load(ctx, group, base, 1, 8, &b);
  if (rc) return rc;
  const size_t Bp = round_up(n, 64);
  const size_t pt = group == 1 ? 64 : 128;
  Staged ss(ctx), so(ctx);
  void *sbi = nullptr, *acc = nullptr, *aff = nullptr;
  if ((rc = ss.in(scalars, n * 32)) || (rc = so.out(out, n * pt)) ||
      (rc = ensure_scratch(ctx, 1, Bp * 32, &sbi)) ||
      (rc = ensure_scratch(ctx, 2, Bp * pt * 2, &acc)) ||
      (rc = ensure_scratch(ctx, 3, Bp * pt, &aff))) {
    zkmi_msm_bases_free(ctx, b);
    return rc;
  }
  rc = transpose_in(ctx, ss.dev, sbi, 1, n, Bp, 32);   // one row of n scalars, batch-inner
  if (!rc) rc = msm_run(ctx, b, (const Fr*)sbi, nullptr, Bp, acc);
  if (!rc) rc = xyzz_to_affine(ctx, group, acc, aff, Bp);
  if (!rc) {
    hipMemcpyAsync(so.dev, aff, n * pt, hipMemcpyDeviceToDevice, ctx->stream);
    rc = so.finish();
  }
  hipStreamSynchronize(ctx->stream);
  zkmi_msm_bases_free(ctx, b);
  return rc;
}

// ---- proving key / constraint system -------------------------------------------------------------
static int upload_u32(zkmi_ctx* ctx, const uint32_t* src, size_t n, uint32_t** out) {
  *out = nullptr;
  if (n == 0) return ZKMI_OK;
  ZK_HIP(hipMalloc((void**)out, n * 4));
  ZK_HIP(hipMemcpy(*out, src, n * 4, hipMemcpyDefault));
  return ZKMI_OK;
}

void zkmi_pk_free(zkmi_ctx* ctx, zkmi_pk* pk) {
  if (!pk) return;
  if (ctx) hipSetDevice(ctx->device);
  zkmi_msm_bases_free(ctx, pk->A);
  zkmi_msm_bases_free(ctx, pk->B1);
  zkmi_msm_bases_free(ctx, pk->K);
  zkmi_msm_bases_free(ctx, pk->Z);
  zkmi_msm_bases_free(ctx, pk->B2);
  zkmi_msm_bases_free(ctx, pk->D1);
  zkmi_msm_bases_free(ctx, pk->D2);
  commit_keys_free(ctx, pk);
  if (pk->idx3) hipFree(pk->idx3);
  if (pk->a_wire) hipFree(pk->a_wire);
  if (pk->b_wire) hipFree(pk->b_wire);
  if (pk->k_wire) hipFree(pk->k_wire);
  delete pk;
}

// wire index of every retained base from a gnark-style infinity map ([]bool, one byte per wire)
static int wires_from_infinity(zkmi_ctx* ctx, const uint8_t* inf, uint32_t n_wires, uint32_t want,
                               const char* name, std::vector<uint32_t>* out) {
  std::vector<uint8_t> host(n_wires);
  if (n_wires && hipMemcpy(host.data(), inf, n_wires, hipMemcpyDefault) != hipSuccess) {
    ctx->err = std::string("pk: cannot read ") + name;
    return ZKMI_ERR_HIP;
  }
  out->clear();
  for (uint32_t i = 0; i < n_wires; i++)
    if (!host[i]) out->push_back(i);
  if (out->size() != want) {
    ctx->err = std::string("pk: ") + name + " retains " + std::to_string(out->size()) +
               " wires but the key holds " + std::to_string(want) + " points";
    return ZKMI_ERR_ARG;
  }
  return ZKMI_OK;
}

int zkmi_pk_load(zkmi_ctx* ctx, const zkmi_pk_desc* d, zkmi_pk** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!d || !out) return ZKMI_ERR_ARG;
  if (d->log_n < 1 || d->log_n > 28) {
    ctx->err = "pk: log_n out of range [1,28]";
    return ZKMI_ERR_ARG;
  }
  // gnark allocates pk.G1.Z with Domain.Cardinality entries and uses the first n - 1
  // [UPSTREAM-RECALL]: longer arrays are accepted and truncated
  if (d->n_z + 1 < (1u << d->log_n)) {
    ctx->err = "pk: n_z must be at least 2^log_n - 1";
    return ZKMI_ERR_ARG;
  }
  const uint32_t n_z = (1u << d->log_n) - 1;
  // wire index of every base: explicit arrays, or gnark's InfinityA / InfinityB / nbPublic
  std::vector<uint32_t> wa, wb, wk;
  {
    int rc;
    struct { const uint32_t* p; uint32_t n; const char* name; std::vector<uint32_t>* v; } idx[3] = {
        {d->a_wire, d->n_a, "a_wire", &wa}, {d->b_wire, d->n_b, "b_wire", &wb},
        {d->k_wire, d->n_k, "k_wire", &wk}};
    for (auto& t : idx) {
      if (!t.p) continue;
      t.v->resize(t.n);
      if (t.n && hipMemcpy(t.v->data(), t.p, (size_t)t.n * 4, hipMemcpyDefault) != hipSuccess) {
        ctx->err = std::string("pk: cannot read ") + t.name;
        return ZKMI_ERR_HIP;
      }
    }
    if (!d->a_wire) {
      if (!d->infinity_a) {
        ctx->err = "pk: neither a_wire nor infinity_a given";
        return ZKMI_ERR_ARG;
      }
      if ((rc = wires_from_infinity(ctx, d->infinity_a, d->n_wires, d->n_a, "infinity_a", &wa)))
        return rc;
    }
    if (!d->b_wire) {
      if (!d->infinity_b) {
        ctx->err = "pk: neither b_wire nor infinity_b given";
        return ZKMI_ERR_ARG;
      }
      if ((rc = wires_from_infinity(ctx, d->infinity_b, d->n_wires, d->n_b, "infinity_b", &wb)))
        return rc;
    }
    if (!d->k_wire) {
      if (d->n_public < 1 || d->n_public > d->n_wires ||
          (d->n_commitments == 0 && d->n_wires - d->n_public != d->n_k)) {
        ctx->err = "pk: without k_wire, n_public must satisfy n_wires - n_public == n_k";
        return ZKMI_ERR_ARG;
      }
      // gnark's setup leaves the private committed wires and the commitment wires out of G1.K
      std::vector<uint8_t> taken(d->n_wires, 0);
      for (uint32_t i = 0; i < d->n_commitments && d->commitments; i++) {
        const zkmi_commitment_desc& c = d->commitments[i];
        std::vector<uint32_t> pw(c.n_private);
        if (c.n_private && (!c.private_wires || hipMemcpy(pw.data(), c.private_wires, (size_t)c.n_private * 4,
                                                           hipMemcpyDefault) != hipSuccess)) {
          ctx->err = "pk: cannot read the private wires of commitment " + std::to_string(i);
          return ZKMI_ERR_ARG;
        }
        for (uint32_t w : pw)
          if (w < d->n_wires) taken[w] = 1;
        if (c.commitment_wire < d->n_wires) taken[c.commitment_wire] = 1;
      }
      wk.clear();
      for (uint32_t w = d->n_public; w < d->n_wires; w++)
        if (!taken[w]) wk.push_back(w);
      if (wk.size() != d->n_k) {
        ctx->err = "pk: n_k does not match n_wires - n_public - committed wires";
        return ZKMI_ERR_ARG;
      }
    }
    // every wire index is range-checked on the host before any kernel can use it as a row number
    for (auto& t : idx)
      for (uint32_t w : *t.v)
        if (w >= d->n_wires) {
          ctx->err = std::string("pk: ") + t.name + " holds a wire index >= n_wires";
          return ZKMI_ERR_ARG;
        }
  }
  if (!window_bits_ok((int)d->window_bits_g1) || !window_bits_ok((int)d->window_bits_g2)) {
    ctx->err = "pk: window_bits must be 0 (auto), in [2,16], 100 + [4,16], 200 + [2,20] or 300 + [3,21]";
    return ZKMI_ERR_ARG;
  }
  if (d->msm_chunk_factor > 64) {
    ctx->err = "pk: msm_chunk_factor must be in [0,64]";
    return ZKMI_ERR_ARG;
  }
  auto* pk = new zkmi_pk();
  pk->log_n = d->log_n;
  pk->n_wires = d->n_wires;
  pk->n_a = d->n_a;
  pk->n_b = d->n_b;
  pk->n_k = d->n_k;
  pk->n_z = n_z;
  pk->max_batch = d->max_batch ? d->max_batch : 1024;
  int rc;
  // commitment keys first (per-window tables of the Pedersen bases, 2 x 33 KB per committed wire at
  // the widest setting): the plan below is sized against the HBM that is free AFTER them
  if ((rc = commit_keys_load(ctx, d, pk))) {
    zkmi_pk_free(ctx, pk);
    return rc;
  }
  // One window plan per group for the whole key.  Auto plans are sized against the free HBM minus
  // the working set of the largest batch the caller will prove (and the caller's own cap).
  const bool auto1 = d->window_bits_g1 == 0, auto2 = d->window_bits_g2 == 0;
  const size_t n1 = (size_t)d->n_a + d->n_b + d->n_k + n_z;
  const size_t n_msm_max = std::max(std::max((size_t)d->n_a, (size_t)d->n_b),
                                    std::max((size_t)d->n_k, (size_t)n_z));
  const size_t n_slots = d->n_slots_hint ? d->n_slots_hint : (size_t)d->n_wires + d->n_wires / 20;
  const double ws = prove_working_set_bytes(d->log_n, n_slots, d->n_wires, pk->max_batch, n_msm_max);
  double usable = free_hbm_bytes() - ws;
  if (usable < 0) usable = 0;
  if (d->table_budget_bytes && (double)d->table_budget_bytes < usable)
    usable = (double)d->table_budget_bytes;
  // auto: per-window tables when they reach as few windows as a shared table would (small keys:
  // no Horner tail), otherwise one shared table per base
  int sc1 = 0, sc2 = 0;
  plan_shared_for_budget(n1, d->n_b, 0.98 * usable, &sc1, &sc2);
  WinPlan p1 = auto1 ? plan_windows_for_budget(n1, 1, 0.64 * usable)
                     : plan_explicit((int)d->window_bits_g1);
  WinPlan p2 = auto2 ? plan_windows_for_budget(d->n_b, 2, 0.34 * usable)
                     : plan_explicit((int)d->window_bits_g2);
  if (auto1 && sc1 && plan_shared(sc1).W < p1.W) p1 = plan_shared(sc1);
  if (auto2 && sc2 && plan_shared(sc2).W < p2.W) p2 = plan_shared(sc2);
  // comb tables (joint tables over k bases) when they need fewer additions per base still; small
  // keys keep the layouts above (their MSMs are latency, not throughput)
  if (n1 >= 4096) {
    int k1 = 0, k2 = 0;
    bool s1 = true, s2 = true;
    plan_comb_for_budget(auto1 ? n1 : 0, auto2 ? d->n_b : 0, 0.98 * usable, &k1, &k2, &s1, &s2,
                         d->sparse_witness == 0);
    if (auto1 && (double)COMB_WINDOWS / k1 < (double)p1.W) p1 = plan_comb(k1, s1);
    if (auto2 && (double)COMB_WINDOWS / k2 < (double)p2.W) p2 = plan_comb(k2, s2);
  }
  // sparse_witness = 2: (almost) every wire is a bit.  The wire MSMs then execute one addition per
  // group (window 0) whatever the group size, so they get small subset-sum tables and the dense
  // quotient MSM (h . Z) gets the HBM: the widest table that fits, sign patterns allowed.
  WinPlan p1z = p1;
  if (d->sparse_witness >= 2 && auto1 && auto2 && n1 >= 4096 && n_z >= 4096) {
    const int kw = 12;
    const double wire_bytes =
        (double)((d->n_a + kw - 1) / kw + (d->n_b + kw - 1) / kw + (d->n_k + kw - 1) / kw) *
            (double)(1u << kw) * 64.0 +
        (double)((d->n_b + kw - 1) / kw) * (double)(1u << kw) * 128.0;
    if (wire_bytes < 0.5 * usable) {
      int kz = 0, k2 = 0;
      bool sz = true, s2 = true;
      plan_comb_for_budget(n_z, 0, 0.98 * usable - wire_bytes, &kz, &k2, &sz, &s2, true);
      p1 = plan_comb(kw, false);
      p2 = plan_comb(kw, false);
      p1z = plan_comb(kz, sz);
    }
  }
  auto load = [&](int group, const void* pts, size_t n, const WinPlan& plan, bool relax,
                  zkmi_msm_bases** out) -> int {
    Staged sb(ctx);
    int r = sb.in(pts, n * (group == 1 ? 64 : 128));
    if (r) return r;
    r = bases_load_plan(ctx, group, sb.dev, n, plan, relax, out);
    if (!r) (*out)->chunk_factor = d->msm_chunk_factor;
    return r;
  };
  if ((rc = upload_u32(ctx, wa.data(), d->n_a, &pk->a_wire)) ||
      (rc = upload_u32(ctx, wb.data(), d->n_b, &pk->b_wire)) ||
      (rc = upload_u32(ctx, wk.data(), d->n_k, &pk->k_wire)) ||
      (rc = load(1, d->g1_a, d->n_a, p1, auto1, &pk->A)) ||
      (rc = load(1, d->g1_b, d->n_b, p1, auto1, &pk->B1)) ||
      (rc = load(1, d->g1_k, d->n_k, p1, auto1, &pk->K)) ||
      (rc = load(1, d->g1_z, n_z, p1z, auto1, &pk->Z)) ||
      (rc = load(2, d->g2_b, d->n_b, p2, auto2, &pk->B2))) {
    zkmi_pk_free(ctx, pk);
    return rc;
  }
  {
    const uint32_t idx[3] = {0, 1, 2};
    if ((rc = upload_u32(ctx, idx, 3, &pk->idx3)) ||
        (rc = zkmi_msm_bases_load(ctx, 1, d->g1_delta, 1, 8, &pk->D1)) ||
        (rc = zkmi_msm_bases_load(ctx, 2, d->g2_delta, 1, 8, &pk->D2))) {
      zkmi_pk_free(ctx, pk);
      return rc;
    }
    pk->D1->side = pk->D2->side = 2;   // run on the assembly stream (enqueue_heavy)
  }
  hipMemcpy(&pk->alpha, d->g1_alpha, 64, hipMemcpyDefault);
  hipMemcpy(&pk->beta1, d->g1_beta, 64, hipMemcpyDefault);
  hipMemcpy(&pk->delta1, d->g1_delta, 64, hipMemcpyDefault);
  hipMemcpy(&pk->beta2, d->g2_beta, 128, hipMemcpyDefault);
  hipMemcpy(&pk->delta2, d->g2_delta, 128, hipMemcpyDefault);
  *out = pk;
  return ZKMI_OK;
}

int zkmi_pk_info(const zkmi_pk* pk, uint64_t* info) {
  if (!pk || !info) return ZKMI_ERR_ARG;
  info[0] = (uint64_t)pk->Z->plan.W;
  info[1] = pk->Z->plan.per_base;
  info[2] = (uint64_t)pk->B2->plan.W;
  info[3] = pk->B2->plan.per_base;
  info[4] = pk->A->table_bytes + pk->B1->table_bytes + pk->K->table_bytes + pk->Z->table_bytes;
  info[5] = pk->B2->table_bytes;
  info[6] = pk->Z->plan.shared;
  info[7] = pk->B2->plan.shared;
  info[8] = pk->Z->plan.comb;
  info[9] = pk->B2->plan.comb;
  return ZKMI_OK;
}

void zkmi_cs_free(zkmi_ctx* ctx, zkmi_cs* cs) {
  if (!cs) return;
  if (ctx) hipSetDevice(ctx->device);
  if (cs->program) hipFree(cs->program);
  if (cs->consts) hipFree(cs->consts);
  delete cs;
}

int zkmi_cs_load(zkmi_ctx* ctx, const zkmi_cs_desc* d, zkmi_cs** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!d || !out) return ZKMI_ERR_ARG;
  const uint32_t S = d->lanes_per_proof;
  if (S == 0 || S > 64 || (S & (S - 1))) {
    ctx->err = "cs: lanes_per_proof must be a power of two, 1 .. 64";
    return ZKMI_ERR_ARG;
  }
  // validate every slot / constant / row index on the host before anything reaches a kernel
  enum { CLS_M = 1, CLS_X, CLS_A, CLS_R, CLS_I, CLS_BITS, CLS_BINV, CLS_HIST, CLS_COMMIT, CLS_EMUL = 11,
         CLS_LIMBS = 12 };
  enum { OP_HIST = 20, OP_HQ = 21, OP_COMMIT = 22, OP_BXOR = 23, OP_BAND = 24, OP_EMUL = 25 };
  std::vector<std::pair<uint32_t, uint32_t>> commit_rows;
  bool has_emul = false;
  const uint32_t* p = d->program;
  const size_t stride = (size_t)(1 + S) * 4;
  std::vector<uint8_t> row_seen(d->n_constraints, 0);
  uint32_t n_abc = 0;
  auto bad = [&](uint32_t r) {
    ctx->err = "cs: malformed program row " + std::to_string(r);
    return ZKMI_ERR_ARG;
  };
  for (uint32_t r = 0; r < d->n_rows; r++) {
    const uint32_t* h = p + r * stride;
    const uint32_t cls = h[0] & 0xff;
    if (h[0] & 0x100) return bad(r);   // a pair row outside a BATCHINV step
    if (cls == CLS_BINV) {
      const uint32_t npairs = h[1], nrows = h[2];
      if ((uint64_t)r + nrows >= (uint64_t)d->n_rows + 0 && nrows) {
        if ((uint64_t)r + nrows > d->n_rows - 1) return bad(r);
      }
      if (nrows != (npairs + S - 1) / S) return bad(r);
      std::vector<uint32_t> dsts, srcs;
      for (uint32_t t = 1; t <= nrows; t++) {
        const uint32_t* hh = p + (size_t)(r + t) * stride;
        if (hh[0] != (CLS_BINV | 0x100u)) return bad(r + t);
        for (uint32_t l = 0; l < S; l++) {
          const uint32_t* q = hh + 4 * (1 + l);
          if ((q[0] & 0x1f) == OP_END) continue;
          if ((q[0] & 0x1f) != OP_PAIR || q[1] >= d->n_slots || q[2] >= d->n_slots || q[1] == q[2])
            return bad(r + t);
          dsts.push_back(q[1]);
          srcs.push_back(q[2]);
        }
      }
      if (dsts.size() != npairs) return bad(r);
      std::sort(srcs.begin(), srcs.end());
      for (uint32_t x : dsts)   // no dst aliases any src (dst rows are the prefix scratch)
        if (std::binary_search(srcs.begin(), srcs.end(), x)) return bad(r);
      r += nrows;
      continue;
    }
    if (cls == CLS_HIST || cls == CLS_EMUL) {
      // header (class, n queries, n rows, table size), quad 0 = (OP_HIST, first counter wire);
      // the rows that follow hold the query slots; every quad carries class 0.
      // CLS_EMUL: header (class, na + nb, n rows, aux), quad 0 = (OP_EMUL, first wire, 0, aux),
      // aux = nout | na << 8 | first modulus constant << 12; rows: the limb slots of a, then of b
      const uint32_t nq = h[1], nrows = h[2], size = h[3];
      const uint32_t* q0 = h + 4;
      if (nrows != (nq + S - 1) / S || (uint64_t)r + nrows > (uint64_t)d->n_rows - 1) return bad(r);
      if (cls == CLS_HIST) {
        if (q0[0] != OP_HIST || size == 0 || (uint64_t)q0[1] + size > d->n_wires) return bad(r);
      } else {
        const uint32_t nout = size & 0xff, na = (size >> 8) & 0xf, c0 = size >> 12;
        if (q0[0] != OP_EMUL || q0[3] != size || nout < 5 || nout > 12 || na < 1 || na > 4 ||
            nq <= na || nq - na > 4 || (uint64_t)q0[1] + nout > d->n_wires ||
            (uint64_t)c0 + 4 > d->n_consts)
          return bad(r);
        has_emul = true;
      }
      for (uint32_t l = 1; l < S; l++)
        if (h[4 * (1 + l)] != 0) return bad(r);
      uint32_t seen = 0;
      for (uint32_t t = 1; t <= nrows; t++) {
        const uint32_t* hh = p + (size_t)(r + t) * stride;
        if (hh[0] != (cls | 0x100u)) return bad(r + t);
        for (uint32_t l = 0; l < S; l++) {
          const uint32_t* q = hh + 4 * (1 + l);
          if (q[0] == OP_END) {
            if (seen < nq && (t - 1) * S + l < nq) return bad(r + t);
            continue;
          }
          if (q[0] != OP_HQ || q[2] >= d->n_slots || (t - 1) * S + l != seen) return bad(r + t);
          seen++;
        }
      }
      if (seen != nq) return bad(r);
      r += nrows;
      continue;
    }
    if (cls == CLS_COMMIT) {
      // one row: header (class, n operands, 0, commitment index), quad 0 = (OP_COMMIT, wire, 0, index)
      const uint32_t* q0 = h + 4;
      if (q0[0] != OP_COMMIT || q0[1] >= d->n_wires || h[3] != commit_rows.size() || q0[3] != h[3])
        return bad(r);
      for (uint32_t l = 1; l < S; l++)
        if (h[4 * (1 + l)] != 0) return bad(r);
      commit_rows.emplace_back(r, q0[1]);
      continue;
    }
    if (cls == CLS_LIMBS) {
      // up to S short decompositions: quads (OP_BITS, first wire, source slot, count | width << 16)
      // with class bits 0, idle quads all zero
      bool any = false;
      for (uint32_t l = 0; l < S; l++) {
        const uint32_t* q = h + 4 * (1 + l);
        if (q[0] == 0 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
          if (l == 0) return bad(r);   // the step's class is read from quad 0
          continue;
        }
        const uint32_t n = q[3] & 0xffffu, wd = q[3] >> 16;
        if (q[0] != OP_BITS || n == 0 || n > 16 || wd > 16 || q[2] >= d->n_slots ||
            (uint64_t)q[1] + n > d->n_slots)
          return bad(r);
        any = true;
      }
      if (!any) return bad(r);
      continue;
    }
    if (cls < CLS_M || cls > CLS_BITS) return bad(r);
    for (uint32_t l = 0; l < S; l++) {
      const uint32_t* q = h + 4 * (1 + l);
      const uint32_t op = q[0] & 0x1f, k = q[0] >> 9, dst = q[1], a = q[2], b = q[3];
      if (((q[0] >> 6) & 7u) != cls) return bad(r);   // every quad carries its step's class
      if (op == OP_END) continue;
      bool ok = false, emits = false;
      switch (cls) {
        case CLS_M:
          emits = op == OP_MULABC;
          ok = (op == OP_MUL || op == OP_MULABC) ? (dst < d->n_slots && a < d->n_slots && b < d->n_slots)
               : op == OP_MULC                   ? (dst < d->n_slots && a < d->n_slots && b < d->n_consts)
               : op == OP_FMA  ? (dst < d->n_slots && a < d->n_slots && b < d->n_slots && k < d->n_slots)
               : op == OP_FMAC ? (dst < d->n_slots && a < d->n_slots && b < d->n_consts && k < d->n_slots)
                               : false;
          break;
        case CLS_X:
          emits = op == OP_XORABC;
          ok = (op == OP_XORABC || op == OP_XOR) && dst < d->n_slots && a < d->n_slots &&
               b < d->n_slots;
          break;
        case CLS_A:
          ok = (op == OP_ADD || op == OP_SUB) ? (dst < d->n_slots && a < d->n_slots && b < d->n_slots)
               : op == OP_ADDC                ? (dst < d->n_slots && a < d->n_slots && b < d->n_consts)
               : (op == OP_NEG || op == OP_COPY) ? (dst < d->n_slots && a < d->n_slots)
               : op == OP_SETC                   ? (dst < d->n_slots && b < d->n_consts)
                                                 : false;
          break;
        case CLS_R:
          emits = true;
          ok = op == OP_ABC && dst < d->n_slots && a < d->n_slots && b < d->n_slots;
          break;
        case CLS_I:
          ok = op == OP_INV   ? (dst < d->n_slots && a < d->n_slots)
               : (op == OP_DIV || op == OP_BXOR || op == OP_BAND)
                   ? (dst < d->n_slots && a < d->n_slots && b < d->n_slots)
                   : false;
          break;
        case CLS_BITS:
          // b = count | width << 16: count limbs of width bits (width 0 / 1: bits)
          ok = l == 0 && op == OP_BITS && a < d->n_slots && (b >> 16) <= 16 && (b & 0xffffu) <= 256 &&
               (uint64_t)(b & 0xffffu) * ((b >> 16) ? (b >> 16) : 1u) <= 256 &&
               (uint64_t)dst + (b & 0xffffu) <= d->n_slots;
          break;
      }
      if (ok && emits) {
        ok = k < d->n_constraints && !row_seen[k];
        if (ok) {
          row_seen[k] = 1;
          n_abc++;
        }
      }
      if (!ok) return bad(r);
    }
  }
  if (n_abc != d->n_constraints) {
    ctx->err = "cs: program emits " + std::to_string(n_abc) + " constraint rows, expected " +
               std::to_string(d->n_constraints);
    return ZKMI_ERR_ARG;
  }
  auto* cs = new zkmi_cs();
  cs->n_wires = d->n_wires;
  cs->n_public = d->n_public;
  cs->n_secret = d->n_secret;
  cs->n_constraints = d->n_constraints;
  cs->n_slots = d->n_slots;
  cs->n_rows = d->n_rows;
  cs->n_consts = d->n_consts;
  cs->lanes_per_proof = S;
  cs->commit_rows = commit_rows;   // (row, commitment wire) in commitment order
  cs->has_emul = has_emul;
  int rc = upload_u32(ctx, d->program, (size_t)d->n_rows * stride, &cs->program);
  if (rc) {
    delete cs;
    return rc;
  }
  if (d->n_consts) {
    if (hipMalloc((void**)&cs->consts, (size_t)d->n_consts * 32) != hipSuccess ||
        hipMemcpy(cs->consts, d->consts, (size_t)d->n_consts * 32, hipMemcpyDefault) !=
            hipSuccess) {
      ctx->err = "cs: constant pool upload failed";
      zkmi_cs_free(ctx, cs);
      return ZKMI_ERR_HIP;
    }
    // the solver works in the 2^261 domain (solve.hip)
    if ((rc = array_to_f_domain(ctx, cs->consts, d->n_consts)) ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      zkmi_cs_free(ctx, cs);
      return rc ? rc : ZKMI_ERR_HIP;
    }
  }
  *out = cs;
  return ZKMI_OK;
}

// inputs proof-major -> slot rows 1..n_in
static int stage_inputs(zkmi_ctx* ctx, const zkmi_cs* cs, const void* inputs_dev, size_t batch,
                        size_t Bp, Fr* slots) {
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  int rc = transpose_in(ctx, inputs_dev, slots + Bp, n_in, batch, Bp, 32);
  if (rc) return rc;
  return rows_to_f_domain(ctx, slots + Bp, n_in, Bp);
}

int zkmi_solve_batch(zkmi_ctx* ctx, const zkmi_cs* cs, const void* inputs, size_t batch,
                     void* wires_out, void* abc_out, int32_t* status_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "a submitted prove batch is in flight; collect it first";
    return ZKMI_ERR_ARG;
  }
  if (batch == 0) return ZKMI_OK;
  if (!cs->commit_rows.empty()) {
    ctx->err = "solve_batch: this system has commitments; its commitment wires come from the "
               "proving key's Pedersen bases: use zkmi_prove_submit";
    return ZKMI_ERR_ARG;
  }
  const size_t Bp = round_up(batch, 64);
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  const size_t nc = cs->n_constraints ? cs->n_constraints : 1;
  Staged si(ctx), sw(ctx), sabc(ctx), sst(ctx);
  int rc;
  if ((rc = si.in(inputs, batch * n_in * 32))) return rc;
  if (wires_out && (rc = sw.out(wires_out, batch * (size_t)cs->n_wires * 32))) return rc;
  if (abc_out && (rc = sabc.out(abc_out, 3 * batch * (size_t)cs->n_constraints * 32))) return rc;
  if ((rc = sst.out(status_out, batch * 4))) return rc;
  void *slots, *a, *b, *c, *st;
  if ((rc = ensure_scratch(ctx, 0, (size_t)cs->n_slots * Bp * 32, &slots)) ||
      (rc = ensure_scratch(ctx, 1, nc * Bp * 32, &a)) ||
      (rc = ensure_scratch(ctx, 2, nc * Bp * 32, &b)) ||
      (rc = ensure_scratch(ctx, 3, nc * Bp * 32, &c)) ||
      (rc = ensure_scratch(ctx, 5, Bp * 4, &st)))
    return rc;
  if ((rc = stage_inputs(ctx, cs, si.dev, batch, Bp, (Fr*)slots))) return rc;
  if ((rc = solve_bi(ctx, cs, (Fr*)slots, (Fr*)a, (Fr*)b, (Fr*)c, (int32_t*)st, Bp))) return rc;
  // back to gnark's image for the caller
  if (wires_out && ((rc = rows_to_std_domain(ctx, (Fr*)slots, cs->n_wires, Bp)) ||
                    (rc = transpose_out(ctx, slots, sw.dev, cs->n_wires, batch, Bp, 32))))
    return rc;
  if (abc_out) {
    if ((rc = rows_to_std_domain(ctx, (Fr*)a, cs->n_constraints, Bp)) ||
        (rc = rows_to_std_domain(ctx, (Fr*)b, cs->n_constraints, Bp)) ||
        (rc = rows_to_std_domain(ctx, (Fr*)c, cs->n_constraints, Bp)))
      return rc;
    const size_t stride = batch * (size_t)cs->n_constraints * 32;
    if ((rc = transpose_out(ctx, a, sabc.dev, cs->n_constraints, batch, Bp, 32)) ||
        (rc = transpose_out(ctx, b, (char*)sabc.dev + stride, cs->n_constraints, batch, Bp, 32)) ||
        (rc = transpose_out(ctx, c, (char*)sabc.dev + 2 * stride, cs->n_constraints, batch, Bp,
                            32)))
      return rc;
  }
  ZK_HIP(hipMemcpyAsync(sst.dev, st, batch * 4, hipMemcpyDeviceToDevice, ctx->stream));
  if ((rc = sw.finish()) || (rc = sabc.finish()) || (rc = sst.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  return ZKMI_OK;
}

static bool any_pending(zkmi_ctx* ctx) { return ctx->sets[0].pending || ctx->sets[1].pending; }

// Stage 1 of a prove: inputs -> value file, witness solve.  Runs on stream2 so that it overlaps
// the NTT/MSM kernels of the previously submitted batch (the solve is a 16-wavefront latency chain).
int zkmi_prove_submit(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_cs* cs, const void* inputs,
                      size_t batch, const void* rs) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (batch == 0) {
    ctx->err = "prove: empty batch";
    return ZKMI_ERR_ARG;
  }
  if (pk->n_wires != cs->n_wires) {
    ctx->err = "prove: proving key and constraint system disagree on the number of wires";
    return ZKMI_ERR_ARG;
  }
  if (cs->n_constraints > (1u << pk->log_n)) {
    ctx->err = "prove: domain smaller than the number of constraints";
    return ZKMI_ERR_ARG;
  }
  if (cs->commit_rows.size() != pk->commits.size()) {
    ctx->err = "prove: the constraint system has " + std::to_string(cs->commit_rows.size()) +
               " commitments, the proving key " + std::to_string(pk->commits.size());
    return ZKMI_ERR_ARG;
  }
  for (size_t i = 0; i < pk->commits.size(); i++)
    if (cs->commit_rows[i].second != pk->commits[i].wire) {
      ctx->err = "prove: commitment " + std::to_string(i) + " lands on different wires in the "
                 "constraint system and in the proving key";
      return ZKMI_ERR_ARG;
    }
  const int si = ctx->next_submit;
  zkmi_ctx::ProveSet& S = ctx->sets[si];
  if (S.pending) {
    ctx->err = "prove: two batches already in flight; collect one first";
    return ZKMI_ERR_ARG;
  }
  const size_t Bp = round_up(batch, 64);
  const size_t n = (size_t)1 << pk->log_n;
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  int rc;
  const int base = si == 0 ? 0 : 8;   // scratch slots 0-3,5 (set 0) / 8-11,13 (set 1)
  void* misc;
  // Size the other set too, so that steady-state submits never allocate -- but only while that
  // set is idle: a pending set keeps raw pointers into its buffers (S.slots, S.a, ...), and its
  // solve / quotient / MSM kernels may be running on them.  A pending set is resized by its own
  // next submit, after its collect.
  if (!ctx->sets[si ^ 1].pending) {
    void* dummy;
    const int ob = si == 0 ? 8 : 0;
    if ((rc = ensure_scratch(ctx, ob + 0, (size_t)cs->n_slots * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 1, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 2, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 3, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 5, Bp * (96 + 4) + batch * (n_in * 32 + 64), &dummy)) ||
        (rc = ensure_scratch(ctx, si == 0 ? 15 : 14, Bp * (7 * 128 + 2 * 256 + 256), &dummy)))
      return rc;
  }
  if ((rc = ensure_scratch(ctx, si == 0 ? 14 : 15, Bp * (7 * 128 + 2 * 256 + 256 + 256 * (4 * 128 + 256)),
                           &S.sums)))
    return rc;
  S.heavy_enqueued = false;
  if ((rc = ensure_scratch(ctx, base + 0, (size_t)cs->n_slots * Bp * 32, &S.slots)) ||
      (rc = ensure_scratch(ctx, base + 1, n * Bp * 32, &S.a)) ||
      (rc = ensure_scratch(ctx, base + 2, n * Bp * 32, &S.b)) ||
      (rc = ensure_scratch(ctx, base + 3, n * Bp * 32, &S.c)) ||
      (rc = ensure_scratch(ctx, base + 5, Bp * (96 + 4) + batch * (n_in * 32 + 64), &misc)))
    return rc;
  S.rs = misc;
  S.st = (char*)misc + Bp * 96;
  char* stage_in = (char*)misc + Bp * 100;
  char* stage_rs = stage_in + batch * n_in * 32;
  hipStream_t q = ctx->stream2;
  // caller buffers are consumed before this function returns only if they are host memory
  // (pageable copies are synchronous); device buffers must stay valid until the collect
  const void* in_dev = inputs;
  const void* rs_dev = rs;
  if (!is_device_ptr(inputs)) {
    ZK_HIP(hipMemcpyAsync(stage_in, inputs, batch * n_in * 32, hipMemcpyHostToDevice, q));
    in_dev = stage_in;
  }
  if (!is_device_ptr(rs)) {
    ZK_HIP(hipMemcpyAsync(stage_rs, rs, batch * 64, hipMemcpyHostToDevice, q));
    rs_dev = stage_rs;
  }
  hipStream_t saved = ctx->stream;
  ctx->stream = q;  // the helpers launch on ctx->stream
  hipEventRecord(S.ev0, q);
  rc = transpose_in(ctx, in_dev, (Fr*)S.slots + Bp, n_in, batch, Bp, 32);
  if (!rc) rc = rows_to_f_domain(ctx, (Fr*)S.slots + Bp, n_in, Bp);
  if (!rc) rc = transpose_in(ctx, rs_dev, S.rs, 2, batch, Bp, 32);
  S.batch = batch;
  S.Bp = Bp;
  S.pk = pk;
  S.cs = cs;
  S.n_constraints = cs->n_constraints;
  S.f_domain = true;
  if (!rc && cs->commit_rows.empty()) {
    rc = solve_bi(ctx, cs, (Fr*)S.slots, (Fr*)S.a, (Fr*)S.b, (Fr*)S.c, (int32_t*)S.st, Bp);
  } else if (!rc) {
    // commitment extension: the program stops at every COMMIT row; the commitment is an MSM over
    // the wires solved so far, its hash becomes the commitment wire's value (commit.hip: the
    // host blocks on this stream for the hash -- the main stream keeps running the previous
    // batch's MSMs meanwhile)
    rc = solve_init(ctx, (Fr*)S.slots, (int32_t*)S.st, Bp);
    uint32_t begin = 0;
    for (size_t i = 0; !rc && i < cs->commit_rows.size(); i++) {
      rc = solve_rows(ctx, cs, (Fr*)S.slots, (Fr*)S.a, (Fr*)S.b, (Fr*)S.c, (int32_t*)S.st, Bp, begin,
                      cs->commit_rows[i].first);
      if (!rc) rc = commit_phase(ctx, S, (uint32_t)i, true);
      begin = cs->commit_rows[i].first + 1;
    }
    if (!rc)
      rc = solve_rows(ctx, cs, (Fr*)S.slots, (Fr*)S.a, (Fr*)S.b, (Fr*)S.c, (int32_t*)S.st, Bp, begin,
                      cs->n_rows);
    if (!rc) rc = commit_finish_submit(ctx, S);
  }
  hipEventRecord(S.ev1, q);
  ctx->stream = saved;
  if (rc) return rc;
  S.pending = true;
  ctx->next_submit ^= 1;
  return ZKMI_OK;
}

struct SumsView {
  G1XYZZ *sA, *sB1, *sK, *sZ, *tR, *tS, *tNRS;
  G2XYZZ *sB2, *tS2;
  ProofOut* proofs;
  G1XYZZ* w1[4];   // deferred window sums of A, B1, K, Z ([<= 256][Bp] each)
  G2XYZZ* w2;
  SumsView(void* base, size_t Bp) {
    char* m = (char*)base;
    sA = (G1XYZZ*)m;   m += Bp * 128;
    sB1 = (G1XYZZ*)m;  m += Bp * 128;
    sK = (G1XYZZ*)m;   m += Bp * 128;
    sZ = (G1XYZZ*)m;   m += Bp * 128;
    tR = (G1XYZZ*)m;   m += Bp * 128;
    tS = (G1XYZZ*)m;   m += Bp * 128;
    tNRS = (G1XYZZ*)m; m += Bp * 128;
    sB2 = (G2XYZZ*)m;  m += Bp * 256;
    tS2 = (G2XYZZ*)m;  m += Bp * 256;
    proofs = (ProofOut*)m;  m += Bp * 256;
    for (int i = 0; i < 4; i++) {
      w1[i] = (G1XYZZ*)m;
      m += 256 * Bp * 128;
    }
    w2 = (G2XYZZ*)m;
  }
};

// The ALU-heavy part of stage 2 (quotient, five MSMs, the one-base delta MSMs) of set `si`, on the
// main stream.  Everything it touches besides the set's own buffers (NTT scratch, MSM partials) is
// only used on the main stream, so consecutive batches simply queue up behind each other.
static int enqueue_heavy(zkmi_ctx* ctx, int si) {
  zkmi_ctx::ProveSet& S = ctx->sets[si];
  const zkmi_pk* pk = S.pk;
  const bool fd = S.f_domain;
  const size_t Bp = S.Bp, n = (size_t)1 << pk->log_n;
  NttPlan* plan;
  int rc = get_plan(ctx, (int)pk->log_n, &plan);
  if (rc) return rc;
  void* t0;
  if ((rc = ensure_scratch(ctx, 4, n * Bp * 32, &t0))) return rc;
  SumsView v(S.sums, Bp);
  Fr* rs_bi = (Fr*)S.rs;
  Fr* slots = (Fr*)S.slots;
  ZK_HIP(hipStreamWaitEvent(ctx->stream, S.ev1, 0));
  hipEventRecord(S.evq[0], ctx->stream);
  Fr* h;
  if ((rc = compute_h_bi(ctx, plan, (Fr*)S.a, (Fr*)S.b, (Fr*)S.c, (Fr*)t0, Bp, S.n_constraints,
                         &h, fd)))
    return rc;
  hipEventRecord(S.evq[1], ctx->stream);
  S.msm_ev_used = 0;
  ctx->msm_ev_set = si;
  // shared-table plans: stop at the window sums, the Horner step runs with the assembly
  const bool d1 = pk->Z->plan.shared || pk->Z->plan.comb;
  const bool d2 = pk->B2->plan.shared || pk->B2->plan.comb;
  hipStream_t q3 = ctx->stream3;
  if ((rc = msm_run(ctx, pk->A, slots, pk->a_wire, Bp, v.sA, fd, d1 ? v.w1[0] : nullptr, q3)) ||
      (rc = msm_run(ctx, pk->B1, slots, pk->b_wire, Bp, v.sB1, fd, d1 ? v.w1[1] : nullptr, q3)) ||
      (rc = msm_run(ctx, pk->K, slots, pk->k_wire, Bp, v.sK, fd, d1 ? v.w1[2] : nullptr, q3)) ||
      (rc = msm_run(ctx, pk->Z, h, nullptr, Bp, v.sZ, false, d1 ? v.w1[3] : nullptr, q3))) {
    ctx->msm_ev_set = -1;
    return rc;
  }
  hipEventRecord(S.evq[2], ctx->stream);
  rc = msm_run(ctx, pk->B2, slots, pk->b_wire, Bp, v.sB2, fd, d2 ? v.w2 : nullptr, q3);
  ctx->msm_ev_set = -1;
  if (rc) return rc;
  hipEventRecord(S.evq[3], ctx->stream);
  // The multiples of delta (r, s, -rs times delta1; s times delta2) are four one-base MSMs of sixteen
  // wavefronts each: latency, not work.  They run on the assembly stream, which is where their
  // results are used, with their own partial-sum scratch (side = 2) -- on the main stream they were
  // 4.2 ms per batch during which 240 CUs idled.
  {
    hipStream_t main_stream = ctx->stream;
    ZK_HIP(hipStreamWaitEvent(q3, S.ev1, 0));   // r, s staged by the submit
    ctx->stream = q3;
    hipLaunchKernelGGL(rs_prep_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, q3, rs_bi, Bp);
    (void)((rc = msm_run(ctx, pk->D1, rs_bi, pk->idx3 + 0, Bp, v.tR)) ||
           (rc = msm_run(ctx, pk->D1, rs_bi, pk->idx3 + 1, Bp, v.tS)) ||
           (rc = msm_run(ctx, pk->D1, rs_bi, pk->idx3 + 2, Bp, v.tNRS)) ||
           (rc = msm_run(ctx, pk->D2, rs_bi, pk->idx3 + 1, Bp, v.tS2)));
    ctx->stream = main_stream;
    if (rc) return rc;
  }
  if ((rc = commit_pok(ctx, S))) return rc;   // commitment extension: proof of knowledge
  hipEventRecord(S.evq[4], ctx->stream);
  S.heavy_enqueued = true;
  return ZKMI_OK;
}

// Stage 2: quotient, MSMs, assembly of the oldest submitted batch; blocks until its proofs are
// in proofs_out / status_out.  The latency-bound assembly (16 wavefronts) runs on a third stream
// and, when another batch is already submitted, that batch's quotient + MSM kernels are queued on
// the main stream BEFORE waiting, so the assembly of batch k overlaps the quotient of batch k+1.
int zkmi_prove_collect(zkmi_ctx* ctx, void* proofs_out, int32_t* status_out) {
  return zkmi_prove_collect_ex(ctx, proofs_out, status_out, nullptr);
}

int zkmi_prove_collect_ex(zkmi_ctx* ctx, void* proofs_out, int32_t* status_out,
                          void* commitments_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  const int si = ctx->next_collect;
  zkmi_ctx::ProveSet& S = ctx->sets[si];
  if (!S.pending) {
    ctx->err = "prove: nothing submitted";
    return ZKMI_ERR_ARG;
  }
  if (!S.pk->commits.empty() && !commitments_out) {
    ctx->err = "prove: this key has commitments: collect with zkmi_prove_collect_ex";
    return ZKMI_ERR_ARG;
  }
  const zkmi_pk* pk = S.pk;
  const size_t batch = S.batch, Bp = S.Bp;
  int rc;
  if (!S.heavy_enqueued && (rc = enqueue_heavy(ctx, si))) return rc;
  SumsView v(S.sums, Bp);
  hipStream_t q3 = ctx->stream3;
  ZK_HIP(hipStreamWaitEvent(q3, S.evq[4], 0));
  hipEventRecord(S.eva[0], q3);
  if (pk->Z->plan.shared || pk->Z->plan.comb) {
    void* ws[4] = {v.w1[0], v.w1[1], v.w1[2], v.w1[3]};
    void* os[4] = {v.sA, v.sB1, v.sK, v.sZ};
    const zkmi_msm_bases* ms[4] = {pk->A, pk->B1, pk->K, pk->Z};
    bool same = true;   // an out-of-memory relaxation may have narrowed one table
    for (int i = 0; i < 3; i++)
      same = same && ms[i]->plan.bits[0] == pk->Z->plan.bits[0] &&
             ms[i]->plan.comb == pk->Z->plan.comb &&
             ms[i]->plan.comb_signed == pk->Z->plan.comb_signed;
    if (same) {
      if ((rc = msm_horner_run(ctx, q3, 4, ms, ws, os, Bp))) return rc;
    } else {
      for (int i = 0; i < 4; i++)
        if ((rc = msm_horner_run(ctx, q3, 1, ms + i, ws + i, os + i, Bp))) return rc;
    }
  }
  if (pk->B2->plan.shared || pk->B2->plan.comb) {
    void* ws[1] = {v.w2};
    void* os[1] = {v.sB2};
    const zkmi_msm_bases* ms[1] = {pk->B2};
    if ((rc = msm_horner_run(ctx, q3, 1, ms, ws, os, Bp))) return rc;
  }
  PkConsts pc{pk->alpha, pk->beta1, pk->beta2};
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, q3, v.sA, v.sB1, v.sK,
                     v.sZ, v.tR, v.tS, v.tNRS, (const Fr*)S.rs, Bp, pc, v.proofs);
  hipLaunchKernelGGL(assemble_g2_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, q3, v.sB2, v.tS2,
                     Bp, pk->beta2, v.proofs);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpyAsync(proofs_out, v.proofs, batch * 256, hipMemcpyDefault, q3));
  ZK_HIP(hipMemcpyAsync(status_out, S.st, batch * 4, hipMemcpyDefault, q3));
  if (const size_t nc = pk->commits.size()) {
    // per proof: Commitments[0 .. n-1], then CommitmentPok
    const size_t pitch = (nc + 1) * 64;
    for (size_t i = 0; i < nc; i++)
      ZK_HIP(hipMemcpy2DAsync((char*)commitments_out + i * 64, pitch,
                              (const char*)S.commit_pts + i * Bp * 64, 64, 64, batch, hipMemcpyDefault, q3));
    ZK_HIP(hipMemcpy2DAsync((char*)commitments_out + nc * 64, pitch,
                            (const char*)S.commit_pok + Bp * 128, 64, 64, batch, hipMemcpyDefault, q3));
  }
  hipEventRecord(S.eva[1], q3);
  // look ahead: queue the next batch's heavy kernels before blocking on this batch's assembly
  zkmi_ctx::ProveSet& N = ctx->sets[si ^ 1];
  if (N.pending && !N.heavy_enqueued && (rc = enqueue_heavy(ctx, si ^ 1))) return rc;
  // wait for THIS batch's copies only: the look-ahead has queued the next batch's MSM tails behind
  // them on the same stream
  ZK_HIP(hipEventSynchronize(S.eva[1]));
  S.pending = false;
  S.heavy_enqueued = false;
  ctx->next_collect ^= 1;
  float ms = 0;
  hipEventElapsedTime(&ms, S.ev0, S.ev1);
  ctx->timings[0] = ms;
  for (int i = 0; i < 3; i++) {
    hipEventElapsedTime(&ms, S.evq[i], S.evq[i + 1]);
    ctx->timings[1 + i] = ms;
  }
  hipEventElapsedTime(&ms, S.eva[0], S.eva[1]);
  ctx->timings[4] = ms;
  hipEventElapsedTime(&ms, S.evq[0], S.evq[4]);
  ctx->timings[5] = ms;  // quotient + MSMs on the main stream (solve and assembly overlap neighbours)
  ctx->timings[6] = ctx->timings[7] = 0;
  for (int i = 0; i < S.msm_ev_used; i++) {
    hipEventElapsedTime(&ms, S.msm_ev[i][0], S.msm_ev[i][1]);
    ctx->timings[S.msm_ev_group[i] == 1 ? 6 : 7] += ms;
  }
  return ZKMI_OK;
}

int zkmi_prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_cs* cs, const void* inputs,
                     size_t batch, const void* rs, void* proofs_out, int32_t* status_out) {
  if (batch == 0) return ZKMI_OK;
  if (any_pending(ctx)) {
    ctx->err = "prove_batch: batches submitted with zkmi_prove_submit are still in flight";
    return ZKMI_ERR_ARG;
  }
  int rc = zkmi_prove_submit(ctx, pk, cs, inputs, batch, rs);
  if (rc) return rc;
  return zkmi_prove_collect(ctx, proofs_out, status_out);
}

int zkmi_last_timings(zkmi_ctx* ctx, double* ms_out) {
  for (int i = 0; i < 8; i++) ms_out[i] = ctx->timings[i];
  return ZKMI_OK;
}

}  // extern "C"
