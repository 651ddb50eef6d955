// C-ABI of libzkmi.so (include/zkmi.h) and the Groth16 prove pipeline that strings the kernels
// together: solve -> quotient (7 NTTs) -> 4 G1 MSMs + 1 G2 MSM -> assembly.
// Stands in for groth16.Prove in gnark backend/groth16/bn254/prove.go [UPSTREAM-RECALL,
// SURVEY.md §3.2].  There is no CPU fallback anywhere in this file.
#include <mutex>

#include "zkmi_internal.h"
#include "ff29.h"

using namespace zk;

namespace zk {

int ensure_scratch(zkmi_ctx* ctx, int slot, size_t bytes, void** out) {
  DevBuf& s = ctx->scratch[slot];
  if (s.bytes < bytes) {
    if (s.p) {
      hipStreamSynchronize(ctx->stream);
      hipFree(s.p);
      s.p = nullptr;
      s.bytes = 0;
    }
    hipError_t e = hipMalloc(&s.p, bytes);
    if (e != hipSuccess) {
      ctx->err = "hipMalloc(" + std::to_string(bytes) + " B) failed: " + hipGetErrorString(e);
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
// One lane per proof.  Ar = sumA + alpha + r*delta ; Bs1 = sumB1 + beta + s*delta ;
// Bs = sumB2 + beta2 + s*delta2 ; Krs = sumK + sumZ - rs*delta + s*Ar + r*Bs1.
struct PkConsts {
  G1Affine alpha, beta1, delta1;
  G2Affine beta2, delta2;
};
struct ProofOut {
  G1Affine ar, krs;
  G2Affine bs;
};

__global__ __launch_bounds__(64) void assemble_kernel(const G1XYZZ* sA, const G1XYZZ* sB1,
                                                      const G1XYZZ* sK, const G1XYZZ* sZ,
                                                      const G2XYZZ* sB2, const Fr* rs, size_t Bp,
                                                      PkConsts pk, ProofOut* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp) return;
  const Fr rm = rs[i], sm = rs[Bp + i];
  const Fr r = from_mont(rm), s = from_mont(sm);
  const Fr nrs = from_mont(neg(mul(rm, sm)));
  G1XYZZ AR = sA[i];
  madd(AR, pk.alpha);
  {
    G1XYZZ t = scalar_mul(pk.delta1, r.v);
    padd(AR, t);
  }
  const G1Affine ar = to_affine(AR);
  G1XYZZ BS1 = sB1[i];
  madd(BS1, pk.beta1);
  {
    G1XYZZ t = scalar_mul(pk.delta1, s.v);
    padd(BS1, t);
  }
  const G1Affine bs1 = to_affine(BS1);
  G1XYZZ KRS = sK[i];
  {
    G1XYZZ z = sZ[i];
    padd(KRS, z);
    G1XYZZ t = scalar_mul(pk.delta1, nrs.v);
    padd(KRS, t);
    t = scalar_mul(ar, s.v);
    padd(KRS, t);
    t = scalar_mul(bs1, r.v);
    padd(KRS, t);
  }
  out[i].ar = ar;
  out[i].krs = to_affine(KRS);
}

__global__ __launch_bounds__(64) void assemble_g2_kernel(const G2XYZZ* sB2, const Fr* rs, size_t Bp,
                                                         G2Affine beta2, G2Affine delta2,
                                                         ProofOut* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bp) return;
  const Fr s = from_mont(rs[Bp + i]);
  G2XYZZ BS = sB2[i];
  madd(BS, beta2);
  G2XYZZ t = scalar_mul(delta2, s.v);
  padd(BS, t);
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
  for (auto& e : ctx->ev) hipEventCreate(&e);
  for (auto& p : ctx->msm_ev) {
    hipEventCreate(&p[0]);
    hipEventCreate(&p[1]);
  }
  ctx->plans.reserve(32);
  *out = ctx;
  return ZKMI_OK;
}

void zkmi_destroy(zkmi_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->plans) {
    hipFree(p.tw_fwd);
    hipFree(p.tw_inv);
    hipFree(p.coset_fwd);
    hipFree(p.coset_inv);
  }
  for (auto& s : ctx->scratch)
    if (s.p) hipFree(s.p);
  for (auto& e : ctx->ev)
    if (e) hipEventDestroy(e);
  for (auto& p : ctx->msm_ev) {
    if (p[0]) hipEventDestroy(p[0]);
    if (p[1]) hipEventDestroy(p[1]);
  }
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

int zkmi_msm_bases_load(zkmi_ctx* ctx, int group, const void* bases, size_t n, int window_bits,
                        zkmi_msm_bases** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!out) return ZKMI_ERR_ARG;
  Staged sb(ctx);
  int rc = sb.in(bases, n * (group == 1 ? 64 : 128));
  if (rc) return rc;
  return msm_bases_build(ctx, group, sb.dev, n, window_bits, out);
}

void zkmi_msm_bases_free(zkmi_ctx* ctx, zkmi_msm_bases* b) {
  if (!b) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
  }
  if (b->table) hipFree(b->table);
  delete b;
}

int zkmi_msm_batch(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const void* scalars, size_t batch,
                   void* out) {
  ZK_HIP(hipSetDevice(ctx->device));
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
  hipMemsetAsync(sbi, 0, Bp * 32, ctx->stream);
  hipMemcpyAsync(sbi, ss.dev, n * 32, hipMemcpyDeviceToDevice, ctx->stream);
  rc = msm_run(ctx, b, (const Fr*)sbi, nullptr, Bp, acc);
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
  if (pk->a_wire) hipFree(pk->a_wire);
  if (pk->b_wire) hipFree(pk->b_wire);
  if (pk->k_wire) hipFree(pk->k_wire);
  delete pk;
}

int zkmi_pk_load(zkmi_ctx* ctx, const zkmi_pk_desc* d, zkmi_pk** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!d || !out) return ZKMI_ERR_ARG;
  if (d->n_z + 1 != (1u << d->log_n)) {
    ctx->err = "pk: n_z must equal 2^log_n - 1";
    return ZKMI_ERR_ARG;
  }
  auto* pk = new zkmi_pk();
  pk->log_n = d->log_n;
  pk->n_wires = d->n_wires;
  pk->n_a = d->n_a;
  pk->n_b = d->n_b;
  pk->n_k = d->n_k;
  pk->n_z = d->n_z;
  int rc;
  // one window size per group for the whole key, sized against free HBM
  const int c1 = d->window_bits_g1
                     ? (int)d->window_bits_g1
                     : default_window((size_t)d->n_a + d->n_b + d->n_k + d->n_z, 1);
  const int c2 = d->window_bits_g2 ? (int)d->window_bits_g2 : default_window(d->n_b, 2);
  if ((rc = upload_u32(ctx, d->a_wire, d->n_a, &pk->a_wire)) ||
      (rc = upload_u32(ctx, d->b_wire, d->n_b, &pk->b_wire)) ||
      (rc = upload_u32(ctx, d->k_wire, d->n_k, &pk->k_wire)) ||
      (rc = zkmi_msm_bases_load(ctx, 1, d->g1_a, d->n_a, c1, &pk->A)) ||
      (rc = zkmi_msm_bases_load(ctx, 1, d->g1_b, d->n_b, c1, &pk->B1)) ||
      (rc = zkmi_msm_bases_load(ctx, 1, d->g1_k, d->n_k, c1, &pk->K)) ||
      (rc = zkmi_msm_bases_load(ctx, 1, d->g1_z, d->n_z, c1, &pk->Z)) ||
      (rc = zkmi_msm_bases_load(ctx, 2, d->g2_b, d->n_b, c2, &pk->B2))) {
    zkmi_pk_free(ctx, pk);
    return rc;
  }
  hipMemcpy(&pk->alpha, d->g1_alpha, 64, hipMemcpyDefault);
  hipMemcpy(&pk->beta1, d->g1_beta, 64, hipMemcpyDefault);
  hipMemcpy(&pk->delta1, d->g1_delta, 64, hipMemcpyDefault);
  hipMemcpy(&pk->beta2, d->g2_beta, 128, hipMemcpyDefault);
  hipMemcpy(&pk->delta2, d->g2_delta, 128, hipMemcpyDefault);
  *out = pk;
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
  // validate every slot / constant index on the host before anything reaches a kernel
  const uint32_t* p = d->program;
  uint32_t n_abc = 0;
  for (uint32_t i = 0; i < d->n_ops; i++) {
    const uint32_t op = p[4 * i] & 0xff, dst = p[4 * i + 1], a = p[4 * i + 2], b = p[4 * i + 3];
    bool ok = true;
    switch (op) {
      case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV:
        ok = dst < d->n_slots && a < d->n_slots && b < d->n_slots;
        break;
      case OP_MULC: case OP_ADDC:
        ok = dst < d->n_slots && a < d->n_slots && b < d->n_consts;
        break;
      case OP_NEG: case OP_INV: case OP_COPY:
        ok = dst < d->n_slots && a < d->n_slots;
        break;
      case OP_SETC:
        ok = dst < d->n_slots && b < d->n_consts;
        break;
      case OP_BITS:
        ok = a < d->n_slots && b <= 256 && (uint64_t)dst + b <= d->n_slots;
        break;
      case OP_ABC:
        ok = dst < d->n_slots && a < d->n_slots && b < d->n_slots;
        n_abc++;
        break;
      case OP_BATCHINV: {
        // dst = number of (OP_PAIR, dst, src) rows that follow; distinct wire slots on both sides
        ok = (uint64_t)i + dst < d->n_ops;
        for (uint32_t k = 1; ok && k <= dst; k++) {
          const uint32_t* q = p + 4 * (size_t)(i + k);
          ok = (q[0] & 0xff) == OP_PAIR && q[1] < d->n_wires && q[2] < d->n_wires && q[1] != q[2];
          for (uint32_t k2 = 1; ok && k2 <= dst; k2++)
            ok = q[1] != p[4 * (size_t)(i + k2) + 2];   // no dst aliases any src
        }
        if (ok) i += dst;
        break;
      }
      default:
        ok = false;
    }
    if (!ok) {
      ctx->err = "cs: malformed instruction at " + std::to_string(i);
      return ZKMI_ERR_ARG;
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
  cs->n_ops = d->n_ops;
  cs->n_consts = d->n_consts;
  int rc = upload_u32(ctx, d->program, (size_t)(d->n_ops + 1) * 4, &cs->program);
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
  }
  *out = cs;
  return ZKMI_OK;
}

// inputs proof-major -> slot rows 1..n_in
static int stage_inputs(zkmi_ctx* ctx, const zkmi_cs* cs, const void* inputs_dev, size_t batch,
                        size_t Bp, Fr* slots) {
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  return transpose_in(ctx, inputs_dev, slots + Bp, n_in, batch, Bp, 32);
}

int zkmi_solve_batch(zkmi_ctx* ctx, const zkmi_cs* cs, const void* inputs, size_t batch,
                     void* wires_out, void* abc_out, int32_t* status_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (batch == 0) return ZKMI_OK;
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
  if (wires_out && (rc = transpose_out(ctx, slots, sw.dev, cs->n_wires, batch, Bp, 32))) return rc;
  if (abc_out) {
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

int zkmi_prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_cs* cs, const void* inputs,
                     size_t batch, const void* rs, void* proofs_out, int32_t* status_out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (batch == 0) return ZKMI_OK;
  if (pk->n_wires != cs->n_wires) {
    ctx->err = "prove: proving key and constraint system disagree on the number of wires";
    return ZKMI_ERR_ARG;
  }
  if (cs->n_constraints > (1u << pk->log_n)) {
    ctx->err = "prove: domain smaller than the number of constraints";
    return ZKMI_ERR_ARG;
  }
  NttPlan* plan;
  int rc = get_plan(ctx, (int)pk->log_n, &plan);
  if (rc) return rc;
  const size_t Bp = round_up(batch, 64);
  const size_t n = (size_t)1 << pk->log_n;
  const size_t n_in = cs->n_public - 1 + cs->n_secret;
  Staged si(ctx), srs(ctx), sp(ctx), sst(ctx);
  if ((rc = si.in(inputs, batch * n_in * 32)) || (rc = srs.in(rs, batch * 64)) ||
      (rc = sp.out(proofs_out, batch * 256)) || (rc = sst.out(status_out, batch * 4)))
    return rc;
  void *slots, *a, *b, *c, *t0, *misc;
  const size_t misc_bytes = Bp * (4 /*status*/ + 64 /*rs*/ + 4 * 128 + 256 /*sums*/ + 256);
  if ((rc = ensure_scratch(ctx, 0, (size_t)cs->n_slots * Bp * 32, &slots)) ||
      (rc = ensure_scratch(ctx, 1, n * Bp * 32, &a)) ||
      (rc = ensure_scratch(ctx, 2, n * Bp * 32, &b)) ||
      (rc = ensure_scratch(ctx, 3, n * Bp * 32, &c)) ||
      (rc = ensure_scratch(ctx, 4, n * Bp * 32, &t0)) ||
      (rc = ensure_scratch(ctx, 5, misc_bytes, &misc)))
    return rc;
  char* m = (char*)misc;
  Fr* rs_bi = (Fr*)m;                       m += Bp * 64;
  G1XYZZ* sA = (G1XYZZ*)m;                  m += Bp * 128;
  G1XYZZ* sB1 = (G1XYZZ*)m;                 m += Bp * 128;
  G1XYZZ* sK = (G1XYZZ*)m;                  m += Bp * 128;
  G1XYZZ* sZ = (G1XYZZ*)m;                  m += Bp * 128;
  G2XYZZ* sB2 = (G2XYZZ*)m;                 m += Bp * 256;
  ProofOut* proofs = (ProofOut*)m;          m += Bp * 256;
  int32_t* st = (int32_t*)m;

  hipEventRecord(ctx->ev[0], ctx->stream);
  if ((rc = stage_inputs(ctx, cs, si.dev, batch, Bp, (Fr*)slots))) return rc;
  if ((rc = transpose_in(ctx, srs.dev, rs_bi, 2, batch, Bp, 32))) return rc;
  if ((rc = solve_bi(ctx, cs, (Fr*)slots, (Fr*)a, (Fr*)b, (Fr*)c, st, Bp))) return rc;
  hipEventRecord(ctx->ev[1], ctx->stream);
  Fr* h;
  if ((rc = compute_h_bi(ctx, plan, (Fr*)a, (Fr*)b, (Fr*)c, (Fr*)t0, Bp, cs->n_constraints, &h)))
    return rc;
  hipEventRecord(ctx->ev[2], ctx->stream);
  ctx->msm_ev_used = 0;
  ctx->msm_ev_on = true;
  if ((rc = msm_run(ctx, pk->A, (const Fr*)slots, pk->a_wire, Bp, sA)) ||
      (rc = msm_run(ctx, pk->B1, (const Fr*)slots, pk->b_wire, Bp, sB1)) ||
      (rc = msm_run(ctx, pk->K, (const Fr*)slots, pk->k_wire, Bp, sK)) ||
      (rc = msm_run(ctx, pk->Z, h, nullptr, Bp, sZ)))
    return rc;
  hipEventRecord(ctx->ev[3], ctx->stream);
  if ((rc = msm_run(ctx, pk->B2, (const Fr*)slots, pk->b_wire, Bp, sB2))) return rc;
  hipEventRecord(ctx->ev[4], ctx->stream);
  ctx->msm_ev_on = false;
  PkConsts pc{pk->alpha, pk->beta1, pk->delta1, pk->beta2, pk->delta2};
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, sA, sB1,
                     sK, sZ, sB2, rs_bi, Bp, pc, proofs);
  hipLaunchKernelGGL(assemble_g2_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, sB2,
                     rs_bi, Bp, pk->beta2, pk->delta2, proofs);
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpyAsync(sp.dev, proofs, batch * 256, hipMemcpyDeviceToDevice, ctx->stream));
  ZK_HIP(hipMemcpyAsync(sst.dev, st, batch * 4, hipMemcpyDeviceToDevice, ctx->stream));
  hipEventRecord(ctx->ev[5], ctx->stream);
  if ((rc = sp.finish()) || (rc = sst.finish())) return rc;
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 5; i++) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]);
    ctx->timings[i] = ms;
  }
  float tot = 0;
  hipEventElapsedTime(&tot, ctx->ev[0], ctx->ev[5]);
  ctx->timings[5] = tot;
  ctx->timings[6] = ctx->timings[7] = 0;
  for (int i = 0; i < ctx->msm_ev_used; i++) {
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->msm_ev[i][0], ctx->msm_ev[i][1]);
    ctx->timings[ctx->msm_ev_group[i] == 1 ? 6 : 7] += ms;
  }
  // per-proof status is reported in status_out; the call itself succeeded
  return ZKMI_OK;
}

int zkmi_last_timings(zkmi_ctx* ctx, double* ms_out) {
  for (int i = 0; i < 8; i++) ms_out[i] = ctx->timings[i];
  return ZKMI_OK;
}

}  // extern "C"
