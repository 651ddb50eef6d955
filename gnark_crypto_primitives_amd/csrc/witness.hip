// The gnark drop-in entry: Groth16 prove from SOLVED witnesses (zkmi_prove_witness_submit /
// zkmi_prove_collect), for a caller that keeps gnark's own solver -- what a cgo shim around
// groth16.Prove(ccs, pk, fullWitness) binds (call sites in the reference:
// tree/test/verifier_bn254_test.go:41,67; gnark backend/groth16/bn254/prove.go [UPSTREAM-RECALL]).
//
//   * host buffers are staged through a ring of pinned chunks owned by the context: worker threads
//     copy a chunk of proofs into page-locked memory, one DMA brings it to the device, a tiled
//     transpose scatters it into the batch-inner matrix -- all on the second stream, so a batch
//     streams in underneath the previous batch's MSM kernels.  Page-locked caller memory
//     (zkmi_host_alloc) skips the host copy;
//   * with the R1CS matrices resident (zkmi_r1cs_load: gnark's constraint.R1C terms {CID, VID} and
//     coefficient table) the caller ships the wire vector only and a = L.w, b = R.w, c = O.w are
//     formed on the device, lane = proof, terms and coefficients through the scalar unit; a.b = c is
//     checked per proof (ZKMI_ERR_UNSATISFIED in status_out).
// There is no CPU fallback in this file.
#include <algorithm>
#include <cstring>
#include <thread>

#include "zkmi_internal.h"
#include "ff29.h"
#if defined(__HIP_DEVICE_COMPILE__)
#include "ff29_asm.h"
#endif

using namespace zk;

struct zkmi_r1cs {
  uint32_t n_wires = 0, n_constraints = 0, n_coeffs = 0;
  uint32_t* ptr[3] = {};   // device, n_constraints + 1 offsets each (L, R, O)
  uint2* terms[3] = {};    // device, x = wire, y = coefficient index | kind << 30
  size_t nnz[3] = {};
  Fr* coeffs = nullptr;    // device, 2^261 images (gnark's Montgomery image times 2^5: fmul_261)
};

namespace zk {

int pointer_kind(const void* p) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain pageable memory: clear the sticky "invalid value"
    return PTR_PAGEABLE;
  }
  if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return PTR_DEVICE;
  if (attr.type == hipMemoryTypeHost) return PTR_PINNED;
  return PTR_PAGEABLE;
}

void witness_ring_free(zkmi_ctx* ctx) {
  auto& r = ctx->ring;
  for (int i = 0; i < 3; i++) {
    if (r.pinned[i]) hipHostFree(r.pinned[i]);
    if (r.dev[i]) hipFree(r.dev[i]);
    if (r.done[i]) hipEventDestroy(r.done[i]);
    r.pinned[i] = r.dev[i] = nullptr;
    r.done[i] = nullptr;
    r.busy[i] = false;
  }
  r.bytes = 0;
}

static int ring_ensure(zkmi_ctx* ctx, size_t bytes) {
  auto& r = ctx->ring;
  if (r.bytes >= bytes) return ZKMI_OK;
  for (int i = 0; i < 3; i++)
    if (r.busy[i]) {
      ZK_HIP(hipEventSynchronize(r.done[i]));
      r.busy[i] = false;
    }
  witness_ring_free(ctx);
  for (int i = 0; i < 3; i++) {
    if (hipHostMalloc(&r.pinned[i], bytes, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&r.dev[i], bytes) != hipSuccess ||
        hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      witness_ring_free(ctx);
      ctx->err = "witness staging ring: cannot allocate 3 x " + std::to_string(bytes) +
                 " B of pinned host and device memory";
      return ZKMI_ERR_OOM;
    }
  }
  r.bytes = bytes;
  return ZKMI_OK;
}

// proof-major chunk [cnt][rows] of 32-byte elements -> columns [p0, p0 + 64 * gridDim.y) of a
// batch-inner matrix [rows][half][Bp]; columns >= cnt are written as zero.  A block moves 16 rows x
// 64 proofs through LDS: reads are 512-byte runs (one proof's 16 rows), writes 1 KiB runs (one
// half-row of 64 proofs); the 33-quad pitch keeps both sides of the LDS exchange conflict-free.
__global__ __launch_bounds__(256) void transpose_tile_kernel(const uint4* __restrict__ src,
                                                             uint4* __restrict__ dst, size_t rows,
                                                             size_t cnt, size_t p0, size_t Bp) {
  __shared__ uint4 tile[64 * 33];
  const size_t row0 = (size_t)blockIdx.x * 16;
  const size_t pb = (size_t)blockIdx.y * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int pl = it * 8 + (t >> 5);
    const int q = t & 31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pb + pl < cnt && row0 + (q >> 1) < rows) v = src[((pb + pl) * rows + row0) * 2 + q];
    tile[pl * 33 + q] = v;
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int q = it * 4 + w;
    const size_t row = row0 + (q >> 1);
    if (row < rows) dst[(row * 2 + (q & 1)) * Bp + p0 + pb + lane] = tile[lane * 33 + q];
  }
}

static int transpose_tile(zkmi_ctx* ctx, const void* src_pm, void* dst_bi, size_t rows, size_t cnt,
                          size_t p0, size_t Bp) {
  if (rows == 0 || cnt == 0) return ZKMI_OK;
  hipLaunchKernelGGL(transpose_tile_kernel, dim3((unsigned)((rows + 15) / 16), (unsigned)((cnt + 63) / 64)),
                     dim3(256), 0, ctx->stream, (const uint4*)src_pm, (uint4*)dst_bi, rows, cnt, p0, Bp);
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

static void parallel_copy(void* dst, const void* src, size_t bytes, int threads) {
  if (threads <= 1 || bytes < ((size_t)8 << 20)) {
    memcpy(dst, src, bytes);
    return;
  }
  const size_t part = round_up((bytes + threads - 1) / threads, 4096);
  std::vector<std::thread> pool;
  for (int i = 1; i < threads; i++) {
    const size_t off = (size_t)i * part;
    if (off >= bytes) break;
    pool.emplace_back([=] { memcpy((char*)dst + off, (const char*)src + off, std::min(part, bytes - off)); });
  }
  memcpy(dst, src, std::min(part, bytes));
  for (auto& th : pool) th.join();
}

// One proof-major caller array [batch][rows] (host pageable, host pinned or device) -> batch-inner
// rows of `dst` on ctx->stream.  Pageable memory has been consumed when this returns; pinned and
// device memory must stay valid until the stream has passed the copies (the matching collect).
static int stage_rows(zkmi_ctx* ctx, const void* src, size_t rows, size_t batch, size_t Bp, void* dst) {
  if (rows == 0) return ZKMI_OK;
  const int kind = pointer_kind(src);
  if (kind == PTR_DEVICE) return transpose_tile(ctx, src, dst, rows, batch, 0, Bp);
  // chunks of whole 64-proof column blocks, about 64 MB each
  const size_t per64 = rows * 32 * 64;
  size_t blocks = std::max<size_t>(1, ((size_t)64 << 20) / per64);
  blocks = std::min(blocks, Bp / 64);
  const size_t chunk_proofs = blocks * 64;
  int rc = ring_ensure(ctx, chunk_proofs * rows * 32);
  if (rc) return rc;
  auto& r = ctx->ring;
  const int threads = ctx->copy_threads > 0 ? ctx->copy_threads : 4;
  for (size_t p0 = 0; p0 < batch; p0 += chunk_proofs) {
    const size_t cnt = std::min(chunk_proofs, batch - p0);
    const size_t bytes = cnt * rows * 32;
    const unsigned s = r.next;
    r.next = (r.next + 1) % 3;
    if (r.busy[s]) {
      ZK_HIP(hipEventSynchronize(r.done[s]));
      r.busy[s] = false;
    }
    const char* from = (const char*)src + p0 * rows * 32;
    if (kind == PTR_PINNED) {
      ZK_HIP(hipMemcpyAsync(r.dev[s], from, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
      parallel_copy(r.pinned[s], from, bytes, threads);
      ZK_HIP(hipMemcpyAsync(r.dev[s], r.pinned[s], bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    // the last chunk also zero-fills the padding columns up to Bp
    const size_t cols = std::min(round_up(cnt, 64), Bp - p0);
    hipLaunchKernelGGL(transpose_tile_kernel, dim3((unsigned)((rows + 15) / 16), (unsigned)(cols / 64)),
                       dim3(256), 0, ctx->stream, (const uint4*)r.dev[s], (uint4*)dst, rows, cnt, p0, Bp);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventRecord(r.done[s], ctx->stream));
    r.busy[s] = true;
  }
  return ZKMI_OK;
}

// a = L.w, b = R.w, c = O.w for one constraint row per wavefront, 64 proofs per wavefront; the
// row's terms and coefficients are wave-uniform (scalar loads), the wire values one coalesced
// 2 x 1 KiB access per term.  Coefficients +1 / -1 (most of gnark's boolean and copy rows) skip the
// product.  Everything in gnark's Montgomery image.
struct R1csDev {
  const uint32_t* ptr[3];
  const uint2* terms[3];
  const Fr* coeffs;
  uint32_t n_constraints;
};
// a b 2^-261 on the 9 x 29-bit form (the asm chain of ff29_asm.h, ~365 instructions against ~540 of
// ff.h's mul): with the coefficients stored as 2^261 images (zkmi_r1cs_load lifts them by 2^5) a
// wire in gnark's 2^256 image times a coefficient comes out in the 2^256 image again
__device__ __forceinline__ Fr fmul_261(const Fr& a, const Fr& b) {
  Fr r;
#if defined(__HIP_DEVICE_COMPILE__)
  pack_canonical<Fr29Params>(r.v, mul_asm(unpack29<Fr29Params>(a.v), unpack29<Fr29Params>(b.v)));
#else
  r = a;   // device-only helper
#endif
  return r;
}
__global__ __launch_bounds__(256) void r1cs_eval_kernel(R1csDev m, const Fr* __restrict__ w,
                                                        Fr* __restrict__ a, Fr* __restrict__ b,
                                                        Fr* __restrict__ c, int32_t* __restrict__ st,
                                                        size_t Bp) {
  const size_t p = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  // gridDim.y is capped (65 535 is the hardware limit): a block walks its rows with that stride
  for (uint32_t k = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
       k < m.n_constraints; k += gridDim.y * 4) {
  Fr acc[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    Fr sum = Fr::zero();
    const uint32_t lo = m.ptr[s][k], hi = m.ptr[s][k + 1];
    for (uint32_t j = lo; j < hi; j++) {
      const uint2 tm = m.terms[s][j];
      const Fr x = bi_ld(w, tm.x, p, Bp);
      const uint32_t kind = tm.y >> 30;
      if (kind == 1)
        sum = add(sum, x);
      else if (kind == 2)
        sum = sub(sum, x);
      else
        sum = add(sum, fmul_261(x, m.coeffs[tm.y & 0x3fffffffu]));
    }
    acc[s] = sum;
  }
  bi_st(a, k, p, Bp, acc[0]);
  bi_st(b, k, p, Bp, acc[1]);
  bi_st(c, k, p, Bp, acc[2]);
  if (mul(acc[0], acc[1]) != acc[2]) st[p] = ZKMI_ERR_UNSATISFIED;
  }
}

}  // namespace zk

extern "C" {

void* zkmi_host_alloc(zkmi_ctx* ctx, size_t bytes) {
  if (!ctx || bytes == 0) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "zkmi_host_alloc: hipHostMalloc(" + std::to_string(bytes) + " B) failed";
    return nullptr;
  }
  return p;
}

void zkmi_host_free(zkmi_ctx* ctx, void* p) {
  if (!p) return;
  if (ctx) hipSetDevice(ctx->device);
  hipHostFree(p);
}

int zkmi_set_copy_threads(zkmi_ctx* ctx, int threads) {
  if (!ctx || threads < 0 || threads > 64) return ZKMI_ERR_ARG;
  ctx->copy_threads = threads;
  return ZKMI_OK;
}

void zkmi_r1cs_free(zkmi_ctx* ctx, zkmi_r1cs* m) {
  if (!m) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
  }
  for (int s = 0; s < 3; s++) {
    if (m->ptr[s]) hipFree(m->ptr[s]);
    if (m->terms[s]) hipFree(m->terms[s]);
  }
  if (m->coeffs) hipFree(m->coeffs);
  delete m;
}

int zkmi_r1cs_load(zkmi_ctx* ctx, const zkmi_r1cs_desc* d, zkmi_r1cs** out) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!d || !out) return ZKMI_ERR_ARG;
  *out = nullptr;
  if (d->n_constraints == 0 || d->n_wires == 0 || d->n_coeffs == 0 || d->n_coeffs >= (1u << 30) ||
      !d->coeffs) {
    ctx->err = "r1cs: n_constraints, n_wires and n_coeffs (< 2^30) must be positive";
    return ZKMI_ERR_ARG;
  }
  // coefficient table on the host: which entries are +1 / -1 (Montgomery images of 1 and r - 1)
  std::vector<Fr> coeffs(d->n_coeffs);
  if (hipMemcpy(coeffs.data(), d->coeffs, (size_t)d->n_coeffs * 32, hipMemcpyDefault) != hipSuccess) {
    ctx->err = "r1cs: cannot read the coefficient table";
    return ZKMI_ERR_HIP;
  }
  const Fr one = Fr::one(), minus_one = neg(Fr::one());
  std::vector<uint8_t> kind(d->n_coeffs);
  for (uint32_t i = 0; i < d->n_coeffs; i++) {
    const Fr& x = coeffs[i];
    // canonical range: a value >= r would make the kernel's single conditional subtractions wrong
    bool lt = false;
    for (int j = 7; j >= 0; j--) {
      if (x.v[j] != FrParams::p(j)) {
        lt = x.v[j] < FrParams::p(j);
        break;
      }
    }
    if (!lt) {
      ctx->err = "r1cs: coefficient " + std::to_string(i) + " is not reduced mod r";
      return ZKMI_ERR_ARG;
    }
    kind[i] = x == one ? 1 : x == minus_one ? 2 : 0;
  }
  auto* m = new zkmi_r1cs();
  m->n_wires = d->n_wires;
  m->n_constraints = d->n_constraints;
  m->n_coeffs = d->n_coeffs;
  const uint32_t* ptrs[3] = {d->l_ptr, d->r_ptr, d->o_ptr};
  const zkmi_term* terms[3] = {d->l_terms, d->r_terms, d->o_terms};
  static const char* names[3] = {"L", "R", "O"};
  for (int s = 0; s < 3; s++) {
    std::vector<uint32_t> ptr((size_t)d->n_constraints + 1);
    if (!ptrs[s] || hipMemcpy(ptr.data(), ptrs[s], ptr.size() * 4, hipMemcpyDefault) != hipSuccess) {
      ctx->err = std::string("r1cs: cannot read the row offsets of ") + names[s];
      zkmi_r1cs_free(ctx, m);
      return ZKMI_ERR_ARG;
    }
    bool ok = ptr[0] == 0;
    for (uint32_t k = 0; ok && k < d->n_constraints; k++) ok = ptr[k] <= ptr[k + 1];
    const size_t nnz = ptr[d->n_constraints];
    std::vector<uint2> tm(nnz);
    if (ok && nnz && (!terms[s] || hipMemcpy(tm.data(), terms[s], nnz * 8, hipMemcpyDefault) != hipSuccess))
      ok = false;
    // every wire / coefficient index is range-checked here, before a kernel can use it as an address
    for (size_t j = 0; ok && j < nnz; j++) {
      const uint32_t cid = tm[j].x, wire = tm[j].y;   // zkmi_term {coeff, wire}
      ok = cid < d->n_coeffs && wire < d->n_wires;
      tm[j] = make_uint2(wire, cid | ((uint32_t)kind[cid < d->n_coeffs ? cid : 0] << 30));
    }
    if (!ok) {
      ctx->err = std::string("r1cs: malformed matrix ") + names[s] +
                 " (offsets not monotone from 0, or a term's coefficient / wire index out of range)";
      zkmi_r1cs_free(ctx, m);
      return ZKMI_ERR_ARG;
    }
    m->nnz[s] = nnz;
    if (hipMalloc((void**)&m->ptr[s], ptr.size() * 4) != hipSuccess ||
        hipMemcpy(m->ptr[s], ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void**)&m->terms[s], std::max<size_t>(nnz, 1) * 8) != hipSuccess ||
        (nnz && hipMemcpy(m->terms[s], tm.data(), nnz * 8, hipMemcpyHostToDevice) != hipSuccess)) {
      (void)hipGetLastError();
      ctx->err = "r1cs: device upload failed";
      zkmi_r1cs_free(ctx, m);
      return ZKMI_ERR_HIP;
    }
  }
  for (Fr& x : coeffs)   // 2^256 image -> 2^261 image, what r1cs_eval_kernel's products expect
    for (int t = 0; t < 5; t++) x = add(x, x);
  if (hipMalloc((void**)&m->coeffs, (size_t)d->n_coeffs * 32) != hipSuccess ||
      hipMemcpy(m->coeffs, coeffs.data(), (size_t)d->n_coeffs * 32, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = "r1cs: device upload failed";
    zkmi_r1cs_free(ctx, m);
    return ZKMI_ERR_HIP;
  }
  *out = m;
  return ZKMI_OK;
}

int zkmi_prove_witness_submit(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_r1cs* r1cs,
                              const void* wires, const void* a, const void* b, const void* c,
                              size_t n_constraints, size_t batch, const void* rs) {
  ZK_HIP(hipSetDevice(ctx->device));
  if (!pk || !wires || !rs || batch == 0) {
    ctx->err = "prove_witness_submit: null argument or empty batch";
    return ZKMI_ERR_ARG;
  }
  const size_t n = (size_t)1 << pk->log_n;
  if (r1cs) {
    if (a || b || c) {
      ctx->err = "prove_witness_submit: with an R1CS handle a, b, c are formed on the device: pass NULL";
      return ZKMI_ERR_ARG;
    }
    if (r1cs->n_wires != pk->n_wires) {
      ctx->err = "prove_witness_submit: R1CS and proving key disagree on the number of wires";
      return ZKMI_ERR_ARG;
    }
    if (n_constraints && n_constraints != r1cs->n_constraints) {
      ctx->err = "prove_witness_submit: n_constraints differs from the loaded R1CS";
      return ZKMI_ERR_ARG;
    }
    n_constraints = r1cs->n_constraints;
  } else if (!a || !b || !c) {
    ctx->err = "prove_witness_submit: a, b, c are required without an R1CS handle";
    return ZKMI_ERR_ARG;
  }
  if (n_constraints == 0 || n_constraints > n) {
    ctx->err = "prove_witness_submit: n_constraints must be in [1, 2^log_n]";
    return ZKMI_ERR_ARG;
  }
  const int si = ctx->next_submit;
  zkmi_ctx::ProveSet& S = ctx->sets[si];
  if (S.pending) {
    ctx->err = "prove: two batches already in flight; collect one first";
    return ZKMI_ERR_ARG;
  }
  const size_t Bp = round_up(batch, 64);
  const size_t nw = pk->n_wires;
  int rc;
  const int base = si == 0 ? 0 : 8;
  void* misc;
  // the same scratch slots as zkmi_prove_submit (value file = wire matrix here), so the two entry
  // points share one HBM plan; the idle set is sized too, so steady-state submits never allocate
  const size_t sums_bytes = Bp * (7 * 128 + 2 * 256 + 256 + 256 * (4 * 128 + 256));
  if (!ctx->sets[si ^ 1].pending) {
    void* dummy;
    const int ob = si == 0 ? 8 : 0;
    if ((rc = ensure_scratch(ctx, ob + 0, nw * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 1, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 2, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 3, n * Bp * 32, &dummy)) ||
        (rc = ensure_scratch(ctx, ob + 5, Bp * (96 + 4) + batch * 64, &dummy)) ||
        (rc = ensure_scratch(ctx, si == 0 ? 15 : 14, sums_bytes, &dummy)))
      return rc;
  }
  if ((rc = ensure_scratch(ctx, si == 0 ? 14 : 15, sums_bytes, &S.sums)) ||
      (rc = ensure_scratch(ctx, base + 0, nw * Bp * 32, &S.slots)) ||
      (rc = ensure_scratch(ctx, base + 1, n * Bp * 32, &S.a)) ||
      (rc = ensure_scratch(ctx, base + 2, n * Bp * 32, &S.b)) ||
      (rc = ensure_scratch(ctx, base + 3, n * Bp * 32, &S.c)) ||
      (rc = ensure_scratch(ctx, base + 5, Bp * (96 + 4) + batch * 64, &misc)))
    return rc;
  S.rs = misc;
  S.st = (char*)misc + Bp * 96;
  char* stage_rs = (char*)misc + Bp * 100;
  S.heavy_enqueued = false;
  hipStream_t saved = ctx->stream;
  ctx->stream = ctx->stream2;   // the helpers launch on ctx->stream
  hipEventRecord(S.ev0, ctx->stream);
  rc = stage_rows(ctx, wires, nw, batch, Bp, S.slots);
  if (!rc && !r1cs) {
    rc = stage_rows(ctx, a, n_constraints, batch, Bp, S.a);
    if (!rc) rc = stage_rows(ctx, b, n_constraints, batch, Bp, S.b);
    if (!rc) rc = stage_rows(ctx, c, n_constraints, batch, Bp, S.c);
  }
  if (!rc) {
    const void* rs_dev = rs;
    if (pointer_kind(rs) != PTR_DEVICE) {
      if (hipMemcpyAsync(stage_rs, rs, batch * 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = ZKMI_ERR_HIP;
      rs_dev = stage_rs;
    }
    if (!rc) rc = transpose_in(ctx, rs_dev, S.rs, 2, batch, Bp, 32);
  }
  if (!rc && hipMemsetAsync(S.st, 0, Bp * 4, ctx->stream) != hipSuccess) rc = ZKMI_ERR_HIP;
  if (!rc && r1cs) {
    R1csDev m;
    for (int s = 0; s < 3; s++) {
      m.ptr[s] = r1cs->ptr[s];
      m.terms[s] = r1cs->terms[s];
    }
    m.coeffs = r1cs->coeffs;
    m.n_constraints = r1cs->n_constraints;
    hipLaunchKernelGGL(r1cs_eval_kernel,
                       dim3((unsigned)(Bp / 64), (unsigned)std::min<size_t>((n_constraints + 3) / 4, 32768)),
                       dim3(256), 0, ctx->stream, m, (const Fr*)S.slots, (Fr*)S.a, (Fr*)S.b, (Fr*)S.c,
                       (int32_t*)S.st, Bp);
    if (hipGetLastError() != hipSuccess) rc = ZKMI_ERR_HIP;
  }
  S.batch = batch;
  S.Bp = Bp;
  S.pk = pk;
  S.cs = nullptr;
  S.n_constraints = n_constraints;
  S.f_domain = false;
  // commitment extension: the caller's solver has produced the commitment wires; the proof still
  // needs the Pedersen commitments themselves and (several commitments) the folding challenge
  for (size_t i = 0; !rc && i < pk->commits.size(); i++) rc = commit_phase(ctx, S, (uint32_t)i, false);
  if (!rc && !pk->commits.empty()) rc = commit_finish_submit(ctx, S);
  hipEventRecord(S.ev1, ctx->stream);
  ctx->stream = saved;
  if (rc) {
    if (rc == ZKMI_ERR_HIP && ctx->err.empty()) ctx->err = "prove_witness_submit: HIP error while staging";
    hipStreamSynchronize(ctx->stream2);   // nothing of a failed submit stays queued on caller memory
    return rc;
  }
  S.pending = true;
  ctx->next_submit ^= 1;
  return ZKMI_OK;
}

// The blocking form (round 2's entry point): one submit + collect.
int zkmi_prove_witness_batch(zkmi_ctx* ctx, const zkmi_pk* pk, const void* wires, const void* a,
                             const void* b, const void* c, size_t n_constraints, size_t batch,
                             const void* rs, void* proofs_out) {
  if (batch == 0) return ZKMI_OK;
  if (ctx->sets[0].pending || ctx->sets[1].pending) {
    ctx->err = "prove_witness_batch: submitted batches are still in flight";
    return ZKMI_ERR_ARG;
  }
  if (!proofs_out) {
    ctx->err = "prove_witness_batch: null argument";
    return ZKMI_ERR_ARG;
  }
  int rc = zkmi_prove_witness_submit(ctx, pk, nullptr, wires, a, b, c, n_constraints, batch, rs);
  if (rc) return rc;
  if (!pk->commits.empty()) {
    ctx->err = "prove_witness_batch: this key has commitments: use zkmi_prove_witness_submit + "
               "zkmi_prove_collect_ex";
    return ZKMI_ERR_ARG;
  }
  std::vector<int32_t> status(batch);
  return zkmi_prove_collect(ctx, proofs_out, status.data());
}

}  // extern "C"
