// Batched R1CS witness solve: one lane per proof runs the circuit's straight-line witness program.
//
// Replaces cs.R1CS.Solve (gnark constraint/bn254) [UPSTREAM-RECALL, SURVEY.md §3.2 step 1].  gnark
// walks the constraints of ONE witness, solving the single unknown wire of each from its linear
// expressions and calling hints (bits.NBits, solver.InvZeroHint) as big.Int callbacks.  The wire
// values are determined by the constraint system, so any correct solver produces the same vector.
//
// Here the frontend has already recorded, per API call, the field operation that produces its
// value (frontend/api.py), so solving is the evaluation of a fixed instruction stream over a
// value file [n_slots][Bp]:
//   * instruction words are wave-uniform -> scalar loads; control flow never diverges;
//   * lane = proof, so every slot access is a coalesced 2 KiB row segment;
//   * slot i < n_wires IS wire i: the full wire matrix the MSMs read is produced in place;
//   * OP_MULABC / OP_XORABC compute a product / a boolean XOR wire AND emit its constraint row
//     (the two shapes almost every constraint of the gadget circuits has): one op, two loads.
//   * OP_ABC copies the three operand values of constraint k into the quotient inputs a, b, c,
//     and for assertion-type constraints checks a*b == c (per-proof status).
// Opcodes: frontend/api.py.
//
// Value domain.  88 % of the solver's issue slots are field products (96 872 per proof at
// Arbo-160).  The value file therefore holds x * 2^261 mod r (canonical, packed 8 x u32: the "F
// domain" of ff29.h) instead of gnark's x * 2^256: additions and subtractions are the same carry
// chains either way, products run on the 9 x 29-bit representation (257 instructions instead of
// ~540).  Inputs and constants are converted when they are staged (one product by 2^5 each); the
// consumers convert for free: the MSMs fold 2^-5 into their Montgomery-to-integer product, the
// quotient's first pass skips its own 2^266 conversion.
#include <cstdlib>

#include "zkmi_internal.h"
#include "ff29.h"

namespace zk {

// products / inverses / integer value of F-domain elements kept in the packed canonical image
__device__ __forceinline__ Fr fmul(const Fr& a, const Fr& b) {
  Fr r;
  pack_canonical<Fr29Params>(r.v,
                             mul_ilp(unpack29<Fr29Params>(a.v), unpack29<Fr29Params>(b.v)));
  return r;
}
__device__ __forceinline__ Fr f_one() {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = Fr29Params::k261(i);
  return r;
}
__device__ Fr finv(const Fr& a) {  // a^(r-2): 0 -> 0 (gnark-crypto Element.Inverse convention)
  const Fr29 A = unpack29<Fr29Params>(a.v);
  Fr29 R = Fr29::one();
  for (int i = 7; i >= 0; i--) {
    uint32_t e = FrParams::p(i);
    if (i == 0) e -= 2;
    for (int b = 31; b >= 0; b--) {
      R = mul_ilp(R, R);
      if ((e >> b) & 1) R = mul_ilp(R, A);
    }
  }
  Fr r;
  pack_canonical<Fr29Params>(r.v, R);
  return r;
}
__device__ __forceinline__ Fr f_plain(const Fr& a) {  // x * 2^261 -> x as a plain integer
  Fr29 o = Fr29::zero();
  o.v[0] = 1;
  Fr r;
  pack_canonical<Fr29Params>(r.v, mul(unpack29<Fr29Params>(a.v), o));
  return r;
}

__global__ __launch_bounds__(256) void solve_kernel(const uint4* __restrict__ prog,
                                                   const Fr* __restrict__ consts, Fr* slots,
                                                   Fr* __restrict__ a, Fr* __restrict__ b,
                                                   Fr* __restrict__ c, int32_t* __restrict__ status,
                                                   size_t Bp, uint32_t n_ops) {
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t k = 0;
  int32_t st = 0;
#define LD(i) bi_ld(slots, (i), lane, Bp)
#define ST(i, v) bi_st(slots, (i), lane, Bp, (v))
  for (uint32_t pc = 0; pc < n_ops; pc++) {
    const uint4 ins = prog[pc];
    const uint32_t op = ins.x & 0xffu;
    const uint32_t d = ins.y, x = ins.z, y = ins.w;
    switch (op) {
      case OP_ADD:
        ST(d, add(LD(x), LD(y)));
        break;
      case OP_SUB:
        ST(d, sub(LD(x), LD(y)));
        break;
      case OP_MUL:
        ST(d, fmul(LD(x), LD(y)));
        break;
      case OP_MULC:
        ST(d, fmul(LD(x), consts[y]));
        break;
      case OP_ADDC:
        ST(d, add(LD(x), consts[y]));
        break;
      case OP_NEG:
        ST(d, neg(LD(x)));
        break;
      case OP_INV:
        ST(d, finv(LD(x)));
        break;
      case OP_DIV:
        ST(d, fmul(LD(x), finv(LD(y))));
        break;
      case OP_SETC:
        ST(d, consts[y]);
        break;
      case OP_COPY:
        ST(d, LD(x));
        break;
      case OP_BITS: {
        Fr v = f_plain(LD(x));
        const Fr one = f_one(), zero = Fr::zero();
        for (uint32_t i = 0; i < y; i++) {
          const uint32_t bit = i < 256 ? (v.v[0] & 1u) : 0u;
#pragma unroll
          for (int l = 0; l < 7; l++) v.v[l] = (v.v[l] >> 1) | (v.v[l + 1] << 31);
          v.v[7] >>= 1;
          ST(d + i, bit ? one : zero);
        }
        break;
      }
      case OP_BATCHINV: {
        // rows pc+1 .. pc+n are (OP_PAIR, dst, src): dst = 1/src (0 for 0) with one inversion.
        // dst rows double as the prefix-product scratch; dst and src slots are distinct wires.
        const uint32_t n = d;
        Fr acc = f_one();
        for (uint32_t k = 1; k <= n; k++) {
          const uint4 pr = prog[pc + k];
          const Fr v = LD(pr.z);
          ST(pr.y, acc);
          if (!v.is_zero()) acc = fmul(acc, v);
        }
        Fr inv = finv(acc);
        for (uint32_t k = n; k >= 1; k--) {
          const uint4 pr = prog[pc + k];
          const Fr v = LD(pr.z);
          if (v.is_zero()) {
            ST(pr.y, Fr::zero());
          } else {
            const Fr res = fmul(inv, LD(pr.y));
            inv = fmul(inv, v);
            ST(pr.y, res);
          }
        }
        pc += n;
        break;
      }
      case OP_MULABC: {  // d = x * y and the row (x, y, d) in one step
        const Fr va = LD(x), vb = LD(y);
        const Fr vc = fmul(va, vb);
        ST(d, vc);
        bi_st_nt(a, k, lane, Bp, va);
        bi_st_nt(b, k, lane, Bp, vb);
        bi_st_nt(c, k, lane, Bp, vc);
        k++;
        break;
      }
      case OP_XORABC: {  // d = x xor y = x + y - 2xy and the row (2x, y, 2xy)
        const Fr va = LD(x), vb = LD(y);
        const Fr ab = fmul(va, vb);
        const Fr ab2 = add(ab, ab);
        ST(d, sub(add(va, vb), ab2));
        bi_st_nt(a, k, lane, Bp, add(va, va));
        bi_st_nt(b, k, lane, Bp, vb);
        bi_st_nt(c, k, lane, Bp, ab2);
        k++;
        break;
      }
      case OP_ABC: {
        const Fr va = LD(d), vb = LD(x), vc = LD(y);
        bi_st_nt(a, k, lane, Bp, va);
        bi_st_nt(b, k, lane, Bp, vb);
        bi_st_nt(c, k, lane, Bp, vc);
        if (ins.x & 0x100u) {
          if (fmul(va, vb) != vc) st = ZKMI_ERR_UNSATISFIED;
        }
        k++;
        break;
      }
      default:
        break;
    }
  }
#undef LD
#undef ST
  status[lane] = st;
}

__global__ void fill_one_row(Fr* row, size_t Bp) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Bp) bi_st(row, 0, i, Bp, f_one());
}

// x <- x * k / 2^256 (ff.h product by a plain-integer constant): k = 2^261 mod r moves gnark's image
// x*2^256 into the F domain, k = 2^251 moves it back.
__global__ __launch_bounds__(256) void scale_rows_kernel(Fr* base, size_t rows, size_t Bp, Fr k) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x, total = rows * Bp;
  for (; i < total; i += stride) {
    const size_t row = i / Bp, l = i % Bp;
    bi_st(base, row, l, Bp, mul(bi_ld(base, row, l, Bp), k));
  }
}
__global__ __launch_bounds__(256) void scale_array_kernel(Fr* a, size_t n, Fr k) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = mul(a[i], k);
}
static Fr k_to_f() {
  Fr k;
  for (int i = 0; i < 8; i++) k.v[i] = Fr29Params::k261(i);
  return k;
}
static Fr k_to_std() {  // 2^251
  Fr k = Fr::zero();
  k.v[7] = 1u << 27;
  return k;
}
int rows_to_f_domain(zkmi_ctx* ctx, Fr* base, size_t rows, size_t Bp) {
  if (rows == 0) return ZKMI_OK;
  size_t g = (rows * Bp + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, base, rows,
                     Bp, k_to_f());
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}
int rows_to_std_domain(zkmi_ctx* ctx, Fr* base, size_t rows, size_t Bp) {
  if (rows == 0) return ZKMI_OK;
  size_t g = (rows * Bp + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, base, rows,
                     Bp, k_to_std());
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}
int array_to_f_domain(zkmi_ctx* ctx, Fr* a, size_t n) {
  if (n == 0) return ZKMI_OK;
  hipLaunchKernelGGL(scale_array_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, a, n, k_to_f());
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

int solve_bi(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
             size_t Bp) {
  hipLaunchKernelGGL(fill_one_row, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, slots, Bp);
  // lanes per block (zkmi_cs_desc.solve_block): 64 spreads the 16 waves of a 1024-proof batch over
  // 16 CUs, 256 packs them on 4
  const unsigned sb = cs->solve_block ? cs->solve_block : 64;
  const unsigned bs = (Bp % sb == 0) ? sb : 64;
  hipLaunchKernelGGL(solve_kernel, dim3((unsigned)(Bp / bs)), dim3(bs), 0, ctx->stream,
                     (const uint4*)cs->program, cs->consts, slots, a, b, c, status, Bp, cs->n_ops);
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

}  // namespace zk
