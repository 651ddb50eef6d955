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
//   * OP_ABC copies the three operand values of constraint k into the quotient inputs a, b, c,
//     and for assertion-type constraints checks a*b == c (per-proof status).
// Opcodes: frontend/api.py.
#include <cstdlib>

#include "zkmi_internal.h"

namespace zk {

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
        ST(d, mul(LD(x), LD(y)));
        break;
      case OP_MULC:
        ST(d, mul(LD(x), consts[y]));
        break;
      case OP_ADDC:
        ST(d, add(LD(x), consts[y]));
        break;
      case OP_NEG:
        ST(d, neg(LD(x)));
        break;
      case OP_INV:
        ST(d, inverse(LD(x)));
        break;
      case OP_DIV:
        ST(d, mul(LD(x), inverse(LD(y))));
        break;
      case OP_SETC:
        ST(d, consts[y]);
        break;
      case OP_COPY:
        ST(d, LD(x));
        break;
      case OP_BITS: {
        Fr v = from_mont(LD(x));
        const Fr one = Fr::one(), zero = Fr::zero();
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
        Fr acc = Fr::one();
        for (uint32_t k = 1; k <= n; k++) {
          const uint4 pr = prog[pc + k];
          const Fr v = LD(pr.z);
          ST(pr.y, acc);
          if (!v.is_zero()) acc = mul(acc, v);
        }
        Fr inv = inverse(acc);
        for (uint32_t k = n; k >= 1; k--) {
          const uint4 pr = prog[pc + k];
          const Fr v = LD(pr.z);
          if (v.is_zero()) {
            ST(pr.y, Fr::zero());
          } else {
            const Fr res = mul(inv, LD(pr.y));
            inv = mul(inv, v);
            ST(pr.y, res);
          }
        }
        pc += n;
        break;
      }
      case OP_ABC: {
        const Fr va = LD(d), vb = LD(x), vc = LD(y);
        bi_st(a, k, lane, Bp, va);
        bi_st(b, k, lane, Bp, vb);
        bi_st(c, k, lane, Bp, vc);
        if (ins.x & 0x100u) {
          if (mul(va, vb) != vc) st = ZKMI_ERR_UNSATISFIED;
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
  if (i < Bp) bi_st(row, 0, i, Bp, Fr::one());
}

int solve_bi(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
             size_t Bp) {
  hipLaunchKernelGGL(fill_one_row, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, slots, Bp);
  // ZKMI_SOLVE_BLOCK: lanes per block (64 spreads the 16 waves of a 1024-proof batch over 16 CUs,
  // 256 packs them on 4)
  static const unsigned sb = [] {
    const char* e = getenv("ZKMI_SOLVE_BLOCK");
    const long v = e ? atol(e) : 64;
    return (unsigned)(v == 128 || v == 256 ? v : 64);
  }();
  const unsigned bs = (Bp % sb == 0) ? sb : 64;
  hipLaunchKernelGGL(solve_kernel, dim3((unsigned)(Bp / bs)), dim3(bs), 0, ctx->stream,
                     (const uint4*)cs->program, cs->consts, slots, a, b, c, status, Bp, cs->n_ops);
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

}  // namespace zk
