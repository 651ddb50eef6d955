// Batched R1CS witness solve: S lanes of a wavefront per proof run the circuit's witness program,
// packed by the frontend into steps of independent operations (VLIW; S = 1 is one lane per proof).
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
#include "emul.h"
#include "ff29.h"
#include "ff29_asm.h"
#include "modinv30.h"

namespace zk {

// products / inverses / integer value of F-domain elements kept in the packed canonical image
__device__ __forceinline__ Fr fmul(const Fr& a, const Fr& b) {
  Fr r;
  // the single-accumulator asm chain of ff29_asm.h: ~215 instructions instead of ~260 (the wave is
  // alone on its SIMD: the solve is paced by its instruction count)
  pack_canonical<Fr29Params>(r.v,
                             mul_asm(unpack29<Fr29Params>(a.v), unpack29<Fr29Params>(b.v)));
  return r;
}
__device__ __forceinline__ Fr f_one() {
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = Fr29Params::k261(i);
  return r;
}
// 1 / a in the F domain: a = x 2^261 as a canonical integer, a^-1 = x^-1 2^-261 by the fixed-count
// divsteps of modinv30.h, times 2^783 through the F-domain product (. 2^-261) = x^-1 2^261.
// 0 -> 0 (gnark-crypto Element.Inverse convention).  ~13 000 instructions; the Fermat power it
// replaces (254 squarings + 130 products on the asm chains) was ~75 000: the twisted-Edwards
// gadgets are chains of hundreds of dependent inversions (elgamal-encrypt: 715).
__device__ Fr finv(const Fr& a) {
  Fr t, c;
  modinv30<ModInvFr>(t.v, a.v);
  constexpr uint32_t c783[8] = {0x601fddb2u, 0xbafa616cu, 0x23e89803u, 0x29b4a83eu,
                                0x44d1496bu, 0x917ad601u, 0xfe59ed6du, 0x1baa96fcu};   // 2^783 mod r
#pragma unroll
  for (int i = 0; i < 8; i++) c.v[i] = c783[i];
  return fmul(t, c);
}
__device__ __forceinline__ Fr f_plain(const Fr& a) {  // x * 2^261 -> x as a plain integer
  Fr29 o = Fr29::zero();
  o.v[0] = 1;
  Fr r;
  pack_canonical<Fr29Params>(r.v, mul(unpack29<Fr29Params>(a.v), o));
  return r;
}

// ---- VLIW solver: S sub-lanes of a wavefront per proof -----------------------------------------
// The frontend packs independent operations of one class into steps (frontend/schedule.py); a
// wavefront holds 64 / S proofs, lane l = (sub-lane l / (64 / S), proof l % (64 / S)), and every
// sub-lane runs its own operand quad of the step.  Program rows are (1 + S) quads of 16 bytes,
// staged through LDS in chunks of 32 rows (see the kernel); the operand quad is one ds_read_b128 per
// lane (the header quad is only read by the two one-instruction classes).
//
// Memory order: a step's operands may have been stored by OTHER lanes of the same wavefront in an
// earlier step.  The sub-lanes of a proof are work-items of ONE wavefront, so the release /
// acquire pair they need is a wavefront-scope fence: it pins the compiler's ordering and costs no
// instruction, because the vector memory instructions of a wavefront reach the CU's L1 / the L2 in
// issue order (a workgroup-scope fence here -- s_waitcnt vmcnt(0) after every step -- exposed the
// store latency: 96 -> 7x ms per batch).
enum { CLS_M = 1, CLS_X, CLS_A, CLS_R, CLS_I, CLS_BITS, CLS_BINV, CLS_HIST, CLS_COMMIT, CLS_B_UNUSED,
       CLS_EMUL, CLS_LIMBS };
// CLS_HIST / CLS_COMMIT do not fit the three class bits of an operand quad: their quads carry
// class 0 and the class sits in the header quad.  OP_HIST = 20, OP_HQ = 21, OP_COMMIT = 22
// (frontend/api.py).
enum { OP_HIST = 20, OP_HQ = 21, OP_COMMIT = 22, OP_BXOR = 23, OP_BAND = 24, OP_EMUL = 25 };

// Stores of the hot step classes.  gfx950 reads the data registers of a vector store out of order
// with later VGPR writes, so the compiler waits for a store to complete (s_waitcnt vmcnt) before it
// reuses the registers that held the data -- which the next step's operand loads do at once: the
// store's L2 round trip ended up on every step's critical path.  The data is therefore staged in
// accumulation registers, which nothing else in the kernel touches: bank BANK = a[8 BANK : 8 BANK + 7].
// WAIT (the first store of a step) waits for the PREVIOUS step's stores, issued a whole field
// product earlier, before the bank is overwritten.  The stores are invisible to the compiler's
// vmcnt bookkeeping; the counter completes in issue order, so an untracked older store can only
// make one of its waits longer, never shorter.
template <int BANK, bool WAIT, bool NT>
__device__ __forceinline__ void st_acc(Fr* base, size_t row, size_t b, size_t Bp, const Fr& x) {
  uint4* p0 = reinterpret_cast<uint4*>(base) + row * 2 * Bp + b;
  uint4* p1 = p0 + Bp;
#define ZK_ST_ACC(A0, A1, A2, A3, A4, A5, A6, A7, LO, HI)                                         \
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
  if (NT)                                                                                          \
    asm volatile("v_accvgpr_write_b32 " A0 ", %2\n v_accvgpr_write_b32 " A1 ", %3\n"               \
                 "v_accvgpr_write_b32 " A2 ", %4\n v_accvgpr_write_b32 " A3 ", %5\n"               \
                 "v_accvgpr_write_b32 " A4 ", %6\n v_accvgpr_write_b32 " A5 ", %7\n"               \
                 "v_accvgpr_write_b32 " A6 ", %8\n v_accvgpr_write_b32 " A7 ", %9\n s_nop 1\n"     \
                 "global_store_dwordx4 %0, " LO ", off nt\n global_store_dwordx4 %1, " HI ", off nt" \
                 :: "v"(p0), "v"(p1), "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]),          \
                    "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7])                             \
                 : A0, A1, A2, A3, A4, A5, A6, A7, "memory");                                      \
  else                                                                                             \
    asm volatile("v_accvgpr_write_b32 " A0 ", %2\n v_accvgpr_write_b32 " A1 ", %3\n"               \
                 "v_accvgpr_write_b32 " A2 ", %4\n v_accvgpr_write_b32 " A3 ", %5\n"               \
                 "v_accvgpr_write_b32 " A4 ", %6\n v_accvgpr_write_b32 " A5 ", %7\n"               \
                 "v_accvgpr_write_b32 " A6 ", %8\n v_accvgpr_write_b32 " A7 ", %9\n s_nop 1\n"     \
                 "global_store_dwordx4 %0, " LO ", off\n global_store_dwordx4 %1, " HI ", off"      \
                 :: "v"(p0), "v"(p1), "v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]),          \
                    "v"(x.v[4]), "v"(x.v[5]), "v"(x.v[6]), "v"(x.v[7])                             \
                 : A0, A1, A2, A3, A4, A5, A6, A7, "memory")
  if (BANK == 0) {
    ZK_ST_ACC("a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a[0:3]", "a[4:7]");
  } else if (BANK == 1) {
    ZK_ST_ACC("a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a[8:11]", "a[12:15]");
  } else if (BANK == 2) {
    ZK_ST_ACC("a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a[16:19]", "a[20:23]");
  } else {
    ZK_ST_ACC("a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a[24:27]", "a[28:31]");
  }
#undef ZK_ST_ACC
}

// value-file slot i of this proof, or constant i: one pair of loads from a per-lane address
__device__ __forceinline__ Fr ld_sel(const Fr* slots, const Fr* consts, uint32_t i, bool is_const,
                                     size_t lane, size_t Bp) {
  const uint4* p0 = is_const ? reinterpret_cast<const uint4*>(consts + i)
                             : reinterpret_cast<const uint4*>(slots) + (size_t)i * 2 * Bp + lane;
  const uint4* p1 = p0 + (is_const ? (size_t)1 : Bp);
  const uint4 lo = *p0, hi = *p1;
  Fr r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}

template <int S, bool EMUL>
__global__ __launch_bounds__(64) void solve_vliw_kernel(const uint4* __restrict__ prog,
                                                        const Fr* __restrict__ consts, Fr* slots,
                                                        Fr* __restrict__ a, Fr* __restrict__ b,
                                                        Fr* __restrict__ c,
                                                        int32_t* __restrict__ status, size_t Bp,
                                                        uint32_t n_rows, uint32_t row_begin,
                                                        uint32_t row_end) {
  constexpr int PPW = 64 / S;   // proofs per wavefront
  const uint32_t sl = threadIdx.x / PPW;   // sub-lane
  const size_t lane = (size_t)blockIdx.x * PPW + (threadIdx.x % PPW);   // proof
  int32_t st = 0;
#define LD(i) bi_ld(slots, (i), lane, Bp)
#define ST(i, v) bi_st(slots, (i), lane, Bp, (v))
#define STA(i, v) st_acc<0, true, false>(slots, (i), lane, Bp, (v))
#define LDSEL(i, isc) ld_sel(slots, consts, (i), (isc), lane, Bp)
  // Program rows reach the lanes through LDS, a chunk of CH rows at a time: the next chunk is
  // loaded into registers while the current one runs and written to LDS at the chunk boundary.
  // With the operand quad read by a vector load one step ahead, the wait for it at the top of every
  // step was s_waitcnt vmcnt(0) -- the step counts of the switch arms differ, so the compiler cannot
  // name the load -- which also waits for the previous step's STORES to be acknowledged before the
  // operand loads can even be issued: one L2 round trip per step on the critical path.  LDS reads
  // count on lgkmcnt, so the operand loads now follow the stores back to back.
  // (fewer rows per chunk for 32 and 64 sub-lanes: the chunk stays ~520 quads, nine prefetch registers)
  constexpr uint32_t CH = S > 32 ? 8 : S > 16 ? 16 : 32, RQ = 1 + S, CQ = CH * RQ, NX = (CQ + 63) / 64;
  __shared__ uint4 pq[CQ];
  const uint32_t total_q = n_rows * RQ;
  uint4 nx[NX];
  uint32_t cur = 0xffffffffu, pre = 0xffffffffu;   // chunk in LDS / chunk in nx (wave-uniform)
  for (uint32_t r = row_begin; r < row_end; r++) {
    const uint32_t ci = r / CH;
    if (ci != cur) {
      if (ci != pre) {
#pragma unroll
        for (uint32_t i = 0; i < NX; i++) {
          const uint32_t o = i * 64 + threadIdx.x, gq = ci * CQ + o;
          nx[i] = (o < CQ && gq < total_q) ? prog[gq] : make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (uint32_t i = 0; i < NX; i++) {
        const uint32_t o = i * 64 + threadIdx.x;
        if (o < CQ) pq[o] = nx[i];
      }
      cur = ci;
      pre = ci + 1;
#pragma unroll
      for (uint32_t i = 0; i < NX; i++) {
        const uint32_t o = i * 64 + threadIdx.x, gq = pre * CQ + o;
        nx[i] = (o < CQ && gq < total_q) ? prog[gq] : make_uint4(0, 0, 0, 0);
      }
      __syncthreads();   // one wavefront per workgroup: orders the LDS writes before the reads
    }
    const uint32_t rr = r - ci * CH;
    const uint4 q = pq[(r - ci * CH) * RQ + 1 + sl];
    // the step's class rides in every operand quad (also the idle ones): no dependent scalar load
    // of the header on the critical path
    const uint32_t cls = __builtin_amdgcn_readfirstlane((q.x >> 6) & 7u);
    const uint32_t op = q.x & 0x1fu, k = q.x >> 9;
    const uint32_t d = q.y, x = q.z, y = q.w;
    switch (cls) {
      case CLS_M:
        if (op != OP_END) {
          // Operand fetches have no divergent alternatives (constant table or value file, addend
          // or none): both sides of a divergent if would land in the same registers, and the
          // compiler then waits with vmcnt(0) -- for the previous step's stores too -- between them.
          const bool kc = op == OP_MULC || op == OP_FMAC, fma = op == OP_FMA || op == OP_FMAC;
          const Fr va = LD(x);
          const Fr vb = LDSEL(y, kc);
          // fused multiply-add (frontend/relin.py): the addend's slot rides in the row-index bits
          const Fr vz = LD(fma ? k : x);
          Fr vc = fmul(va, vb);
          if (fma) vc = add(vc, vz);
          STA(d, vc);
          if (op == OP_MULABC) {
            st_acc<1, false, true>(a, k, lane, Bp, va);
            st_acc<2, false, true>(b, k, lane, Bp, vb);
            st_acc<3, false, true>(c, k, lane, Bp, vc);
          }
        }
        break;
      case CLS_X:
        if (op != OP_END) {   // d = x xor y = x + y - 2xy and the row (2x, y, 2xy)
          const Fr va = LD(x), vb = LD(y);
          const Fr ab = fmul(va, vb);
          const Fr ab2 = add(ab, ab);
          STA(d, sub(add(va, vb), ab2));
          if (op == OP_XORABC) {   // OP_XOR: value only (PLONK lowering: rows come from OP_ABC)
            st_acc<1, false, true>(a, k, lane, Bp, add(va, va));
            st_acc<2, false, true>(b, k, lane, Bp, vb);
            st_acc<3, false, true>(c, k, lane, Bp, ab2);
          }
        }
        break;
      case CLS_A:
        if (op != OP_END) {
          const bool kc = op == OP_SETC || op == OP_ADDC;
          const bool two = kc || op == OP_ADD || op == OP_SUB;
          const Fr va = LD(x);                      // OP_SETC: x = 0, the constant-one slot
          const Fr vb = LDSEL(two ? y : x, kc);
          Fr v;
          if (op == OP_SETC)
            v = vb;
          else if (op == OP_ADD || op == OP_ADDC)
            v = add(va, vb);
          else if (op == OP_SUB)
            v = sub(va, vb);
          else if (op == OP_NEG)
            v = neg(va);
          else
            v = va;   // OP_COPY
          STA(d, v);
        }
        break;
      case CLS_R:
        if (op != OP_END) {
          const Fr va = LD(d), vb = LD(x), vc = LD(y);
          st_acc<1, true, true>(a, k, lane, Bp, va);
          st_acc<2, false, true>(b, k, lane, Bp, vb);
          st_acc<3, false, true>(c, k, lane, Bp, vc);
          if ((q.x & 0x20u) && fmul(va, vb) != vc) st = ZKMI_ERR_UNSATISFIED;
        }
        break;
      case CLS_I:
        // the frontend never mixes the two kinds in one step (schedule.py: CLS_B), so the branch is
        // wave-uniform up to idle sub-lanes
        if (op == OP_BXOR || op == OP_BAND) {
          // byte-op hints of the lookup tables: both operands as plain integers (low word), XOR / AND,
          // back into the F domain (plain m times 2^522 through the F-domain product)
          constexpr uint32_t c522[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u,
                                        0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};
          Fr k522;
#pragma unroll
          for (int t = 0; t < 8; t++) k522.v[t] = c522[t];
          const Fr pa = f_plain(LD(x)), pb = f_plain(LD(y));
          Fr pm = Fr::zero();
          pm.v[0] = op == OP_BXOR ? (pa.v[0] ^ pb.v[0]) : (pa.v[0] & pb.v[0]);
          STA(d, fmul(pm, k522));
        } else if (op != OP_END) {
          const Fr va = LD(x), vy = LD(op == OP_DIV ? y : x);
          STA(d, op == OP_DIV ? fmul(va, finv(vy)) : finv(va));
        }
        break;
      case CLS_BITS: {
        // one instruction, quad of sub-lane 0: (dst0, src, n bits); the sub-lanes split the bits
        // q0.w = count | width << 16: `count` limbs of `width` bits (width 0 / 1: bits)
        const uint4 q0 = prog[(size_t)r * (1 + S) + 1];
        const uint32_t n = q0.w & 0xffffu, wd = __builtin_amdgcn_readfirstlane(q0.w >> 16);
        const uint32_t per = (n + S - 1) / S;
        const uint32_t i0 = sl * per, i1 = i0 + per < n ? i0 + per : n;
        const Fr v = f_plain(LD(q0.z));
        const Fr one = f_one(), zero = Fr::zero();
        if (wd <= 1) {
          for (uint32_t i = i0; i < i1; i++) {
            const uint32_t bit = i < 256 ? (v.v[i >> 5] >> (i & 31)) & 1u : 0u;
            ST(q0.y + i, bit ? one : zero);
          }
        } else {
          // limb i = bits [i wd, (i + 1) wd) of the integer (wd <= 16), as the F-domain image of
          // that small integer: plain m times 2^522 through the F-domain product (. 2^-261)
          constexpr uint32_t c522[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u,
                                        0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};   // 2^522 mod r
          Fr k522;
#pragma unroll
          for (int t = 0; t < 8; t++) k522.v[t] = c522[t];
          for (uint32_t i = i0; i < i1; i++) {
            const uint32_t pos = i * wd;
            uint32_t m = 0;
            if (pos < 256) {
              const uint32_t lo = v.v[pos >> 5], hi = (pos >> 5) < 7 ? v.v[(pos >> 5) + 1] : 0u;
              const uint64_t two = ((uint64_t)hi << 32) | lo;
              m = (uint32_t)(two >> (pos & 31)) & ((1u << wd) - 1u);
            }
            Fr pm = Fr::zero();
            pm.v[0] = m;
            ST(q0.y + i, fmul(pm, k522));
          }
        }
        // compiler-visible stores: complete them here, or the register-reuse waits they force
        // (see st_acc) would reappear at the top of every following step
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        break;
      }
      case CLS_BINV: {
        // hdr.y pairs in the hdr.z rows that follow, S pairs per row: every sub-lane inverts its
        // own column of pairs with one field inversion (Montgomery's trick); dst rows double as
        // the prefix-product scratch; dst and src slots are distinct wires
        const uint4 hdr = prog[(size_t)r * (1 + S)];
        const uint32_t nrows = __builtin_amdgcn_readfirstlane(hdr.z);
        const uint4* pr = prog + (size_t)(r + 1) * (1 + S) + 1 + sl;
        Fr acc = f_one();
        for (uint32_t t = 0; t < nrows; t++) {
          const uint4 p = pr[(size_t)t * (1 + S)];
          if ((p.x & 0x1fu) != OP_PAIR) continue;
          const Fr v = LD(p.z);
          ST(p.y, acc);
          if (!v.is_zero()) acc = fmul(acc, v);
        }
        Fr inv = finv(acc);
        for (uint32_t t = nrows; t-- > 0;) {
          const uint4 p = pr[(size_t)t * (1 + S)];
          if ((p.x & 0x1fu) != OP_PAIR) continue;
          const Fr v = LD(p.z);
          if (v.is_zero()) {
            ST(p.y, Fr::zero());
          } else {
            const Fr res = fmul(inv, LD(p.y));
            inv = fmul(inv, v);
            ST(p.y, res);
          }
        }
        r += nrows;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as after CLS_BITS
        break;
      }
      default: {
        // class 0: the header quad names the class.  CLS_HIST: multiplicities of the table
        // 0 .. size - 1 among hdr.y queries (the hdr.z rows that follow, S queries per row) into
        // the consecutive wires starting at slot q0.y.  The sub-lanes zero the counters, then every
        // sub-lane walks its own column of the query rows and counts with an atomic add on the
        // low word of the counter's slot (the sub-lanes of a proof share the counters; a query
        // outside the table counts nowhere), then the sub-lanes turn the integers into field
        // elements in place.  CLS_COMMIT rows are never executed: the host ends a launch in
        // front of them.
        const uint4 hdr = prog[(size_t)r * (1 + S)];
        if ((hdr.x & 0xffu) == CLS_HIST && !(hdr.x & 0x100u)) {
          const uint4 q0 = prog[(size_t)r * (1 + S) + 1];
          const uint32_t nrows = __builtin_amdgcn_readfirstlane(hdr.z);
          const uint32_t size = __builtin_amdgcn_readfirstlane(hdr.w);
          const Fr zero = Fr::zero();
          for (uint32_t j = sl; j < size; j += S) ST(q0.y + j, zero);
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
          const uint4* pr = prog + (size_t)(r + 1) * (1 + S) + 1 + sl;
          for (uint32_t t = 0; t < nrows; t++) {
            const uint4 p = pr[(size_t)t * (1 + S)];
            if ((p.x & 0x1fu) != OP_HQ) continue;
            const Fr v = f_plain(LD(p.z));
            if ((v.v[1] | v.v[2] | v.v[3] | v.v[4] | v.v[5] | v.v[6] | v.v[7]) == 0 && v.v[0] < size) {
              uint32_t* cnt = reinterpret_cast<uint32_t*>(
                  reinterpret_cast<uint4*>(slots) + (size_t)(q0.y + v.v[0]) * 2 * Bp + lane);
              __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
          constexpr uint32_t c522h[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u,
                                         0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};   // 2^522 mod r
          Fr k522h;
#pragma unroll
          for (int t = 0; t < 8; t++) k522h.v[t] = c522h[t];
          for (uint32_t j = sl; j < size; j += S) {
            uint32_t* cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(slots) +
                                                        (size_t)(q0.y + j) * 2 * Bp + lane);
            Fr pm = Fr::zero();
            pm.v[0] = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ST(q0.y + j, fmul(pm, k522h));
          }
          r += nrows;
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        // CLS_LIMBS: up to S short decompositions in one step (the limb hints of the range checker:
        // a handful of limbs each; CLS_BITS splits ONE long decomposition over the sub-lanes instead).
        // Every sub-lane's quad is (OP_BITS, first slot, source slot, count | width << 16), count <= 16.
        if ((hdr.x & 0xffu) == CLS_LIMBS) {
          if (op == OP_BITS) {
            const uint32_t n = y & 0xffffu, wd = y >> 16;
            const Fr v = f_plain(LD(x));
            constexpr uint32_t c522l[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u,
                                           0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};   // 2^522 mod r
            Fr k522l;
#pragma unroll
            for (int t = 0; t < 8; t++) k522l.v[t] = c522l[t];
            const uint32_t w1 = wd ? wd : 1u;
            for (uint32_t i = 0; i < n; i++) {
              const uint32_t pos = i * w1;
              uint32_t m = 0;
              if (pos < 256) {
                const uint32_t lo = v.v[pos >> 5], hi = (pos >> 5) < 7 ? v.v[(pos >> 5) + 1] : 0u;
                const uint64_t two = ((uint64_t)hi << 32) | lo;
                m = (uint32_t)(two >> (pos & 31)) & ((1u << w1) - 1u);
              }
              Fr pm = Fr::zero();
              pm.v[0] = m;
              ST(d + i, fmul(pm, k522l));
            }
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as after CLS_BITS
        }
        if constexpr (EMUL) {
          // CLS_EMUL: header (class, na + nb, n rows, aux), quad 0 = (OP_EMUL, first slot, 0, aux),
          // aux = nout | na << 8 | first modulus constant << 12; the rows that follow hold the limb
          // slots of a, then of b.  a = sum a_i 2^(64 i) (a limb may exceed 64 bits), b likewise,
          // p = the four 64-bit constants: nout - 4 limbs of floor(a b / p), then four of a b mod p,
          // into consecutive wires.  Sub-lane 0 works, as in CLS_HIST.
          if ((hdr.x & 0xffu) == CLS_EMUL && !(hdr.x & 0x100u)) {
            const uint4 q0 = prog[(size_t)r * (1 + S) + 1];
            const uint32_t nq = __builtin_amdgcn_readfirstlane(hdr.y);
            const uint32_t nrows = __builtin_amdgcn_readfirstlane(hdr.z);
            const uint32_t aux = __builtin_amdgcn_readfirstlane(hdr.w);
            const uint32_t nk = (aux & 0xffu) - 4, na = (aux >> 8) & 0xfu, c0 = aux >> 12;
            if (sl == 0) {
              uint32_t A[12], B[12], T[24], Rm[9], P[8];
#pragma unroll
              for (int i = 0; i < 12; i++) A[i] = B[i] = 0;
              const uint4* pr = prog + (size_t)(r + 1) * (1 + S) + 1;
              for (uint32_t t = 0; t < nq; t++) {
                const uint4 p = pr[(size_t)(t / S) * (1 + S) + (t % S)];
                if ((p.x & 0x1fu) != OP_HQ) continue;
                const Fr v = f_plain(LD(p.z));
                if (t < na)
                  emul_acc_at(A, v.v, t);
                else
                  emul_acc_at(B, v.v, t - na);
              }
#pragma unroll
              for (int i = 0; i < 4; i++) {
                const Fr v = f_plain(consts[c0 + i]);
                P[2 * i] = v.v[0];
                P[2 * i + 1] = v.v[1];
              }
              emul_mul(T, A, B);
              emul_divmod(T, Rm, P);
              constexpr uint32_t c522[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u,
                                            0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};   // 2^522 mod r
              Fr k522;
#pragma unroll
              for (int t = 0; t < 8; t++) k522.v[t] = c522[t];
#pragma unroll
              for (int j = 0; j < 8; j++) {
                if ((uint32_t)j < nk) {
                  Fr pm = Fr::zero();
                  pm.v[0] = T[2 * j];
                  pm.v[1] = T[2 * j + 1];
                  ST(q0.y + j, fmul(pm, k522));
                }
              }
#pragma unroll
              for (int j = 0; j < 4; j++) {
                Fr pm = Fr::zero();
                pm.v[0] = Rm[2 * j];
                pm.v[1] = Rm[2 * j + 1];
                ST(q0.y + nk + j, fmul(pm, k522));
              }
            }
            r += nrows;
            __builtin_amdgcn_s_waitcnt(0x0F70);
          }
        }
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
#undef LD
#undef ST
#undef STA
#undef LDSEL
  // a proof is unsatisfied if any of its sub-lanes saw a failing row
  if (st) atomicMin(&status[lane], st);
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

__global__ void zero_status(int32_t* st, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st[i] = 0;
}

int solve_init(zkmi_ctx* ctx, Fr* slots, int32_t* status, size_t Bp) {
  hipLaunchKernelGGL(fill_one_row, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, slots, Bp);
  hipLaunchKernelGGL(zero_status, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, status, Bp);
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

int solve_bi(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
             size_t Bp) {
  int rc = solve_init(ctx, slots, status, Bp);
  if (rc) return rc;
  return solve_rows(ctx, cs, slots, a, b, c, status, Bp, 0, cs->n_rows);
}

int solve_rows(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
               size_t Bp, uint32_t row_begin, uint32_t row_end) {
  if (row_begin >= row_end) return ZKMI_OK;
  // one wavefront per 64 / S proofs
  const uint32_t S = cs->lanes_per_proof;
  const dim3 grid((unsigned)(Bp * S / 64)), block(64);
  const uint4* prog = (const uint4*)cs->program;
#define ZK_SOLVE(SS)                                                                          \
  if (cs->has_emul)                                                                               \
    hipLaunchKernelGGL((solve_vliw_kernel<SS, true>), grid, block, 0, ctx->stream, prog,          \
                       cs->consts, slots, a, b, c, status, Bp, cs->n_rows, row_begin, row_end);   \
  else                                                                                            \
    hipLaunchKernelGGL((solve_vliw_kernel<SS, false>), grid, block, 0, ctx->stream, prog,         \
                       cs->consts, slots, a, b, c, status, Bp, cs->n_rows, row_begin, row_end)
  switch (S) {
    case 1: ZK_SOLVE(1); break;
    case 2: ZK_SOLVE(2); break;
    case 4: ZK_SOLVE(4); break;
    case 8: ZK_SOLVE(8); break;
    case 16: ZK_SOLVE(16); break;
    case 32: ZK_SOLVE(32); break;
    case 64: ZK_SOLVE(64); break;
    default:
      ctx->err = "cs: lanes_per_proof must be a power of two, 1 .. 64";
      return ZKMI_ERR_ARG;
  }
#undef ZK_SOLVE
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

}  // namespace zk
