// Internal declarations shared by the HIP translation units of libzkmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkmi.h"
#include "ec.h"

namespace zk {

// Device layout used everywhere on the hot path: "batch-inner".  A logical matrix of field
// elements [row][proof] is stored with the proof index fastest, so that the 64 lanes of a
// wavefront work on 64 different proofs at the same row: loads/stores are 2 KiB contiguous per
// wave, and everything indexed by row (twiddles, constants, window-table slices, program words)
// is wave-uniform and comes through the scalar unit.  Bp = batch rounded up to 64.
// Within a row the two 16-byte halves of the elements are stored as two planes
// ([row][half][proof]): a lane's 32-byte element is then two dwordx4 accesses that are each
// perfectly coalesced across the wave (1 KiB contiguous).  With the halves interleaved
// ([row][proof][half]) every vector store wrote 16 of each 32 bytes and rocprofv3 showed 3x read /
// 6x write amplification on the NTT passes (profiles/r01_pmc_*.csv).

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct NttPlan {
  int log_n = 0;
  Fr* tw_fwd = nullptr;     // n/2 powers of w
  Fr* tw_inv = nullptr;     // n/2 powers of w^-1
  Fr* coset_fwd = nullptr;  // g^i, n
  Fr* coset_inv = nullptr;  // g^-i / n, n
  Fr n_inv;                 // 1/n
  Fr den;                   // 1/(g^n - 1)
  // the same constants in the 2^261 Montgomery domain of ff29.h (canonical, packed 8 x u32)
  Fr* tw29_fwd = nullptr;
  Fr* tw29_inv = nullptr;
  Fr* coset29_fwd = nullptr;
  Fr* coset29n_fwd = nullptr;  // g^i / n (F domain): scaling between a fused iNTT -> coset NTT
  Fr n_inv29;
  Fr den29;
  Fr ninv_den;  // den / n in gnark's image: uniform post factor of the quotient's c transform
};

}  // namespace zk

struct zkmi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<zk::NttPlan> plans;
  hipEvent_t ev[8] = {};
  double timings[8] = {};
  // scratch arena for the prove pipeline, grown on demand
  zk::DevBuf scratch[24];
  // software pipeline over batches: the latency-bound witness solve of batch k+1 runs on
  // `stream2` (16 wavefronts at batch 1024) underneath the NTT/MSM kernels of batch k.
  // `stream3` runs the 16-wavefront assembly of batch k underneath the quotient kernels of batch k+1.
  hipStream_t stream2 = nullptr, stream3 = nullptr;
  struct ProveSet {
    bool pending = false;
    bool heavy_enqueued = false;   // quotient + MSMs of this batch are already on the main stream
    void* sums = nullptr;          // MSM results, delta multiples, assembled proofs
    hipEvent_t evq[5] = {};        // main stream: start, after NTTs, after G1 MSMs, after G2, done
    hipEvent_t eva[2] = {};        // stream3: assembly start / end
    hipEvent_t msm_ev[8][2] = {};  // around every proving-key msm_accumulate launch
    int msm_ev_group[8] = {};
    int msm_ev_used = 0;
    size_t batch = 0, Bp = 0;
    const zkmi_pk* pk = nullptr;
    const zkmi_cs* cs = nullptr;      // null for zkmi_prove_witness_batch (solved by the caller)
    size_t n_constraints = 0;         // rows of a, b, c that hold data (the rest of the domain is 0)
    bool f_domain = true;             // value file and a, b, c in the solver's 2^261 domain
    void *slots = nullptr, *a = nullptr, *b = nullptr, *c = nullptr, *rs = nullptr, *st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // solve start / end on stream2
    // commitment extension: Pedersen commitments of the batch (n_commitments x Bp G1 affine) and
    // the per-proof powers of the folding challenge (n_commitments x Bp fr, in the value file's
    // domain), written by the submit; proof of knowledge (Bp XYZZ -> affine) by the collect
    void *commit_pts = nullptr, *commit_ch = nullptr, *commit_pok = nullptr, *commit_acc = nullptr;
    std::vector<zk::Fr> commit_host;   // commitment wire values (plain integers), n x batch
  } sets[2];
  int next_submit = 0, next_collect = 0;
  // set whose proving-key MSM launches are currently being bracketed with HIP events (or null)
  int msm_ev_set = -1;
  // deferred MSM tails (msm_run with a finish stream): the chunk partials of consecutive MSMs
  // alternate between two buffers; part_ev[k] = last reduction that read buffer k has finished
  // (recorded on the finish stream), acc_ev[k] = last accumulate that wrote it (main stream)
  hipEvent_t part_ev[2] = {}, acc_ev[2] = {};
  bool part_ev_valid[2] = {false, false};
  unsigned part_next = 0;
  // witness entry (witness.hip): ring of three pinned host chunks + three device chunks through
  // which proof-major host arrays stream in (done[i]: the transpose that consumed chunk i)
  struct StageRing {
    void* pinned[3] = {};
    void* dev[3] = {};
    hipEvent_t done[3] = {};
    bool busy[3] = {false, false, false};
    size_t bytes = 0;
    unsigned next = 0;
  } ring;
  int copy_threads = 0;   // host threads that fill a pinned chunk from pageable memory; 0 = 4
};

// Window plan of a fixed-base table: W windows whose sizes sum to exactly 255 bits (scalars are
// below 2^254, so the top window also absorbs the last signed-digit carry).  `rem` windows of
// (c0 + 1) bits come first, then W - rem windows of c0 bits.  Window j of base i holds the
// 2^(bits_j - 1) positive multiples d * 2^(shift_j) * P_i at entry offset i*per_base + off_j.
struct WinPlan {
  int W = 0;
  uint32_t per_base = 0;   // table entries per base
  uint8_t bits[64] = {};
  uint32_t off[64] = {};
  // shared = 1: ONE table d * P_i, d = 1..2^(c-1), serves every window (off[j] = 0); the windows
  // are accumulated separately and combined with Horner's rule at the end (msm.hip)
  uint8_t shared = 0;
  // comb = k > 0: joint table over groups of k consecutive bases, T[g][m] = sum_{i in m} P_{gk+i}
  // for every non-empty subset m (2^k entries, entry 0 unused), 254 one-bit windows: 254 / k mixed
  // additions per (base, proof).  W = 254; bits/off are not used.
  uint8_t comb = 0;
  // comb_signed: the comb table holds sign patterns, entry e = P_(k-1) + sum_{i<k-1} +-P_i
  // (2^(k-1) entries per group); W = 255: window 254 is the parity correction (msm.hip)
  uint8_t comb_signed = 0;
};

struct zkmi_msm_bases {
  int group = 1;        // 1 = G1, 2 = G2
  size_t n = 0;         // number of bases
  WinPlan plan;
  void* table = nullptr;  // affine entries [(i * W + j) << (c-1) | (d-1)]
  size_t table_bytes = 0;
  uint8_t* inf = nullptr;  // device, n flags: base i is the point at infinity (skipped)
  size_t n_groups = 0;     // comb plans: ceil(n / k)
  int entries_may_be_inf = 0;  // comb plans: some subset of a group sums to the identity
  uint32_t chunk_factor = 0;   // (window, chunk) blocks in flight / wave slots; 0 = default
  zk::G1Affine stotal1 = {};   // signed comb tables: sum of all bases (G1 / G2 by `group`)
  zk::G2Affine stotal2 = {};
  // side = 1: per-window plan run on the second stream (commitment MSMs of a submit, beside the
  // previous batch's MSMs on the main stream): its own partial-sum scratch, no deferred tails.
  // side = 2: the same for the one-base tables of delta, run on the assembly stream
  int side = 0;
};

// one commitment of a key (zkmi_commitment_desc on the device)
struct zkmi_commit_key {
  uint32_t n_private = 0, n_hashed = 0, wire = 0;
  std::vector<uint32_t> hashed;     // host copy (rows read back for the hash)
  uint32_t* private_dev = nullptr;  // device: wire index of every basis point
  zkmi_msm_bases* basis = nullptr;  // side tables
  zkmi_msm_bases* sigma = nullptr;  // main-stream tables (proof of knowledge)
};

struct zkmi_pk {
  uint32_t log_n = 0, n_wires = 0, n_a = 0, n_b = 0, n_k = 0, n_z = 0, max_batch = 1024;
  uint32_t *a_wire = nullptr, *b_wire = nullptr, *k_wire = nullptr;  // device
  zkmi_msm_bases *A = nullptr, *B1 = nullptr, *K = nullptr, *Z = nullptr, *B2 = nullptr;
  zkmi_msm_bases *D1 = nullptr, *D2 = nullptr;  // one-base tables of delta (G1, G2)
  uint32_t* idx3 = nullptr;                      // device {0, 1, 2}
  zk::G1Affine alpha, beta1, delta1;
  zk::G2Affine beta2, delta2;
  std::vector<zkmi_commit_key> commits;   // commitment extension (empty for a plain key)
};

struct zkmi_cs {
  uint32_t n_wires = 0, n_public = 0, n_secret = 0, n_constraints = 0, n_slots = 0, n_rows = 0,
           n_consts = 0, lanes_per_proof = 1;
  uint32_t* program = nullptr;  // device
  zk::Fr* consts = nullptr;     // device
  // COMMIT rows of the program in order: (row index, commitment index); the solver kernel is
  // launched once per segment between them
  std::vector<std::pair<uint32_t, uint32_t>> commit_rows;
  // the program holds OP_EMUL units (std/math/emulated product hints): the solver variant with the
  // big-integer division arm is launched
  bool has_emul = false;
};

namespace zk {

#if defined(__HIPCC__)
// element (row, b) of a batch-inner matrix with row pitch Bp
__device__ __forceinline__ Fr bi_ld(const Fr* base, size_t row, size_t b, size_t Bp) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + row * 2 * Bp + b;
  const uint4 lo = p[0], hi = p[Bp];
  Fr r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
__device__ __forceinline__ void bi_st(Fr* base, size_t row, size_t b, size_t Bp, const Fr& x) {
  uint4* p = reinterpret_cast<uint4*>(base) + row * 2 * Bp + b;
  p[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
  p[Bp] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
// write-once streams (the solver's a, b, c rows): non-temporal, so that they do not evict the MSM
// table lines the kernels running beside the solve are hitting in L2 / Infinity Cache
__device__ __forceinline__ void bi_st_nt(Fr* base, size_t row, size_t b, size_t Bp, const Fr& x) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4* p = reinterpret_cast<u32x4*>(base) + row * 2 * Bp + b;
  u32x4 lo = {x.v[0], x.v[1], x.v[2], x.v[3]}, hi = {x.v[4], x.v[5], x.v[6], x.v[7]};
  __builtin_nontemporal_store(lo, p);
  __builtin_nontemporal_store(hi, p + Bp);
}
#endif

// witness-program opcodes (frontend/api.py)
enum { OP_END = 0, OP_ADD, OP_SUB, OP_MUL, OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_BITS, OP_SETC,
       OP_ABC, OP_COPY, OP_DIV, OP_BATCHINV, OP_PAIR, OP_MULABC, OP_XORABC, OP_XOR, OP_FMAC, OP_FMA };

#define ZK_HIP(call)                                                         \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(_e);          \
      return ZKMI_ERR_HIP;                                                   \
    }                                                                        \
  } while (0)

static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

int ensure_scratch(zkmi_ctx* ctx, int slot, size_t bytes, void** out);
// where a caller's pointer lives (hipPointerGetAttributes)
enum { PTR_PAGEABLE = 0, PTR_PINNED = 1, PTR_DEVICE = 2 };
int pointer_kind(const void* p);
void witness_ring_free(zkmi_ctx* ctx);

// layout conversion (proof-major <-> batch-inner), rows x batch elements of `elem_bytes`
int transpose_in(zkmi_ctx* ctx, const void* src_pm, void* dst_bi, size_t rows, size_t batch,
                 size_t Bp, size_t elem_bytes);
int transpose_out(zkmi_ctx* ctx, const void* src_bi, void* dst_pm, size_t rows, size_t batch,
                  size_t Bp, size_t elem_bytes);

// ntt.hip
int get_plan(zkmi_ctx* ctx, int log_n, NttPlan** out);
// src -> dst (distinct buffers), batch-inner [n][Bp]; rows >= n_valid of src are read as zero
int ntt_bi(zkmi_ctx* ctx, const NttPlan* plan, const Fr* src, Fr* dst, size_t Bp, bool inverse,
           bool coset, size_t n_valid);
// h = (a*b - c) * den, elementwise over n*Bp
int pointwise_h(zkmi_ctx* ctx, const NttPlan* plan, const Fr* a, const Fr* b, const Fr* c, Fr* h,
                size_t Bp);
// full quotient: a,b,c batch-inner [n][Bp] (rows >= n_valid zero) -> h in `a_out`; t0,t1 scratch
// abc_f: a, b, c are in the F domain (solver output); the result h is always in gnark's image
int compute_h_bi(zkmi_ctx* ctx, const NttPlan* plan, Fr* a, Fr* b, Fr* c, Fr* t0, size_t Bp,
                 size_t n_valid, Fr** h_out, bool abc_f = false);

// msm.hip
// smallest number of windows whose table for n_total bases fits `budget_bytes`
WinPlan plan_windows_for_budget(size_t n_total, int group, double budget_bytes);
WinPlan plan_uniform(int c);
WinPlan plan_with_windows(int W);
int msm_bases_build(zkmi_ctx* ctx, int group, const void* bases_dev, size_t n, const WinPlan& plan,
                    zkmi_msm_bases** out);
// scalars batch-inner: element (row, b) at scalars[row * Bp + b]; row_idx (device, may be null)
// maps base i to its scalar row.  out_xyzz: Bp accumulators.
// scalars_f: the scalars are in the F domain (solver output) instead of gnark's image
WinPlan plan_shared(int c);
WinPlan plan_comb(int k, bool signed_tables = true);
void plan_comb_for_budget(size_t n1, size_t n2, double usable_bytes, int* k1, int* k2, bool* sg1,
                          bool* sg2, bool allow_signed = true);
void plan_shared_for_budget(size_t n1, size_t n2, double usable_bytes, int* c1, int* c2);
int msm_run(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const Fr* scalars, const uint32_t* row_idx,
            size_t Bp, void* out_xyzz, bool scalars_f = false, void* wsum_out = nullptr,
            hipStream_t finish_stream = nullptr);
// With wsum_out (shared-table plans only) msm_run stops at the W window sums ([W][Bp] XYZZ) and the
// caller finishes with msm_horner_run -- 255 dependent doublings per proof, latency-bound, which
// the prover runs on its assembly stream under the next batch's kernels.  With finish_stream the
// sums over the chunk partials run there too (ordered after the accumulate launch by an event; the
// partials of consecutive MSMs alternate between two buffers), so the main stream holds nothing
// but the digit pass and the accumulate kernel.
int msm_horner_run(zkmi_ctx* ctx, hipStream_t stream, int count,
                   const zkmi_msm_bases* const* bases, void* const* wsums, void* const* outs,
                   size_t Bp);

int xyzz_to_affine(zkmi_ctx* ctx, int group, const void* in, void* out, size_t n);

// solve.hip
// slots [n_slots][Bp] with rows 0..n_inputs pre-filled; writes every wire row, a/b/c rows
// [0, n_constraints) and status[Bp] (0 or ZKMI_ERR_UNSATISFIED)
// The solver's value file and its a/b/c outputs are in the F domain (x * 2^261, see solve.hip);
// inputs must be converted with rows_to_f_domain after staging.
int solve_bi(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
             size_t Bp);
// the two halves of solve_bi for programs with COMMIT rows: initialise ONE row and status, then run
// program rows [row_begin, row_end)
int solve_init(zkmi_ctx* ctx, Fr* slots, int32_t* status, size_t Bp);
int solve_rows(zkmi_ctx* ctx, const zkmi_cs* cs, Fr* slots, Fr* a, Fr* b, Fr* c, int32_t* status,
               size_t Bp, uint32_t row_begin, uint32_t row_end);

// commit.hip: commitment extension of the prover
// the commitments of set S at a COMMIT row of the solver path (or all of them, witness path):
// MSM over the key's basis from the value file, hash_to_field on the host, challenge into the wire
int commit_phase(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S, uint32_t index, bool write_wire);
// folding challenge powers for the proof of knowledge (after the last commit_phase)
int commit_finish_submit(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S);
// proof of knowledge on ctx->stream (main stream, enqueue_heavy)
int commit_pok(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S);
int commit_keys_load(zkmi_ctx* ctx, const zkmi_pk_desc* d, zkmi_pk* pk);
void commit_keys_free(zkmi_ctx* ctx, zkmi_pk* pk);
int rows_to_f_domain(zkmi_ctx* ctx, Fr* base, size_t rows, size_t Bp);
int rows_to_std_domain(zkmi_ctx* ctx, Fr* base, size_t rows, size_t Bp);
int array_to_f_domain(zkmi_ctx* ctx, Fr* a, size_t n);

}  // namespace zk
