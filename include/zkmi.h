/* zkmi.h -- C-ABI of the MI355X-native Groth16/BN254 prover hot path (libzkmi.so).
 *
 * The reference (vocdoni/gnark-crypto-primitives) has no FFI boundary of its own: its circuits
 * reach the prover only through gnark (go.mod:8-9) from test call sites such as
 * tree/test/verifier_bn254_test.go:41 (frontend.Compile) and :67 (assert.SolvingSucceeded ->
 * solver; groth16.Prove under the prover_checks build tag).  Each entry point below names the
 * gnark / gnark-crypto interface it stands in for [UPSTREAM-RECALL, SURVEY.md §3.2, §8b]; the cgo
 * stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions (chosen so a cgo caller passes unsafe.Pointer(&slice[0]) with zero copies):
 *   - fr.Element / fp.Element: 4 x uint64 little-endian limbs, Montgomery form, R = 2^256.
 *   - G1Affine = {X, Y} 64 B; G2Affine = {X.A0, X.A1, Y.A0, Y.A1} 128 B; infinity = all zero.
 *   - Every pointer argument may be a host pointer or a HIP device pointer (detected with
 *     hipPointerGetAttributes); device pointers avoid the PCIe copy.
 *   - "batch" arrays are proof-major: element (proof p, index i) at [p * stride + i].
 *   - All functions return 0 on success or a negative zkmi_status; they never abort.  A context
 *     is thread-compatible (one caller at a time); distinct contexts are independent.
 */
#ifndef ZKMI_H
#define ZKMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkmi_ctx zkmi_ctx;
typedef struct zkmi_pk zkmi_pk;
typedef struct zkmi_cs zkmi_cs;

enum zkmi_status {
  ZKMI_OK = 0,
  ZKMI_ERR_ARG = -1,
  ZKMI_ERR_HIP = -2,
  ZKMI_ERR_NO_DEVICE = -3,
  ZKMI_ERR_OOM = -4,
  ZKMI_ERR_UNSATISFIED = -5 /* at least one proof's witness does not satisfy the system */
};

/* -- context ----------------------------------------------------------------------------- */
/* Selects HIP device `device`, creates the streams and the NTT twiddle cache.  Fails with
 * ZKMI_ERR_NO_DEVICE when no gfx950 device is visible: there is no CPU fallback. */
int zkmi_init(int device, zkmi_ctx** out);
void zkmi_destroy(zkmi_ctx* ctx);
const char* zkmi_last_error(zkmi_ctx* ctx);
/* Blocks until all work queued on the context's stream has finished. */
int zkmi_sync(zkmi_ctx* ctx);
/* HIP stream (hipStream_t) the context launches on -- for callers that time with HIP events. */
void* zkmi_stream(zkmi_ctx* ctx);

/* -- field / group primitives (parity tests; a Go shim that keeps gnark's own solver) ------- */
/* r[i] = a[i] * b[i] in fr (which = 0) or fp (which = 1).   stands in for fr.Element.Mul / fp.Element.Mul */
int zkmi_field_mul(zkmi_ctx* ctx, int which, const void* a, const void* b, void* r, size_t n);
/* Throughput microbenchmark: every lane runs `iters` dependent Montgomery products; returns
 * products per second in *rate (integer-ALU roofline measurement).  which: 0 = fr,
 * 1 = fp (8 x 32-bit limbs, ff.h), 2 = fp in the 9 x 29-bit form the G1 MSM uses (ff29.h). */
int zkmi_field_mul_bench(zkmi_ctx* ctx, int which, size_t n_threads, int iters, double* rate);

/* Batched NTT over fr, natural order in and out, in place.  stands in for
 * fft.Domain.FFT / FFTInverse (gnark-crypto ecc/bn254/fr/fft) with fft.OnCoset() = `coset`.
 *   data: batch x 2^log_n fr elements, proof-major. */
int zkmi_ntt_batch(zkmi_ctx* ctx, void* data, int log_n, size_t batch, int inverse, int coset);

/* h = coefficients of (A.B - C)/Z_H.   stands in for computeH in gnark
 * backend/groth16/bn254/prove.go.  a, b, c: batch x 2^log_n evaluations (zero padded);
 * h_out: batch x 2^log_n coefficients, natural order. */
int zkmi_h_batch(zkmi_ctx* ctx, const void* a, const void* b, const void* c, void* h_out,
                 int log_n, size_t batch);

/* Fixed-base MSM handle: bases are uploaded once and expanded into HBM-resident tables of signed
 * window multiples.  group: 1 = G1, 2 = G2.
 *   window_bits = 0: widest windows the free HBM allows, ONE table d*P (d = 1..2^(c-1)) per base
 *                    shared by all windows, per-window accumulators combined by Horner's rule;
 *   100 + c (c in 4..16): that layout with an explicit width;
 *   c in 2..16: the per-window layout (a table d*2^(c*j)*P for every window j, one accumulator);
 *   200 + k (k in 2..20): comb tables -- one joint table of all subset sums per group of k bases,
 *                    254 one-bit windows (254 / k additions per base and proof);
 *   300 + k (k in 3..21): sign-pattern comb tables -- every scalar rewritten as a sum of 254 signed
 *                    powers of two, one entry P_(k-1) + sum_{i<k-1} +-P_i per sign pattern
 *                    (2^(k-1) entries serve k bases: one more base per group than 200 + k in the
 *                    same HBM, but no digit is ever "nothing to add"), 255 windows (254 + a parity
 *                    correction).  The auto plan's choice for dense witnesses (the Arbo-160 key:
 *                    319 for G1).
 *   Auto picks the layout with the fewest additions that fits. */
typedef struct zkmi_msm_bases zkmi_msm_bases;
int zkmi_msm_bases_load(zkmi_ctx* ctx, int group, const void* bases, size_t n, int window_bits,
                        zkmi_msm_bases** out);
void zkmi_msm_bases_free(zkmi_ctx* ctx, zkmi_msm_bases* b);
/* out[p] = sum_i scalars[p][i] * bases[i].   stands in for G1Jac/G2Jac.MultiExp (gnark-crypto
 * ecc/bn254/multiexp.go) called once per proof with the same bases.
 *   scalars: batch x n fr elements (Montgomery), proof-major; out: batch affine points. */
int zkmi_msm_batch(zkmi_ctx* ctx, const zkmi_msm_bases* bases, const void* scalars, size_t batch,
                   void* out);
/* out[i] = scalars[i] * base.   stands in for curve.BatchScalarMultiplicationG1/G2 used by
 * groth16.Setup (gnark backend/groth16/bn254/setup.go). */
int zkmi_fixed_base_mul(zkmi_ctx* ctx, int group, const void* base, const void* scalars, size_t n,
                        void* out);

/* -- proving key --------------------------------------------------------------------------- */
/* Mirrors gnark's groth16 bn254 ProvingKey: Domain, G1{Alpha,Beta,Delta,A,B,Z,K}, G2{Beta,Delta,B},
 * InfinityA / InfinityB [UPSTREAM-RECALL, SURVEY.md §3.2].  Which wire each retained base belongs
 * to can be given either way:
 *   - gnark's own fields: infinity_a / infinity_b (one byte per wire, Go []bool: 1 = the point is
 *     at infinity and absent from g1_a / g1_b / g2_b) with a_wire = b_wire = NULL, and n_public
 *     (pk.G1.K holds wires n_public .. n_wires-1 in order) with k_wire = NULL;
 *   - or explicit index arrays a_wire / b_wire / k_wire (what this repo's own setup emits).
 * n_a / n_b / n_k are always the lengths of g1_a / g1_b (= g2_b) / g1_k. */
/* One commitment of gnark's Groth16 commitment extension (api.Commit; std/rangecheck and the
 * lookup arguments build on it): constraint.Groth16Commitment {PrivateCommitted,
 * PublicAndCommitmentCommitted, CommitmentIndex} and pedersen.ProvingKey {Basis, BasisExpSigma} of
 * pk.CommitmentKeys[i] [UPSTREAM-RECALL, SURVEY.md §3.2 step 6].  The prover commits to the private
 * wires (MSM over basis), derives the commitment wire's value by hash_to_field over the commitment
 * and the hashed wires' values (RFC 9380 expand_message_xmd, SHA-256, DST "bsb22-commitment"), and
 * proves knowledge with the same scalars over basis_exp_sigma (folded over all commitments with
 * powers of Hash(commitment wire values, DST "G16-BSB22")). */
typedef struct {
  uint32_t n_private;            /* basis points = committed private wires */
  uint32_t n_hashed;             /* public wires / earlier commitment wires hashed with it */
  uint32_t commitment_wire;      /* wire that receives the challenge */
  uint32_t reserved;
  const uint32_t* private_wires; /* n_private wire indices, basis order */
  const uint32_t* hashed_wires;  /* n_hashed wire indices */
  const void* basis;             /* n_private G1 affine */
  const void* basis_exp_sigma;   /* n_private G1 affine */
} zkmi_commitment_desc;

typedef struct {
  uint32_t log_n;       /* domain size 2^log_n */
  uint32_t n_wires;     /* columns of the constraint system, including ONE */
  uint32_t n_a, n_b, n_k, n_z;
  const uint32_t* a_wire; /* n_a: wire index of g1_a[i] (InfinityA filtered out), or NULL */
  const uint32_t* b_wire; /* n_b: wire index of g1_b[i] and g2_b[i], or NULL */
  const uint32_t* k_wire; /* n_k: wire index of g1_k[i] (private wires), or NULL */
  const void* g1_a;
  const void* g1_b;
  const void* g1_k;
  const void* g1_z; /* n_z >= 2^log_n - 1 points; the first 2^log_n - 1 are used */
  const void* g2_b;
  const void* g1_alpha;
  const void* g1_beta;
  const void* g1_delta;
  const void* g2_beta;
  const void* g2_delta;
  uint32_t window_bits_g1; /* as zkmi_msm_bases_load: 0 = default */
  uint32_t window_bits_g2;
  /* gnark-shaped alternative to the index arrays (used when the matching *_wire is NULL) */
  const uint8_t* infinity_a; /* n_wires bytes */
  const uint8_t* infinity_b; /* n_wires bytes */
  uint32_t n_public;         /* public wires including ONE */
  /* -- HBM plan (all optional: 0 = default) -- */
  /* Largest batch that will be proved with this key.  The auto window plan (window_bits = 0) sizes
   * the MSM tables so that the prover's working set for max_batch -- two pipeline sets of value
   * file + a, b, c, the NTT scratch, MSM digits and partial sums -- still fits beside them
   * (wider batches get narrower tables instead of ZKMI_ERR_OOM at prove time).  0 = 1024. */
  uint32_t max_batch;
  /* Upper bound for the key's MSM tables in bytes (G1 + G2), e.g. to leave room for a second key
   * on the same GPU.  0 = whatever the free HBM minus the working set allows. */
  uint64_t table_budget_bytes;
  /* Value slots of the constraint system that will be solved with this key (zkmi_cs_desc.n_slots)
   * for the working-set estimate; 0 = n_wires + 5 %. */
  uint32_t n_slots_hint;
  /* Comb / shared-table MSMs: (window, chunk) blocks in flight as a multiple of the chip's wave
   * slots (measured best: 16 for comb, 8 for shared tables).  0 = default. */
  uint32_t msm_chunk_factor;
  /* 1 = most wire values of this circuit are bits or small integers (e.g. Keccak, bit
   * decompositions): the auto plan then keeps subset-sum comb tables for every MSM, whose zero
   * digits are skipped, instead of sign-pattern tables (one more base per group for the same HBM,
   * but no digit of a sign pattern is ever "nothing to add").  2 = (almost) all wires are bits: the
   * wire MSMs additionally get small tables (one addition per group is all they execute) and the
   * quotient MSM, whose scalars are dense, takes the freed HBM with its own, wider plan.
   * 0 = dense field elements (Poseidon). */
  uint32_t sparse_witness;
  /* Commitment extension: 0 / NULL for a plain Groth16 key.  With k_wire = NULL the library leaves
   * the private committed wires and the commitment wires out of pk.G1.K, as gnark's setup does. */
  uint32_t n_commitments;
  const zkmi_commitment_desc* commitments;
} zkmi_pk_desc;
/* Copies the key to the device and builds the MSM window tables; host buffers may be freed
 * afterwards.  One-off per circuit (gnark's icicle backend does the same lazily). */
int zkmi_pk_load(zkmi_ctx* ctx, const zkmi_pk_desc* desc, zkmi_pk** out);
void zkmi_pk_free(zkmi_ctx* ctx, zkmi_pk* pk);
/* Window plan chosen for the key: info[0] = windows per G1 scalar, [1] = table entries per G1
 * base, [2] = windows per G2 scalar, [3] = table entries per G2 base, [4] = G1 table bytes (all
 * four MSMs), [5] = G2 table bytes, [6] / [7] = 1 when the G1 / G2 tables are shared-table plans,
 * [8] / [9] = group size k of the G1 / G2 comb tables (0 otherwise; then [0] / [2] = 254). */
int zkmi_pk_info(const zkmi_pk* pk, uint64_t* info /* 10 */);

/* -- constraint system (witness program) ----------------------------------------------------- */
/* What cs.R1CS.Solve needs, in the scheduled form produced by
 * gnark_crypto_primitives_amd.frontend.compile_circuit: the circuit's field
 * operations packed into steps of up to `lanes_per_proof` independent operations of one class;
 * the GPU runs a step with that many lanes of a wavefront per proof.
 *   program: n_rows x (1 + lanes_per_proof) x 4 words.  Row = header (class, active, aux, 0) +
 *   one operand quad per sub-lane (op | check << 5 | class << 6 | constraint_row << 9, dst slot,
 *   a, b).  Systems with commitments hold one COMMIT row per commitment (header class 9, aux =
 *   commitment index): the solver stops in front of it, the prover commits, hashes, writes the
 *   challenge into the commitment wire and resumes (zkmi_prove_submit needs a key whose
 *   n_commitments matches).  Unit rows with operand rows behind them: multiplicities of a lookup
 *   table (class 8: gnark std/lookup/logderivarg's count hint) and the quotient / remainder of a
 *   multi-limb product by a 4 x 64-bit modulus (class 11: the mulHint of gnark's std/math/emulated;
 *   header (11, na + nb, rows, nout | na << 8 | first modulus constant << 12), the limb slots of the
 *   two operands in the rows that follow, nout consecutive wires written). */
typedef struct {
  uint32_t n_wires, n_public, n_secret, n_constraints;
  uint32_t n_slots, n_rows, n_consts;
  uint32_t lanes_per_proof; /* a power of two, 1 .. 64 */
  const uint32_t* program;
  const void* consts;      /* n_consts fr elements, Montgomery */
} zkmi_cs_desc;
int zkmi_cs_load(zkmi_ctx* ctx, const zkmi_cs_desc* desc, zkmi_cs** out);
void zkmi_cs_free(zkmi_ctx* ctx, zkmi_cs* cs);

/* Witness solve only.  stands in for cs.R1CS.Solve (gnark constraint/bn254).
 *   inputs: batch x (n_public - 1 + n_secret) fr elements, Montgomery, proof-major
 *   wires_out (optional): batch x n_wires ; abc_out (optional): 3 x batch x n_constraints
 *   status_out: per proof, 0 or ZKMI_ERR_UNSATISFIED */
int zkmi_solve_batch(zkmi_ctx* ctx, const zkmi_cs* cs, const void* inputs, size_t batch,
                     void* wires_out, void* abc_out, int32_t* status_out);

/* Full Groth16 prove for a batch of independent witnesses.  stands in for groth16.Prove(ccs, pk,
 * fullWitness) called `batch` times (gnark backend/groth16/bn254/prove.go).
 *   inputs: as zkmi_solve_batch
 *   rs: batch x 2 fr elements (Montgomery): the prover's blinding scalars r, s.  gnark samples
 *       them inside Prove; they are an input here so results are reproducible.
 *   proofs_out: batch x 256 B: Ar (G1) | Krs (G1) | Bs (G2), the field order of gnark's Proof
 *   status_out: per proof */
int zkmi_prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_cs* cs, const void* inputs,
                     size_t batch, const void* rs, void* proofs_out, int32_t* status_out);

/* The same prove split in two so that consecutive batches overlap: `submit` stages the inputs and
 * runs the witness solve on a second HIP stream, `collect` runs quotient + MSMs + assembly of the
 * OLDEST submitted batch and blocks until its proofs are written.  At most two batches may be in
 * flight; submit(k+1) before collect(k) hides the latency-bound solve of batch k+1 under batch
 * k's MSMs, and collect(k) queues batch k+1's quotient and MSM kernels before it waits, so batch
 * k's (equally latency-bound) assembly runs underneath them on a third stream.
 * Device-pointer inputs / rs must stay valid until the matching collect.  While a batch is in
 * flight the other entry points of the same context return ZKMI_ERR_ARG. */
int zkmi_prove_submit(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_cs* cs, const void* inputs,
                      size_t batch, const void* rs);
int zkmi_prove_collect(zkmi_ctx* ctx, void* proofs_out, int32_t* status_out);
/* The same for keys with the commitment extension: commitments_out receives, per proof,
 * (n_commitments + 1) G1 affine points: proof.Commitments[0..n-1] then proof.CommitmentPok.
 * zkmi_prove_collect on such a key returns ZKMI_ERR_ARG (a proof without them cannot verify). */
int zkmi_prove_collect_ex(zkmi_ctx* ctx, void* proofs_out, int32_t* status_out,
                          void* commitments_out);

/* -- the gnark drop-in entry: prove from SOLVED witnesses --------------------------------------- */
/* For a caller that keeps gnark's own solver (cs.Solve -> solution.W, and optionally solution.A,
 * .B, .C) and hands quotient, MSMs and assembly to the GPU -- the place of groth16.Prove(ccs, pk,
 * fullWitness) (reference call sites: tree/test/verifier_bn254_test.go:41,67; what gnark's icicle
 * backend accelerates [UPSTREAM-RECALL, SURVEY.md §3.2, §8b]).  No zkmi_cs is needed.
 *
 * Page-locked host memory for the batch arrays.  A shim assembles a batch from per-proof witness
 * vectors anyway; assembling it in memory from zkmi_host_alloc lets the DMA engine read it in place
 * (no staging copy, truly asynchronous submit).  Ordinary (pageable) memory is accepted everywhere
 * and staged through a ring of pinned chunks owned by the context. */
void* zkmi_host_alloc(zkmi_ctx* ctx, size_t bytes);
void zkmi_host_free(zkmi_ctx* ctx, void* p);
/* Host threads that copy pageable memory into the pinned ring (0 = default 4). */
int zkmi_set_copy_threads(zkmi_ctx* ctx, int threads);

/* The R1CS matrices, so that a caller ships the wire vector only and a = L.w, b = R.w, c = O.w are
 * formed on the GPU ("zkmi_cs_load ... CSR A,B,C").  Mirrors gnark's
 * constraint.R1CS [UPSTREAM-RECALL]: a coefficient table (cs.Coefficients, fr Montgomery) and, per
 * constraint, three linear expressions of terms {coefficient index, wire index} (constraint.Term
 * {CID, VID}); *_ptr are n_constraints + 1 offsets into the term arrays, starting at 0. */
typedef struct {
  uint32_t coeff; /* index into coeffs */
  uint32_t wire;  /* column: 0 = ONE, then public, secret, internal wires */
} zkmi_term;
typedef struct {
  uint32_t n_wires, n_constraints, n_coeffs;
  const void* coeffs; /* n_coeffs fr elements, Montgomery, reduced */
  const uint32_t* l_ptr;
  const zkmi_term* l_terms;
  const uint32_t* r_ptr;
  const zkmi_term* r_terms;
  const uint32_t* o_ptr;
  const zkmi_term* o_terms;
} zkmi_r1cs_desc;
typedef struct zkmi_r1cs zkmi_r1cs;
/* Validates every index on the host, copies the matrices to the device; host buffers may be freed
 * afterwards. */
int zkmi_r1cs_load(zkmi_ctx* ctx, const zkmi_r1cs_desc* desc, zkmi_r1cs** out);
void zkmi_r1cs_free(zkmi_ctx* ctx, zkmi_r1cs* r1cs);

/* Stage 1 of a prove from solved witnesses; the matching zkmi_prove_collect returns the proofs.
 * Pipelines exactly like zkmi_prove_submit (two batches in flight, same streams, same HBM plan):
 * batch k+1 streams in over PCIe underneath batch k's MSM kernels.
 *   wires: batch x pk.n_wires fr elements (Montgomery, gnark's image), proof-major: the full wire
 *          vector including the ONE wire at index 0; 16-byte aligned
 *   r1cs != NULL: a = b = c = NULL, n_constraints = 0 or the loaded system's; the device forms
 *          a, b, c and checks a.b = c (per-proof ZKMI_ERR_UNSATISFIED in collect's status_out)
 *   r1cs == NULL: a, b, c: batch x n_constraints fr elements each (<L_k,w>, <R_k,w>, <O_k,w>,
 *          solution.A/B/C); taken as they are (status_out = 0); the library pads to the domain
 *   rs: batch x 2 fr, as zkmi_prove_batch
 * Pageable host arrays have been consumed when the call returns; page-locked and device arrays
 * must stay valid until the matching collect. */
int zkmi_prove_witness_submit(zkmi_ctx* ctx, const zkmi_pk* pk, const zkmi_r1cs* r1cs,
                              const void* wires, const void* a, const void* b, const void* c,
                              size_t n_constraints, size_t batch, const void* rs);
/* Blocking form: zkmi_prove_witness_submit(r1cs = NULL) + zkmi_prove_collect (returns
 * ZKMI_ERR_ARG while submitted batches are in flight). */
int zkmi_prove_witness_batch(zkmi_ctx* ctx, const zkmi_pk* pk, const void* wires, const void* a,
                             const void* b, const void* c, size_t n_constraints, size_t batch,
                             const void* rs, void* proofs_out);

/* -- PLONK (BASELINE config 5 names this backend) ----------------------------------------------- */
/* Stands in for plonk.Prove of gnark backend/plonk/bn254 [UPSTREAM-RECALL]: KZG commitments over
 * the SRS, blinded wire polynomials, permutation grand product, quotient on the coset 5<w_4n>
 * split in three, linearisation and two openings.  The Fiat-Shamir hashing stays on the host, as
 * in gnark, between five round calls; protocol and transcript: DESIGN.md (parity unpinned: the
 * reference holds no PLONK vector).  All arrays: fr in gnark's Montgomery image; batch arrays are
 * proof-major; host or device pointers. */
typedef struct zkmi_plonk_pk zkmi_plonk_pk;
typedef struct {
  uint32_t log_n;          /* domain 2^log_n >= number of gates, 4 .. 24 */
  uint32_t n_public;       /* public inputs (without ONE): the first n_public gates */
  const void* coef;        /* 8 x n: qL, qR, qO, qM, qC, S1, S2, S3 in coefficient form */
  const void* coset;       /* 8 x 4n: the same polynomials on the coset 5 <w_4n>, natural order */
  const void* sigma;       /* 3 x n: S1, S2, S3 values on the domain (k_c w^r of the image position) */
  const void* omega;       /* n: w^i */
  const void* coset_x;     /* 4n: 5 w_4n^j */
  const void* l1_coset;    /* 4n: L_1 on the coset */
  const void* zh_inv;      /* 4: 1 / Z_H on the coset (period 4) */
  const void* srs_g1;      /* n + 6 G1 affine points: [tau^i]_1 */
  uint32_t window_bits;    /* MSM table plan of the SRS, as zkmi_msm_bases_load */
  uint32_t max_batch;      /* 0 = 64 */
  /* Optional (lag_k = 0: unused): commit the wire columns a, b, c in the Lagrange basis, as gnark
   * does with its Lagrange-form SRS [UPSTREAM-RECALL], instead of their coefficient forms.  The
   * scalars are then the witness values themselves: for circuits whose wires are mostly bits or
   * small integers (Keccak, bit decompositions) almost every digit of the subset-sum tables is
   * zero and skipped.  lag_g1[c] (c = 0, 1, 2 for a, b, c): n + 2 G1 points in the order the MSM
   * groups them (lag_k per group); lag_rows[c][j] = row of column c whose value multiplies point j:
   * a gate row r < n for [L_r(tau)]_1, n / n + 1 for the blinding points [tau^(n+1) - tau]_1 /
   * [tau^n - 1]_1 (scalars b1 / b2 of the column).  Order rows of large values first; only speed
   * depends on the order.  Same commitments as in coefficient form. */
  const void* lag_g1[3];
  const uint32_t* lag_rows[3];
  uint32_t lag_k;          /* bases per subset-sum table of the Lagrange points, 4 .. 16; 0 = off */
} zkmi_plonk_pk_desc;
int zkmi_plonk_pk_load(zkmi_ctx* ctx, const zkmi_plonk_pk_desc* desc, zkmi_plonk_pk** out);
void zkmi_plonk_pk_free(zkmi_ctx* ctx, zkmi_plonk_pk* pk);
/* Round 1: witness solve (cs: the circuit's gate rows, frontend/scs.py), blinding (blind: batch x 9
 * fr: b1 X + b2 for a, b, c and b7 X^2 + b8 X + b9 for z), commits_out: batch x 3 G1 ([a] [b] [c]). */
int zkmi_plonk_round1(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const zkmi_cs* cs, const void* inputs,
                      size_t batch, const void* blind, void* commits_out, int32_t* status_out);
/* Round 2: beta_gamma: batch x 2 fr; commit_z_out: batch G1. */
int zkmi_plonk_round2(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* beta_gamma, void* commit_z_out);
/* Round 3: alpha: batch fr; commits_t_out: batch x 3 G1 ([t_lo] [t_mid] [t_hi], n + 2 coefficients each). */
int zkmi_plonk_round3(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* alpha, void* commits_t_out);
/* Round 4: zeta_zetaw: batch x 2 fr (zeta, zeta w); evals_out: batch x 6 fr: a, b, c, S1, S2 at zeta,
 * z at zeta w. */
int zkmi_plonk_round4(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* zeta_zetaw, void* evals_out);
/* Round 5: scalars: batch x 14 fr (coefficients of the linearisation polynomial: qM, qL, qR, qO, S3,
 * z, t_lo, t_mid, t_hi; its constant term minus sum v^i e_i; v; zeta; zeta w; z(zeta w));
 * commits_w_out: batch x 2 G1 ([W_zeta] [W_zeta_w]). */
int zkmi_plonk_round5(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const void* scalars, void* commits_w_out);

/* plonk.Prove for a batch in one call: the five rounds above with the Fiat-Shamir transcript (SHA-256,
 * DESIGN.md §3.6) computed on the host inside the library between them.  vk_digest: the 32-byte
 * digest of the verifying key that opens the transcript (plonk.py: ProvingKey.vk_digest).
 *   proofs_out: batch x 768 bytes: 9 G1 affine points ([a] [b] [c] [z] [t_lo] [t_mid] [t_hi]
 *   [W_zeta] [W_zeta_w]) then 6 fr (a, b, c, S1, S2 at zeta, z at zeta w), gnark's Montgomery image. */
int zkmi_plonk_prove(zkmi_ctx* ctx, zkmi_plonk_pk* pk, const zkmi_cs* cs, const void* inputs,
                     size_t batch, const void* blind, const uint8_t* vk_digest, void* proofs_out,
                     int32_t* status_out);

/* Per-stage device time of the last zkmi_prove_collect / zkmi_prove_batch in milliseconds, from
 * HIP events on the library's streams: [0] solve (stage 1, second stream), [1] quotient (NTTs +
 * pointwise), [2] G1 MSMs, [3] G2 MSM, [4] assembly (third stream; overlaps the next batch's
 * quotient when one is submitted), [5] main-stream span = [1] + [2] + [3] + delta multiples,
 * [6] sum over the four G1 msm_accumulate launches alone (one event pair around each launch),
 * [7] the G2 msm_accumulate launch alone. */
int zkmi_last_timings(zkmi_ctx* ctx, double* ms_out /* 8 doubles */);

#ifdef __cplusplus
}
#endif
#endif
