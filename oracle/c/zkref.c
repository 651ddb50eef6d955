/* zkref.c -- CPU ORACLE (TEST INFRASTRUCTURE, not product code).
 *
 * Plain-C restatement of the Groth16/BN254 prover arithmetic that the reference reaches through
 * gnark / gnark-crypto (third-party modules pinned in /root/reference/go.mod:8-9:
 *   github.com/consensys/gnark        v0.14.1-0.20251203003358-cce547909fed
 *   github.com/consensys/gnark-crypto v0.19.3-0.20251115174214-022ec58e8c19
 * neither is present in /root/reference, and no Go toolchain exists here).  The algorithms are
 * restated from their published descriptions:
 *   - fr/fp Element.Mul: Montgomery CIOS on 4x64-bit limbs, R = 2^256
 *   - fft.Domain.FFT/FFTInverse: radix-2 DIF / DIT with implicit bit reversal, coset by g = 5
 *   - computeH (backend/groth16/bn254/prove.go): 3 iFFT, 3 coset FFT, (a*b-c)/(g^n-1), coset iFFT
 *   - G1/G2 MultiExp: signed-digit Pippenger with extended-Jacobian buckets
 *   - r1cs Solve: walk constraints, one unknown wire each, hints as callbacks
 *   - groth16.Prove assembly: Ar, Bs, Krs
 * PARITY STATUS: prover-level results are pinned by no vector in the reference
 * ("parity unpinned").  What pins this file: the field constants K6, the
 * gadget-level KATs K1-K4 (through the Python oracle that cross-checks this one), NTT vs naive
 * DFT, MSM vs naive double-and-add, and pairing verification of complete proofs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle/c   (gcc -O3 -fopenmp -shared)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t fe[4];

typedef struct {
  fe p, one, r2;
  uint64_t inv;
} field_t;

static const field_t FR = {
    {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full},
    {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull},
    0xc2e1f593efffffffull};
static const field_t FQ = {
    {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full},
    {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full},
    0x87d20782e4866389ull};

/* ---------------------------------------------------------------- field ------------------- */
static inline int fe_is_zero(const fe a) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static inline int fe_eq(const fe a, const fe b) {
  return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) == 0;
}
static inline void fe_set(fe r, const fe a) { memcpy(r, a, 32); }
static inline void fe_zero(fe r) { memset(r, 0, 32); }

static inline int fe_geq(const fe a, const fe b) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static inline void fe_sub_raw(fe r, const fe a, const fe b) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static inline void fe_add(fe r, const fe a, const fe b, const field_t* F) {
  u128 c = 0;
  fe t;
  for (int i = 0; i < 4; i++) {
    c += (u128)a[i] + b[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (fe_geq(t, F->p))
    fe_sub_raw(r, t, F->p);
  else
    fe_set(r, t);
}
static inline void fe_sub(fe r, const fe a, const fe b, const field_t* F) {
  fe t;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t[i] + F->p[i];
      r[i] = (uint64_t)c;
      c >>= 64;
    }
  } else
    fe_set(r, t);
}
static inline void fe_neg(fe r, const fe a, const field_t* F) {
  if (fe_is_zero(a))
    fe_zero(r);
  else
    fe_sub_raw(r, F->p, a);
}
static inline void fe_dbl(fe r, const fe a, const field_t* F) { fe_add(r, a, a, F); }

/* Montgomery CIOS, 4 limbs */
static inline void fe_mul(fe r, const fe a, const fe b, const field_t* F) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->inv;
    c = (u128)m * F->p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * F->p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe res = {t[0], t[1], t[2], t[3]};
  if (t[4] || fe_geq(res, F->p)) fe_sub_raw(res, res, F->p);
  fe_set(r, res);
}
static inline void fe_sqr(fe r, const fe a, const field_t* F) { fe_mul(r, a, a, F); }
static inline void fe_from_mont(fe r, const fe a, const field_t* F) {
  fe o = {1, 0, 0, 0};
  fe_mul(r, a, o, F);
}
static inline void fe_to_mont(fe r, const fe a, const field_t* F) { fe_mul(r, a, F->r2, F); }

static void fe_pow(fe r, const fe a, const fe e, const field_t* F) {
  fe acc;
  fe_set(acc, F->one);
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      fe_sqr(acc, acc, F);
      if ((e[i] >> b) & 1) fe_mul(acc, acc, a, F);
    }
  fe_set(r, acc);
}
static void fe_inv(fe r, const fe a, const field_t* F) {
  fe e;
  fe two = {2, 0, 0, 0};
  fe_sub_raw(e, F->p, two);
  fe_pow(r, a, e, F);
}

/* ---------------------------------------------------------------- Fq2 --------------------- */
typedef struct {
  fe c0, c1;
} fe2;
static inline int fe2_is_zero(const fe2* a) { return fe_is_zero(a->c0) && fe_is_zero(a->c1); }
static inline void fe2_add(fe2* r, const fe2* a, const fe2* b) {
  fe_add(r->c0, a->c0, b->c0, &FQ);
  fe_add(r->c1, a->c1, b->c1, &FQ);
}
static inline void fe2_sub(fe2* r, const fe2* a, const fe2* b) {
  fe_sub(r->c0, a->c0, b->c0, &FQ);
  fe_sub(r->c1, a->c1, b->c1, &FQ);
}
static inline void fe2_neg(fe2* r, const fe2* a) {
  fe_neg(r->c0, a->c0, &FQ);
  fe_neg(r->c1, a->c1, &FQ);
}
static inline void fe2_mul(fe2* r, const fe2* a, const fe2* b) {
  /* schoolbook, u^2 = -1 */
  fe t0, t1, t2, t3;
  fe_mul(t0, a->c0, b->c0, &FQ);
  fe_mul(t1, a->c1, b->c1, &FQ);
  fe_mul(t2, a->c0, b->c1, &FQ);
  fe_mul(t3, a->c1, b->c0, &FQ);
  fe_sub(r->c0, t0, t1, &FQ);
  fe_add(r->c1, t2, t3, &FQ);
}
static inline void fe2_sqr(fe2* r, const fe2* a) { fe2_mul(r, a, a); }
static void fe2_inv(fe2* r, const fe2* a) {
  fe n, t, ni;
  fe_sqr(n, a->c0, &FQ);
  fe_sqr(t, a->c1, &FQ);
  fe_add(n, n, t, &FQ);
  fe_inv(ni, n, &FQ);
  fe_mul(r->c0, a->c0, ni, &FQ);
  fe_mul(t, a->c1, ni, &FQ);
  fe_neg(r->c1, t, &FQ);
}

/* ---------------------------------------------------------------- curves ------------------ */
/* generic-over-field point code, instantiated for G1 (fe over FQ) and G2 (fe2) */
#define G1F_T fe
#define DEFINE_GROUP(PFX, T, SET, ZERO, ISZ, EQ, ADD, SUB, NEG, MUL, SQR, INV, ONE)              \
  typedef struct {                                                                               \
    T x, y;                                                                                      \
  } PFX##_aff;                                                                                   \
  typedef struct {                                                                               \
    T x, y, zz, zzz;                                                                             \
  } PFX##_xyzz;                                                                                  \
  static inline int PFX##_aff_is_inf(const PFX##_aff* a) { return ISZ(a->x) && ISZ(a->y); }      \
  static inline void PFX##_set_inf(PFX##_xyzz* r) {                                              \
    ONE(r->x);                                                                                   \
    ONE(r->y);                                                                                   \
    ZERO(r->zz);                                                                                 \
    ZERO(r->zzz);                                                                                \
  }                                                                                              \
  static void PFX##_dbl_aff(PFX##_xyzz* r, const PFX##_aff* a) {                                 \
    if (PFX##_aff_is_inf(a)) {                                                                   \
      PFX##_set_inf(r);                                                                          \
      return;                                                                                    \
    }                                                                                            \
    T u, v, w, s, x2, m, t;                                                                      \
    ADD(u, a->y, a->y);                                                                          \
    SQR(v, u);                                                                                   \
    MUL(w, u, v);                                                                                \
    MUL(s, a->x, v);                                                                             \
    SQR(x2, a->x);                                                                               \
    ADD(m, x2, x2);                                                                              \
    ADD(m, m, x2);                                                                               \
    SQR(t, m);                                                                                   \
    SUB(t, t, s);                                                                                \
    SUB(t, t, s);                                                                                \
    T y3, wy;                                                                                    \
    SUB(y3, s, t);                                                                               \
    MUL(y3, m, y3);                                                                              \
    MUL(wy, w, a->y);                                                                            \
    SUB(y3, y3, wy);                                                                             \
    SET(r->x, t);                                                                                \
    SET(r->y, y3);                                                                               \
    SET(r->zz, v);                                                                               \
    SET(r->zzz, w);                                                                              \
  }                                                                                              \
  static void PFX##_dbl(PFX##_xyzz* r, const PFX##_xyzz* p) {                                    \
    if (ISZ(p->zz)) {                                                                            \
      *r = *p;                                                                                   \
      return;                                                                                    \
    }                                                                                            \
    T u, v, w, s, x2, m, t, y3, wy, zz, zzz;                                                     \
    ADD(u, p->y, p->y);                                                                          \
    SQR(v, u);                                                                                   \
    MUL(w, u, v);                                                                                \
    MUL(s, p->x, v);                                                                             \
    SQR(x2, p->x);                                                                               \
    ADD(m, x2, x2);                                                                              \
    ADD(m, m, x2);                                                                               \
    SQR(t, m);                                                                                   \
    SUB(t, t, s);                                                                                \
    SUB(t, t, s);                                                                                \
    SUB(y3, s, t);                                                                               \
    MUL(y3, m, y3);                                                                              \
    MUL(wy, w, p->y);                                                                            \
    SUB(y3, y3, wy);                                                                             \
    MUL(zz, v, p->zz);                                                                           \
    MUL(zzz, w, p->zzz);                                                                         \
    SET(r->x, t);                                                                                \
    SET(r->y, y3);                                                                               \
    SET(r->zz, zz);                                                                              \
    SET(r->zzz, zzz);                                                                            \
  }                                                                                              \
  static void PFX##_madd(PFX##_xyzz* acc, const PFX##_aff* q) {                                  \
    if (PFX##_aff_is_inf(q)) return;                                                             \
    if (ISZ(acc->zz)) {                                                                          \
      SET(acc->x, q->x);                                                                         \
      SET(acc->y, q->y);                                                                         \
      ONE(acc->zz);                                                                              \
      ONE(acc->zzz);                                                                             \
      return;                                                                                    \
    }                                                                                            \
    T u2, s2, p, r, pp, ppp, qq, x3, y3, t;                                                      \
    MUL(u2, q->x, acc->zz);                                                                      \
    MUL(s2, q->y, acc->zzz);                                                                     \
    SUB(p, u2, acc->x);                                                                          \
    SUB(r, s2, acc->y);                                                                          \
    if (ISZ(p)) {                                                                                \
      if (ISZ(r))                                                                                \
        PFX##_dbl_aff(acc, q);                                                                   \
      else                                                                                       \
        PFX##_set_inf(acc);                                                                      \
      return;                                                                                    \
    }                                                                                            \
    SQR(pp, p);                                                                                  \
    MUL(ppp, p, pp);                                                                             \
    MUL(qq, acc->x, pp);                                                                         \
    SQR(x3, r);                                                                                  \
    SUB(x3, x3, ppp);                                                                            \
    SUB(x3, x3, qq);                                                                             \
    SUB(x3, x3, qq);                                                                             \
    SUB(y3, qq, x3);                                                                             \
    MUL(y3, r, y3);                                                                              \
    MUL(t, acc->y, ppp);                                                                         \
    SUB(y3, y3, t);                                                                              \
    SET(acc->x, x3);                                                                             \
    SET(acc->y, y3);                                                                             \
    MUL(acc->zz, acc->zz, pp);                                                                   \
    MUL(acc->zzz, acc->zzz, ppp);                                                                \
  }                                                                                              \
  static void PFX##_padd(PFX##_xyzz* acc, const PFX##_xyzz* b) {                                 \
    if (ISZ(b->zz)) return;                                                                      \
    if (ISZ(acc->zz)) {                                                                          \
      *acc = *b;                                                                                 \
      return;                                                                                    \
    }                                                                                            \
    T u1, u2, s1, s2, p, r, pp, ppp, qq, x3, y3, t;                                              \
    MUL(u1, acc->x, b->zz);                                                                      \
    MUL(u2, b->x, acc->zz);                                                                      \
    MUL(s1, acc->y, b->zzz);                                                                     \
    MUL(s2, b->y, acc->zzz);                                                                     \
    SUB(p, u2, u1);                                                                              \
    SUB(r, s2, s1);                                                                              \
    if (ISZ(p)) {                                                                                \
      if (ISZ(r)) {                                                                              \
        PFX##_xyzz d;                                                                            \
        PFX##_dbl(&d, acc);                                                                      \
        *acc = d;                                                                                \
      } else                                                                                     \
        PFX##_set_inf(acc);                                                                      \
      return;                                                                                    \
    }                                                                                            \
    SQR(pp, p);                                                                                  \
    MUL(ppp, p, pp);                                                                             \
    MUL(qq, u1, pp);                                                                             \
    SQR(x3, r);                                                                                  \
    SUB(x3, x3, ppp);                                                                            \
    SUB(x3, x3, qq);                                                                             \
    SUB(x3, x3, qq);                                                                             \
    SUB(y3, qq, x3);                                                                             \
    MUL(y3, r, y3);                                                                              \
    MUL(t, s1, ppp);                                                                             \
    SUB(y3, y3, t);                                                                              \
    SET(acc->x, x3);                                                                             \
    SET(acc->y, y3);                                                                             \
    MUL(t, acc->zz, b->zz);                                                                      \
    MUL(acc->zz, t, pp);                                                                         \
    MUL(t, acc->zzz, b->zzz);                                                                    \
    MUL(acc->zzz, t, ppp);                                                                       \
  }                                                                                              \
  static void PFX##_to_aff(PFX##_aff* r, const PFX##_xyzz* p) {                                  \
    if (ISZ(p->zz)) {                                                                            \
      ZERO(r->x);                                                                                \
      ZERO(r->y);                                                                                \
      return;                                                                                    \
    }                                                                                            \
    T izz, izzz;                                                                                 \
    INV(izz, p->zz);                                                                             \
    INV(izzz, p->zzz);                                                                           \
    MUL(r->x, p->x, izz);                                                                        \
    MUL(r->y, p->y, izzz);                                                                       \
  }                                                                                              \
  /* k (canonical, little-endian 4x64) times q, MSB-first double-and-add */                      \
  static void PFX##_scalar_mul(PFX##_xyzz* r, const PFX##_aff* q, const uint64_t k[4]) {         \
    PFX##_xyzz acc, d;                                                                           \
    PFX##_set_inf(&acc);                                                                         \
    for (int i = 3; i >= 0; i--)                                                                 \
      for (int b = 63; b >= 0; b--) {                                                            \
        PFX##_dbl(&d, &acc);                                                                     \
        acc = d;                                                                                 \
        if ((k[i] >> b) & 1) PFX##_madd(&acc, q);                                                \
      }                                                                                          \
    *r = acc;                                                                                    \
  }                                                                                              \
  /* Pippenger, signed c-bit digits, one bucket set per window (gnark-crypto multiexp shape) */  \
  static void PFX##_msm(PFX##_aff* out, const PFX##_aff* bases, const uint64_t* scalars_mont,    \
                        size_t n, int c) {                                                       \
    int nwin = (256 + c - 1) / c + 1;                                                            \
    size_t nb = (size_t)1 << (c - 1);                                                            \
    int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * n * nwin);                              \
    for (size_t i = 0; i < n; i++) {                                                             \
      fe s;                                                                                      \
      fe_from_mont(s, scalars_mont + 4 * i, &FR);                                                \
      int carry = 0;                                                                             \
      for (int w = 0; w < nwin; w++) {                                                           \
        int bit = w * c;                                                                         \
        int64_t d = carry;                                                                       \
        if (bit < 256) {                                                                         \
          int limb = bit >> 6, off = bit & 63;                                                   \
          uint64_t v = s[limb] >> off;                                                           \
          if (off + c > 64 && limb < 3) v |= s[limb + 1] << (64 - off);                          \
          d += (int64_t)(v & (((uint64_t)1 << c) - 1));                                          \
        }                                                                                        \
        if (d > (int64_t)nb) {                                                                   \
          d -= ((int64_t)1 << c);                                                                \
          carry = 1;                                                                             \
        } else                                                                                   \
          carry = 0;                                                                             \
        digits[i * nwin + w] = (int32_t)d;                                                       \
      }                                                                                          \
    }                                                                                            \
    PFX##_xyzz total;                                                                            \
    PFX##_set_inf(&total);                                                                       \
    PFX##_xyzz* buckets = (PFX##_xyzz*)malloc(sizeof(PFX##_xyzz) * nb);                          \
    for (int w = nwin - 1; w >= 0; w--) {                                                        \
      for (int k = 0; k < c; k++) {                                                              \
        PFX##_xyzz d;                                                                            \
        PFX##_dbl(&d, &total);                                                                   \
        total = d;                                                                               \
      }                                                                                          \
      for (size_t b = 0; b < nb; b++) PFX##_set_inf(&buckets[b]);                                \
      for (size_t i = 0; i < n; i++) {                                                           \
        int32_t d = digits[i * nwin + w];                                                        \
        if (d > 0)                                                                               \
          PFX##_madd(&buckets[d - 1], &bases[i]);                                                \
        else if (d < 0) {                                                                        \
          PFX##_aff nq = bases[i];                                                               \
          NEG(nq.y, nq.y);                                                                       \
          PFX##_madd(&buckets[-d - 1], &nq);                                                     \
        }                                                                                        \
      }                                                                                          \
      PFX##_xyzz run, sum;                                                                       \
      PFX##_set_inf(&run);                                                                       \
      PFX##_set_inf(&sum);                                                                       \
      for (size_t b = nb; b-- > 0;) {                                                            \
        PFX##_padd(&run, &buckets[b]);                                                           \
        PFX##_padd(&sum, &run);                                                                  \
      }                                                                                          \
      PFX##_padd(&total, &sum);                                                                  \
    }                                                                                            \
    free(buckets);                                                                               \
    free(digits);                                                                                \
    PFX##_to_aff(out, &total);                                                                   \
  }

#define FQ_SET(r, a) fe_set(r, a)
#define FQ_ZERO(r) fe_zero(r)
#define FQ_ISZ(a) fe_is_zero(a)
#define FQ_EQ(a, b) fe_eq(a, b)
#define FQ_ADD(r, a, b) fe_add(r, a, b, &FQ)
#define FQ_SUB(r, a, b) fe_sub(r, a, b, &FQ)
#define FQ_NEG(r, a) fe_neg(r, a, &FQ)
#define FQ_MUL(r, a, b) fe_mul(r, a, b, &FQ)
#define FQ_SQR(r, a) fe_sqr(r, a, &FQ)
#define FQ_INV(r, a) fe_inv(r, a, &FQ)
#define FQ_ONE(r) fe_set(r, FQ.one)
DEFINE_GROUP(g1, fe, FQ_SET, FQ_ZERO, FQ_ISZ, FQ_EQ, FQ_ADD, FQ_SUB, FQ_NEG, FQ_MUL, FQ_SQR, FQ_INV,
             FQ_ONE)

#define F2_SET(r, a) ((r) = (a))
#define F2_ZERO(r) memset(&(r), 0, sizeof(fe2))
#define F2_ISZ(a) fe2_is_zero(&(a))
#define F2_EQ(a, b) (memcmp(&(a), &(b), sizeof(fe2)) == 0)
#define F2_ADD(r, a, b) fe2_add(&(r), &(a), &(b))
#define F2_SUB(r, a, b) fe2_sub(&(r), &(a), &(b))
#define F2_NEG(r, a) fe2_neg(&(r), &(a))
#define F2_MUL(r, a, b)      \
  do {                       \
    fe2 _t;                  \
    fe2_mul(&_t, &(a), &(b)); \
    (r) = _t;                \
  } while (0)
#define F2_SQR(r, a)    \
  do {                  \
    fe2 _t;             \
    fe2_sqr(&_t, &(a)); \
    (r) = _t;           \
  } while (0)
#define F2_INV(r, a)    \
  do {                  \
    fe2 _t;             \
    fe2_inv(&_t, &(a)); \
    (r) = _t;           \
  } while (0)
#define F2_ONE(r)               \
  do {                          \
    fe_set((r).c0, FQ.one);     \
    fe_zero((r).c1);            \
  } while (0)
DEFINE_GROUP(g2, fe2, F2_SET, F2_ZERO, F2_ISZ, F2_EQ, F2_ADD, F2_SUB, F2_NEG, F2_MUL, F2_SQR, F2_INV,
             F2_ONE)

/* ---------------------------------------------------------------- NTT --------------------- */
/* omega_{2^28} = 5^((r-1)/2^28), Montgomery form computed at first use */
static fe ROOT28;
static fe GEN5;
static int consts_ready = 0;
static void init_consts(void) {
  if (consts_ready) return;
  fe five = {5, 0, 0, 0};
  fe_to_mont(GEN5, five, &FR);
  /* (r-1)/2^28 */
  fe e, rm1;
  fe onei = {1, 0, 0, 0};
  fe_sub_raw(rm1, FR.p, onei);
  for (int i = 0; i < 4; i++) e[i] = rm1[i];
  for (int s = 0; s < 28; s++) {
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? (e[i + 1] << 63) : 0);
  }
  fe_pow(ROOT28, GEN5, e, &FR);
  consts_ready = 1;
}
static void root_of_unity(fe w, int log_n) {
  init_consts();
  fe_set(w, ROOT28);
  for (int i = 28; i > log_n; i--) fe_sqr(w, w, &FR);
}
static size_t bitrev(size_t x, int bits) {
  size_t r = 0;
  for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
static void bit_reverse_perm(fe* a, int log_n) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev(i, log_n);
    if (i < j) {
      fe t;
      fe_set(t, a[i]);
      fe_set(a[i], a[j]);
      fe_set(a[j], t);
    }
  }
}
/* in-place decimation-in-frequency: natural in, bit-reversed out (gnark fft.DIF) */
static void ntt_dif(fe* a, int log_n, const fe* tw /* n/2 powers of the root */) {
  size_t n = (size_t)1 << log_n;
  for (size_t m = n >> 1, step = 1; m >= 1; m >>= 1, step <<= 1) {
    for (size_t k = 0; k < n; k += 2 * m)
      for (size_t j = 0; j < m; j++) {
        fe u, v;
        fe_add(u, a[k + j], a[k + j + m], &FR);
        fe_sub(v, a[k + j], a[k + j + m], &FR);
        fe_mul(v, v, tw[j * step], &FR);
        fe_set(a[k + j], u);
        fe_set(a[k + j + m], v);
      }
    if (m == 1) break;
  }
}
/* in-place decimation-in-time: bit-reversed in, natural out (gnark fft.DIT) */
static void ntt_dit(fe* a, int log_n, const fe* tw) {
  size_t n = (size_t)1 << log_n;
  for (size_t m = 1, step = n >> 1; m < n; m <<= 1, step >>= 1)
    for (size_t k = 0; k < n; k += 2 * m)
      for (size_t j = 0; j < m; j++) {
        fe t, u;
        fe_mul(t, a[k + j + m], tw[j * step], &FR);
        fe_set(u, a[k + j]);
        fe_add(a[k + j], u, t, &FR);
        fe_sub(a[k + j + m], u, t, &FR);
      }
}
static fe* make_twiddles(int log_n, int inverse) {
  size_t n = (size_t)1 << log_n;
  fe w;
  root_of_unity(w, log_n);
  if (inverse) fe_inv(w, w, &FR);
  fe* tw = (fe*)malloc(sizeof(fe) * (n / 2 ? n / 2 : 1));
  fe_set(tw[0], FR.one);
  for (size_t i = 1; i < n / 2; i++) fe_mul(tw[i], tw[i - 1], w, &FR);
  return tw;
}

/* natural-order in, natural-order out.
 * forward: y[k] = sum_i x[i] (g w^k)^i   (g = 5 when coset, else 1)
 * inverse: x[i] = g^-i / n * sum_k y[k] w^-ik  */
void zkref_ntt(uint64_t* data, int log_n, int inverse, int coset) {
  init_consts();
  fe* a = (fe*)data;
  size_t n = (size_t)1 << log_n;
  fe* tw = make_twiddles(log_n, inverse);
  if (!inverse) {
    if (coset) {
      fe g;
      fe_set(g, FR.one);
      for (size_t i = 0; i < n; i++) {
        fe_mul(a[i], a[i], g, &FR);
        fe_mul(g, g, GEN5, &FR);
      }
    }
    ntt_dif(a, log_n, tw);
    bit_reverse_perm(a, log_n);
  } else {
    bit_reverse_perm(a, log_n);
    ntt_dit(a, log_n, tw);
    fe ninv, nn = {n, 0, 0, 0};
    fe_to_mont(nn, nn, &FR);
    fe_inv(ninv, nn, &FR);
    fe gi, g;
    fe_inv(gi, GEN5, &FR);
    fe_set(g, ninv);
    for (size_t i = 0; i < n; i++) {
      fe_mul(a[i], a[i], g, &FR);
      if (coset) fe_mul(g, g, gi, &FR);
    }
  }
  free(tw);
}

/* naive O(n^2) DFT for pinning the NTT (small n) */
void zkref_dft_naive(const uint64_t* in, uint64_t* out, int log_n, int inverse, int coset) {
  init_consts();
  size_t n = (size_t)1 << log_n;
  const fe* x = (const fe*)in;
  fe* y = (fe*)out;
  fe w;
  root_of_unity(w, log_n);
  if (inverse) fe_inv(w, w, &FR);
  for (size_t k = 0; k < n; k++) {
    fe wk, acc, pw;
    fe ek = {k, 0, 0, 0};
    fe_pow(wk, w, ek, &FR);
    if (!inverse && coset) fe_mul(wk, wk, GEN5, &FR);
    fe_zero(acc);
    fe_set(pw, FR.one);
    for (size_t i = 0; i < n; i++) {
      fe t;
      fe_mul(t, x[i], pw, &FR);
      fe_add(acc, acc, t, &FR);
      fe_mul(pw, pw, wk, &FR);
    }
    fe_set(y[k], acc);
  }
  if (inverse) {
    fe ninv, nn = {n, 0, 0, 0}, gi, g;
    fe_to_mont(nn, nn, &FR);
    fe_inv(ninv, nn, &FR);
    fe_inv(gi, GEN5, &FR);
    fe_set(g, ninv);
    for (size_t i = 0; i < n; i++) {
      fe_mul(y[i], y[i], g, &FR);
      if (coset) fe_mul(g, g, gi, &FR);
    }
  }
}

/* computeH: h = coefficients of (A*B - C)/Z on the size-n domain (n = 2^log_n), inputs are the
 * per-constraint evaluations a,b,c (length n, zero padded), output h[0..n) natural order.
 * Restates gnark backend/groth16/bn254/prove.go computeH. */
void zkref_compute_h(uint64_t* a, uint64_t* b, uint64_t* c, int log_n) {
  init_consts();
  size_t n = (size_t)1 << log_n;
  zkref_ntt(a, log_n, 1, 0);
  zkref_ntt(b, log_n, 1, 0);
  zkref_ntt(c, log_n, 1, 0);
  zkref_ntt(a, log_n, 0, 1);
  zkref_ntt(b, log_n, 0, 1);
  zkref_ntt(c, log_n, 0, 1);
  fe den, en = {n, 0, 0, 0};
  fe_pow(den, GEN5, en, &FR);
  fe_sub(den, den, FR.one, &FR);
  fe_inv(den, den, &FR);
  fe* A = (fe*)a;
  fe* B = (fe*)b;
  fe* C = (fe*)c;
  for (size_t i = 0; i < n; i++) {
    fe t;
    fe_mul(t, A[i], B[i], &FR);
    fe_sub(t, t, C[i], &FR);
    fe_mul(A[i], t, den, &FR);
  }
  zkref_ntt(a, log_n, 1, 1);
}

/* ---------------------------------------------------------------- exported helpers -------- */
void zkref_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_mul(r + 4 * i, a + 4 * i, b + 4 * i, &FR);
}
void zkref_fq_mul(const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_mul(r + 4 * i, a + 4 * i, b + 4 * i, &FQ);
}
void zkref_fr_add(const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_add(r + 4 * i, a + 4 * i, b + 4 * i, &FR);
}
void zkref_fr_sub(const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_sub(r + 4 * i, a + 4 * i, b + 4 * i, &FR);
}
void zkref_fr_inv(const uint64_t* a, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_inv(r + 4 * i, a + 4 * i, &FR);
}
void zkref_fr_to_mont(const uint64_t* a, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_to_mont(r + 4 * i, a + 4 * i, &FR);
}
void zkref_fr_from_mont(const uint64_t* a, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_from_mont(r + 4 * i, a + 4 * i, &FR);
}
void zkref_fq_to_mont(const uint64_t* a, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_to_mont(r + 4 * i, a + 4 * i, &FQ);
}
void zkref_fq_from_mont(const uint64_t* a, uint64_t* r, size_t n) {
  for (size_t i = 0; i < n; i++) fe_from_mont(r + 4 * i, a + 4 * i, &FQ);
}

/* MSM with Montgomery-form scalars and bases (gnark memory image); c = window bits (0: auto) */
static int auto_c(size_t n) {
  int c = 4;
  while (((size_t)1 << (c + 3)) < n && c < 16) c++;
  return c;
}
void zkref_msm_g1(const uint64_t* bases, const uint64_t* scalars, size_t n, int c, uint64_t* out) {
  if (c <= 0) c = auto_c(n);
  g1_msm((g1_aff*)out, (const g1_aff*)bases, scalars, n, c);
}
void zkref_msm_g2(const uint64_t* bases, const uint64_t* scalars, size_t n, int c, uint64_t* out) {
  if (c <= 0) c = auto_c(n);
  g2_msm((g2_aff*)out, (const g2_aff*)bases, scalars, n, c);
}
/* naive sum_i k_i * P_i by double-and-add (pins the Pippenger above) */
void zkref_msm_g1_naive(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t* out) {
  g1_xyzz acc;
  g1_set_inf(&acc);
  for (size_t i = 0; i < n; i++) {
    fe k;
    fe_from_mont(k, scalars + 4 * i, &FR);
    g1_xyzz t;
    g1_scalar_mul(&t, (const g1_aff*)bases + i, k);
    g1_padd(&acc, &t);
  }
  g1_to_aff((g1_aff*)out, &acc);
}
void zkref_msm_g2_naive(const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t* out) {
  g2_xyzz acc;
  g2_set_inf(&acc);
  for (size_t i = 0; i < n; i++) {
    fe k;
    fe_from_mont(k, scalars + 4 * i, &FR);
    g2_xyzz t;
    g2_scalar_mul(&t, (const g2_aff*)bases + i, k);
    g2_padd(&acc, &t);
  }
  g2_to_aff((g2_aff*)out, &acc);
}

/* out[i] = k_i * base, batch (setup helper; gnark: BatchScalarMultiplicationG1/G2).
 * Fixed-base 8-bit windows. Scalars Montgomery. */
void zkref_g1_batch_mul(const uint64_t* base, const uint64_t* scalars, size_t n, uint64_t* out) {
  const int c = 8, nw = 32;
  g1_aff* table = (g1_aff*)malloc(sizeof(g1_aff) * nw * 255);
  g1_xyzz cur = {{0}}, t;
  memcpy(&cur.x, base, 32);
  memcpy(&cur.y, base + 4, 32);
  fe_set(cur.zz, FQ.one);
  fe_set(cur.zzz, FQ.one);
  for (int w = 0; w < nw; w++) {
    g1_aff b0;
    g1_to_aff(&b0, &cur);
    g1_xyzz acc;
    g1_set_inf(&acc);
    for (int d = 1; d < 256; d++) {
      g1_madd(&acc, &b0);
      g1_to_aff(&table[w * 255 + d - 1], &acc);
    }
    for (int k = 0; k < c; k++) {
      g1_dbl(&t, &cur);
      cur = t;
    }
  }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    fe k;
    fe_from_mont(k, scalars + 4 * i, &FR);
    g1_xyzz acc;
    g1_set_inf(&acc);
    for (int w = 0; w < nw; w++) {
      int d = (int)((k[w / 8] >> ((w % 8) * 8)) & 0xff);
      if (d) g1_madd(&acc, &table[w * 255 + d - 1]);
    }
    g1_to_aff((g1_aff*)out + i, &acc);
  }
  free(table);
}
void zkref_g2_batch_mul(const uint64_t* base, const uint64_t* scalars, size_t n, uint64_t* out) {
  const int c = 8, nw = 32;
  g2_aff* table = (g2_aff*)malloc(sizeof(g2_aff) * nw * 255);
  g2_xyzz cur, t;
  memcpy(&cur.x, base, 64);
  memcpy(&cur.y, base + 8, 64);
  F2_ONE(cur.zz);
  F2_ONE(cur.zzz);
  for (int w = 0; w < nw; w++) {
    g2_aff b0;
    g2_to_aff(&b0, &cur);
    g2_xyzz acc;
    g2_set_inf(&acc);
    for (int d = 1; d < 256; d++) {
      g2_madd(&acc, &b0);
      g2_to_aff(&table[w * 255 + d - 1], &acc);
    }
    for (int k = 0; k < c; k++) {
      g2_dbl(&t, &cur);
      cur = t;
    }
  }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    fe k;
    fe_from_mont(k, scalars + 4 * i, &FR);
    g2_xyzz acc;
    g2_set_inf(&acc);
    for (int w = 0; w < nw; w++) {
      int d = (int)((k[w / 8] >> ((w % 8) * 8)) & 0xff);
      if (d) g2_madd(&acc, &table[w * 255 + d - 1]);
    }
    g2_to_aff((g2_aff*)out + i, &acc);
  }
  free(table);
}

/* a + b on affine points (for proof assembly checks) */
void zkref_g1_add(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  g1_xyzz acc;
  g1_set_inf(&acc);
  g1_madd(&acc, (const g1_aff*)a);
  g1_madd(&acc, (const g1_aff*)b);
  g1_to_aff((g1_aff*)out, &acc);
}
void zkref_g2_add(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  g2_xyzz acc;
  g2_set_inf(&acc);
  g2_madd(&acc, (const g2_aff*)a);
  g2_madd(&acc, (const g2_aff*)b);
  g2_to_aff((g2_aff*)out, &acc);
}

#include "zkref_prove.inc"
#include "zkref_plonk.inc"
