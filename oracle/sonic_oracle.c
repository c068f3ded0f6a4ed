/* CPU oracle (2 of 2): plain-C restatement of the sdiehl/sonic prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nothing under sonic_amd/ links or calls it.
 *
 * PARITY UNPINNED at the byte level: the reference holds no golden vectors for this path and
 * cannot be built here (see oracle/sonic_ref.py header).  This file is pinned against
 * oracle/sonic_ref.py (the literal big-integer restatement) through the fixtures committed under
 * tests/golden/, and against the reference's acceptance properties through the known-trapdoor
 * exponent identities (tests/test_oracle.py).
 *
 * Shape: 64-bit limbs with unsigned __int128 (the HIP product uses 32-bit limbs -- the two share
 * no code).  Two MSM flavours: `orc_msm_fold` is the reference's left fold of acc <> (P `mul` v)
 * (src/Sonic/CommitmentScheme.hs:25-29,45-48); `orc_msm_pippenger` is the threaded bucket method
 * used as the "cpu-opt" baseline.  Polynomials are dense coefficient
 * arrays over an exponent range [lo, lo+len); the reference's sparse (exponent, coeff) lists
 * (poly-0.4.0.0 VLaurent) are the same objects with zeros dropped.
 *
 * Reference sites followed: SRS.hs:27-43 (index maps), CommitmentScheme.hs:20-48 (shift, basis
 * choice, e'=0 hole, quotient), Constraints.hs:23-68, Utils.hs:17-27, Protocol.hs:53-109 (order
 * of operations and of random draws), Signature.hs:38-72.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ generic Montgomery field */
#define DEFINE_FIELD(F, N)                                                                       \
  typedef struct { u64 l[N]; } F##_t;                                                            \
  static F##_t F##_P, F##_R2, F##_ONE;                                                           \
  static u64 F##_INV;                                                                            \
  static inline int F##_is_zero(const F##_t *a) { u64 t = 0; for (int i = 0; i < N; i++) t |= a->l[i]; return t == 0; } \
  static inline int F##_eq(const F##_t *a, const F##_t *b) { u64 t = 0; for (int i = 0; i < N; i++) t |= a->l[i] ^ b->l[i]; return t == 0; } \
  static inline int F##_geq_raw(const u64 *a, const u64 *b) {                                    \
    for (int i = N - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }   \
    return 1; }                                                                                  \
  static inline void F##_sub_raw(u64 *r, const u64 *a, const u64 *b) {                           \
    u64 br = 0; for (int i = 0; i < N; i++) { u128 t = (u128)a[i] - b[i] - br; r[i] = (u64)t; br = (u64)(t >> 64) & 1; } } \
  static inline void F##_add(F##_t *r, const F##_t *a, const F##_t *b) {                         \
    u128 c = 0; u64 t[N];                                                                        \
    for (int i = 0; i < N; i++) { c += (u128)a->l[i] + b->l[i]; t[i] = (u64)c; c >>= 64; }        \
    if (F##_geq_raw(t, F##_P.l)) F##_sub_raw(r->l, t, F##_P.l); else memcpy(r->l, t, sizeof t); } \
  static inline void F##_sub(F##_t *r, const F##_t *a, const F##_t *b) {                         \
    u64 t[N]; u64 br = 0;                                                                        \
    for (int i = 0; i < N; i++) { u128 d = (u128)a->l[i] - b->l[i] - br; t[i] = (u64)d; br = (u64)(d >> 64) & 1; } \
    if (br) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)t[i] + F##_P.l[i]; t[i] = (u64)c; c >>= 64; } } \
    memcpy(r->l, t, sizeof t); }                                                                 \
  static inline void F##_neg(F##_t *r, const F##_t *a) {                                         \
    if (F##_is_zero(a)) { *r = *a; return; } F##_sub_raw(r->l, F##_P.l, a->l); }                 \
  static inline void F##_mul(F##_t *r, const F##_t *a, const F##_t *b) {                         \
    u64 t[N + 2]; memset(t, 0, sizeof t);                                                        \
    for (int i = 0; i < N; i++) {                                                                \
      u128 c = 0;                                                                                \
      for (int j = 0; j < N; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (u64)c; c >>= 64; } \
      c += t[N]; t[N] = (u64)c; t[N + 1] = (u64)(c >> 64);                                       \
      u64 m = t[0] * F##_INV;                                                                    \
      c = (u128)m * F##_P.l[0] + t[0]; c >>= 64;                                                 \
      for (int j = 1; j < N; j++) { c += (u128)m * F##_P.l[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; } \
      c += t[N]; t[N - 1] = (u64)c; t[N] = t[N + 1] + (u64)(c >> 64);                            \
    }                                                                                            \
    if (t[N] || F##_geq_raw(t, F##_P.l)) F##_sub_raw(r->l, t, F##_P.l); else memcpy(r->l, t, N * 8); } \
  static inline void F##_sqr(F##_t *r, const F##_t *a) { F##_mul(r, a, a); }                     \
  static inline void F##_to_mont(F##_t *r, const F##_t *a) { F##_mul(r, a, &F##_R2); }           \
  static inline void F##_from_mont(F##_t *r, const F##_t *a) { F##_t one; memset(&one, 0, sizeof one); one.l[0] = 1; F##_mul(r, a, &one); } \
  /* a^e for a multi-limb exponent (little-endian limbs), square-and-multiply MSB first */      \
  static void F##_pow_limbs(F##_t *r, const F##_t *a, const u64 *e, int nl) {                    \
    F##_t acc = F##_ONE; int started = 0;                                                        \
    for (int i = nl * 64 - 1; i >= 0; i--) {                                                     \
      if (started) F##_sqr(&acc, &acc);                                                          \
      if ((e[i / 64] >> (i % 64)) & 1) { F##_mul(&acc, &acc, a); started = 1; } }                \
    *r = acc; }                                                                                  \
  static void F##_inv(F##_t *r, const F##_t *a) { /* Fermat: a^(p-2) */                          \
    u64 e[N]; memcpy(e, F##_P.l, sizeof e); e[0] -= 2; F##_pow_limbs(r, a, e, N); }              \
  static void F##_init(const u64 *p) {                                                           \
    memcpy(F##_P.l, p, N * 8);                                                                   \
    u64 inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv; F##_INV = (u64)0 - inv;       \
    /* R mod p by doubling 1, 64N times; R^2 by 64N more */                                      \
    F##_t x; memset(&x, 0, sizeof x); x.l[0] = 1;                                                \
    for (int i = 0; i < 64 * N; i++) F##_add(&x, &x, &x);                                        \
    F##_ONE = x;                                                                                 \
    for (int i = 0; i < 64 * N; i++) F##_add(&x, &x, &x);                                        \
    F##_R2 = x; }

DEFINE_FIELD(fq, 6)
DEFINE_FIELD(fr, 4)

static const u64 Q_LIMBS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                               0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 R_LIMBS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                               0x73eda753299d7d48ULL};
static const u64 GX_LIMBS[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const u64 GY_LIMBS[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};

/* ------------------------------------------------------------------ G1: y^2 = x^3 + 4 */
typedef struct { fq_t x, y; } g1a_t;        /* affine, Montgomery; infinity = (0,0) */
typedef struct { fq_t x, y, z; } g1j_t;     /* Jacobian, z == 0 is infinity */
static g1a_t G1_GEN;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_all(void) {
  fq_init(Q_LIMBS); fr_init(R_LIMBS);
  fq_t t; memcpy(t.l, GX_LIMBS, 48); fq_to_mont(&G1_GEN.x, &t);
  memcpy(t.l, GY_LIMBS, 48); fq_to_mont(&G1_GEN.y, &t);
}
static inline void ensure_init(void) { pthread_once(&g_once, init_all); }

static inline int g1a_is_inf(const g1a_t *p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static inline void g1j_set_inf(g1j_t *p) { memset(p, 0, sizeof *p); p->x = fq_ONE; p->y = fq_ONE; }
static inline int g1j_is_inf(const g1j_t *p) { return fq_is_zero(&p->z); }

static void g1j_double(g1j_t *r, const g1j_t *p) { /* dbl-2009-l, a = 0 */
  if (g1j_is_inf(p) || fq_is_zero(&p->y)) { g1j_set_inf(r); return; }
  fq_t A, B, C, D, E, F, t, X3, Y3, Z3;
  fq_sqr(&A, &p->x); fq_sqr(&B, &p->y); fq_sqr(&C, &B);
  fq_add(&t, &p->x, &B); fq_sqr(&t, &t); fq_sub(&t, &t, &A); fq_sub(&t, &t, &C); fq_add(&D, &t, &t);
  fq_add(&E, &A, &A); fq_add(&E, &E, &A); fq_sqr(&F, &E);
  fq_sub(&X3, &F, &D); fq_sub(&X3, &X3, &D);
  fq_sub(&t, &D, &X3); fq_mul(&Y3, &E, &t);
  fq_add(&C, &C, &C); fq_add(&C, &C, &C); fq_add(&C, &C, &C); fq_sub(&Y3, &Y3, &C);
  fq_mul(&Z3, &p->y, &p->z); fq_add(&Z3, &Z3, &Z3);
  r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_add_affine(g1j_t *r, const g1j_t *p, const g1a_t *q) { /* madd-2007-bl */
  if (g1a_is_inf(q)) { *r = *p; return; }
  if (g1j_is_inf(p)) { r->x = q->x; r->y = q->y; r->z = fq_ONE; return; }
  fq_t Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
  fq_sqr(&Z1Z1, &p->z); fq_mul(&U2, &q->x, &Z1Z1);
  fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
  if (fq_eq(&U2, &p->x)) { if (fq_eq(&S2, &p->y)) g1j_double(r, p); else g1j_set_inf(r); return; }
  fq_sub(&H, &U2, &p->x); fq_sqr(&HH, &H); fq_add(&I, &HH, &HH); fq_add(&I, &I, &I);
  fq_mul(&J, &H, &I); fq_sub(&rr, &S2, &p->y); fq_add(&rr, &rr, &rr); fq_mul(&V, &p->x, &I);
  fq_sqr(&X3, &rr); fq_sub(&X3, &X3, &J); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
  fq_sub(&t, &V, &X3); fq_mul(&Y3, &rr, &t); fq_mul(&t, &p->y, &J); fq_add(&t, &t, &t); fq_sub(&Y3, &Y3, &t);
  fq_add(&Z3, &p->z, &H); fq_sqr(&Z3, &Z3); fq_sub(&Z3, &Z3, &Z1Z1); fq_sub(&Z3, &Z3, &HH);
  r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_add(g1j_t *r, const g1j_t *p, const g1j_t *q) { /* add-2007-bl */
  if (g1j_is_inf(q)) { *r = *p; return; }
  if (g1j_is_inf(p)) { *r = *q; return; }
  fq_t Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
  fq_sqr(&Z1Z1, &p->z); fq_sqr(&Z2Z2, &q->z);
  fq_mul(&U1, &p->x, &Z2Z2); fq_mul(&U2, &q->x, &Z1Z1);
  fq_mul(&S1, &p->y, &q->z); fq_mul(&S1, &S1, &Z2Z2);
  fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
  if (fq_eq(&U1, &U2)) { if (fq_eq(&S1, &S2)) g1j_double(r, p); else g1j_set_inf(r); return; }
  fq_sub(&H, &U2, &U1); fq_add(&I, &H, &H); fq_sqr(&I, &I); fq_mul(&J, &H, &I);
  fq_sub(&rr, &S2, &S1); fq_add(&rr, &rr, &rr); fq_mul(&V, &U1, &I);
  fq_sqr(&X3, &rr); fq_sub(&X3, &X3, &J); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
  fq_sub(&t, &V, &X3); fq_mul(&Y3, &rr, &t); fq_mul(&t, &S1, &J); fq_add(&t, &t, &t); fq_sub(&Y3, &Y3, &t);
  fq_add(&Z3, &p->z, &q->z); fq_sqr(&Z3, &Z3); fq_sub(&Z3, &Z3, &Z1Z1); fq_sub(&Z3, &Z3, &Z2Z2); fq_mul(&Z3, &Z3, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}
static void g1j_to_affine(g1a_t *r, const g1j_t *p) {
  if (g1j_is_inf(p)) { memset(r, 0, sizeof *r); return; }
  fq_t zi, zi2; fq_inv(&zi, &p->z); fq_sqr(&zi2, &zi);
  fq_mul(&r->x, &p->x, &zi2); fq_mul(&zi2, &zi2, &zi); fq_mul(&r->y, &p->y, &zi2);
}
static void g1a_neg(g1a_t *r, const g1a_t *p) { r->x = p->x; fq_neg(&r->y, &p->y); }

/* `mul`: MSB-first double-and-add on a standard-form 4-limb scalar */
static void g1_mul_raw(g1j_t *r, const g1a_t *p, const u64 *k) {
  g1j_t acc; g1j_set_inf(&acc);
  int started = 0;
  for (int i = 255; i >= 0; i--) {
    if (started) g1j_double(&acc, &acc);
    if ((k[i / 64] >> (i % 64)) & 1) { g1j_add_affine(&acc, &acc, p); started = 1; }
  }
  *r = acc;
}

/* ------------------------------------------------------------------ canonical byte encodings */
static int fr_from_bytes(fr_t *r, const uint8_t *b) { /* 32 B LE canonical -> Montgomery */
  fr_t t; memcpy(t.l, b, 32); if (fr_geq_raw(t.l, fr_P.l)) return -1; fr_to_mont(r, &t); return 0;
}
static void fr_to_bytes(uint8_t *b, const fr_t *a) { fr_t t; fr_from_mont(&t, a); memcpy(b, t.l, 32); }
static int g1a_from_bytes(g1a_t *r, const uint8_t *b) {
  fq_t x, y; memcpy(x.l, b, 48); memcpy(y.l, b + 48, 48);
  if (fq_geq_raw(x.l, fq_P.l) || fq_geq_raw(y.l, fq_P.l)) return -1;
  if (fq_is_zero(&x) && fq_is_zero(&y)) { memset(r, 0, sizeof *r); return 0; }
  fq_to_mont(&r->x, &x); fq_to_mont(&r->y, &y); return 0;
}
static void g1a_to_bytes(uint8_t *b, const g1a_t *p) {
  fq_t t; fq_from_mont(&t, &p->x); memcpy(b, t.l, 48); fq_from_mont(&t, &p->y); memcpy(b + 48, t.l, 48);
}

/* ------------------------------------------------------------------ threads helper */
typedef void (*range_fn)(void *ctx, long lo, long hi, int tid);
typedef struct { range_fn fn; void *ctx; long lo, hi; int tid; } range_job;
static void *range_tramp(void *p) { range_job *j = p; j->fn(j->ctx, j->lo, j->hi, j->tid); return NULL; }
static void parallel_for(long n, int threads, range_fn fn, void *ctx) {
  if (threads < 1) threads = 1;
  if (threads > 512) threads = 512;
  if (threads == 1 || n < 2) { fn(ctx, 0, n, 0); return; }
  pthread_t th[512]; range_job jobs[512];
  long per = (n + threads - 1) / threads;
  int used = 0;
  for (int t = 0; t < threads; t++) {
    long lo = t * per, hi = lo + per > n ? n : lo + per;
    if (lo >= hi) break;
    jobs[t] = (range_job){fn, ctx, lo, hi, t};
    pthread_create(&th[t], NULL, range_tramp, &jobs[t]); used++;
  }
  for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ MSM */
/* Reference-shaped: foldl' (\acc (e,v) -> acc <> (P `mul` v)) mempty  (CommitmentScheme.hs:25-29) */
static void msm_fold(g1j_t *out, const g1a_t *pts, const fr_t *scal_mont, long n) {
  g1j_t acc; g1j_set_inf(&acc);
  for (long i = 0; i < n; i++) {
    fr_t k; fr_from_mont(&k, &scal_mont[i]);
    g1j_t t; g1_mul_raw(&t, &pts[i], k.l);
    g1j_add(&acc, &acc, &t);
  }
  *out = acc;
}

typedef struct { const g1a_t *pts; const u64 *scal; long n; int c, W, S; g1j_t *win; } pip_ctx;
static inline unsigned get_bits(const u64 *k, int pos, int c) {
  if (pos >= 256) return 0;
  int li = pos / 64, sh = pos % 64;
  u64 v = k[li] >> sh;
  if (sh + c > 64 && li + 1 < 4) v |= k[li + 1] << (64 - sh);
  return (unsigned)(v & (((u64)1 << c) - 1));
}
/* task t = (window w, slice s): window sum of the slice's points */
static void pip_tasks(void *vctx, long tlo, long thi, int tid) {
  (void)tid;
  pip_ctx *x = vctx;
  long nb = (long)1 << x->c;
  g1j_t *buckets = malloc(sizeof(g1j_t) * nb);
  for (long t = tlo; t < thi; t++) {
    long w = t / x->S, s = t % x->S;
    long per = (x->n + x->S - 1) / x->S, lo = s * per, hi = lo + per > x->n ? x->n : lo + per;
    for (long b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
    for (long i = lo; i < hi; i++) {
      unsigned dgt = get_bits(x->scal + 4 * i, (int)w * x->c, x->c);
      if (dgt) g1j_add_affine(&buckets[dgt], &buckets[dgt], &x->pts[i]);
    }
    g1j_t run, sum; g1j_set_inf(&run); g1j_set_inf(&sum);
    for (long b = nb - 1; b >= 1; b--) { g1j_add(&run, &run, &buckets[b]); g1j_add(&sum, &sum, &run); }
    x->win[t] = sum;
  }
  free(buckets);
}
static void msm_pippenger(g1j_t *out, const g1a_t *pts, const fr_t *scal_mont, long n, int threads) {
  if (n <= 32) { msm_fold(out, pts, scal_mont, n); return; }
  if (threads < 1) threads = 1;
  int c0 = 4; while ((1L << (c0 + 5)) < n && c0 < 16) c0++;   /* c ~ log2(n) - 5, capped */
  int W0 = (255 + c0 - 1) / c0;
  int S = (threads + W0 - 1) / W0;                /* point slices per window so that W*S >= threads */
  long per = (n + S - 1) / S;
  int c = 4; while ((1L << (c + 5)) < per && c < 16) c++;
  int W = (255 + c - 1) / c;
  u64 *scal = malloc(32 * n);
  for (long i = 0; i < n; i++) { fr_t k; fr_from_mont(&k, &scal_mont[i]); memcpy(scal + 4 * i, k.l, 32); }
  g1j_t *win = malloc(sizeof(g1j_t) * W * S);
  pip_ctx ctx = {pts, scal, n, c, W, S, win};
  parallel_for((long)W * S, threads, pip_tasks, &ctx);
  g1j_t acc; g1j_set_inf(&acc);
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) g1j_double(&acc, &acc);
    for (int s = 0; s < S; s++) g1j_add(&acc, &acc, &win[w * S + s]);
  }
  *out = acc; free(win); free(scal);
}

static int g_msm_mode = 1;    /* 0 = reference-shaped fold, 1 = pippenger */
static int g_threads = 1;
static void msm(g1j_t *out, const g1a_t *pts, const fr_t *scal, long n) {
  if (g_msm_mode == 0) msm_fold(out, pts, scal, n); else msm_pippenger(out, pts, scal, n, g_threads);
}

/* ------------------------------------------------------------------ SRS (SRS.hs:27-43) */
typedef struct {
  long d;
  g1a_t *g;   /* g[e + d]  = g^{x^e},       e in [-d, d]  : gNegativeX[k] = g[-(k+1)], gPositiveX[k] = g[k] */
  g1a_t *ga;  /* ga[e + d] = g^{alpha x^e}, e in [-d, d]\{0}: gNegativeAlphaX[k] = ga[-(k+1)],
                 gPositiveAlphaX[k] = ga[k+1]; slot e = 0 is (0,0) -- g^alpha is not shared (SRS.hs:38) */
} srs_t;

/* fixed-base table: tab[w][j] = j * 2^(8w) * G, j in 1..255, affine */
static g1a_t *g_fb_table = NULL;
static pthread_mutex_t g_fb_mu = PTHREAD_MUTEX_INITIALIZER;
static void batch_to_affine(g1a_t *out, const g1j_t *in, long n) {
  fq_t *pref = malloc(sizeof(fq_t) * (n + 1));
  pref[0] = fq_ONE;
  for (long i = 0; i < n; i++) { if (g1j_is_inf(&in[i])) pref[i + 1] = pref[i]; else fq_mul(&pref[i + 1], &pref[i], &in[i].z); }
  fq_t inv; fq_inv(&inv, &pref[n]);
  for (long i = n - 1; i >= 0; i--) {
    if (g1j_is_inf(&in[i])) { memset(&out[i], 0, sizeof out[i]); continue; }
    fq_t zi, zi2; fq_mul(&zi, &inv, &pref[i]); fq_mul(&inv, &inv, &in[i].z);
    fq_sqr(&zi2, &zi); fq_mul(&out[i].x, &in[i].x, &zi2); fq_mul(&zi2, &zi2, &zi); fq_mul(&out[i].y, &in[i].y, &zi2);
  }
  free(pref);
}
static void fb_table_init(void) {
  pthread_mutex_lock(&g_fb_mu);
  if (!g_fb_table) {
    g1j_t *tj = malloc(sizeof(g1j_t) * 32 * 256);
    g1j_t base; base.x = G1_GEN.x; base.y = G1_GEN.y; base.z = fq_ONE;
    for (int w = 0; w < 32; w++) {
      g1j_set_inf(&tj[w * 256]);
      tj[w * 256 + 1] = base;
      for (int j = 2; j < 256; j++) g1j_add(&tj[w * 256 + j], &tj[w * 256 + j - 1], &base);
      for (int k = 0; k < 8; k++) g1j_double(&base, &base);
    }
    g1a_t *t = malloc(sizeof(g1a_t) * 32 * 256);
    batch_to_affine(t, tj, 32 * 256); free(tj);
    g_fb_table = t;
  }
  pthread_mutex_unlock(&g_fb_mu);
}
static void g1_gen_mul(g1j_t *r, const fr_t *k_mont) { /* `mul gen k` via the byte table */
  fr_t k; fr_from_mont(&k, k_mont);
  g1j_t acc; g1j_set_inf(&acc);
  for (int w = 0; w < 32; w++) {
    unsigned b = (unsigned)(k.l[w / 8] >> (8 * (w % 8))) & 0xff;
    if (b) g1j_add_affine(&acc, &acc, &g_fb_table[w * 256 + b]);
  }
  *r = acc;
}
typedef struct { srs_t *s; fr_t x, xinv, alpha; } srs_ctx;
static void srs_fill(void *vctx, long lo, long hi, int tid) {
  (void)tid;
  srs_ctx *c = vctx; long d = c->s->d;
  /* slots lo..hi of the 2d+1 exponents e = slot - d */
  long cnt = hi - lo;
  g1j_t *tj = malloc(sizeof(g1j_t) * 2 * cnt);
  long e0 = lo - d;
  fr_t p; /* x^e0 */
  { u64 ee[1]; long a = e0 < 0 ? -e0 : e0; ee[0] = (u64)a; fr_pow_limbs(&p, e0 < 0 ? &c->xinv : &c->x, ee, 1); }
  for (long i = 0; i < cnt; i++) {
    g1_gen_mul(&tj[i], &p);
    if (e0 + i == 0) g1j_set_inf(&tj[cnt + i]);
    else { fr_t ap; fr_mul(&ap, &p, &c->alpha); g1_gen_mul(&tj[cnt + i], &ap); }
    fr_mul(&p, &p, &c->x);
  }
  g1a_t *ta = malloc(sizeof(g1a_t) * 2 * cnt);
  batch_to_affine(ta, tj, 2 * cnt);
  memcpy(c->s->g + lo, ta, sizeof(g1a_t) * cnt);
  memcpy(c->s->ga + lo, ta + cnt, sizeof(g1a_t) * cnt);
  free(ta); free(tj);
}

/* ------------------------------------------------------------------ dense Laurent polynomials */
typedef struct { long lo, len; fr_t *c; } lpoly;  /* coefficient of X^e at c[e - lo] */
static lpoly lp_new(long lo, long len) { lpoly p = {lo, len, calloc(len > 0 ? len : 1, sizeof(fr_t))}; return p; }
static void lp_free(lpoly *p) { free(p->c); p->c = NULL; }
static void fr_pow_signed(fr_t *r, const fr_t *x, const fr_t *xinv, long e) {
  u64 ee[1]; ee[0] = (u64)(e < 0 ? -e : e); fr_pow_limbs(r, e < 0 ? xinv : x, ee, 1);
}
/* `eval` (poly): sum_e c_e z^e */
static void lp_eval(fr_t *out, const lpoly *p, const fr_t *z, const fr_t *zinv) {
  fr_t acc; memset(&acc, 0, sizeof acc);
  fr_t pw; fr_pow_signed(&pw, z, zinv, p->lo);
  for (long i = 0; i < p->len; i++) { fr_t t; fr_mul(&t, &p->c[i], &pw); fr_add(&acc, &acc, &t); fr_mul(&pw, &pw, z); }
  *out = acc;
}

/* Fr NTT (radix-2, in place, bit-reversal + DIT); omega = 7^((r-1)/2^k) */
static void ntt(fr_t *a, int logn, int inverse) {
  long n = 1L << logn;
  for (long i = 1, j = 0; i < n; i++) { long bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; } }
  fr_t seven, root; { fr_t s; memset(&s, 0, sizeof s); s.l[0] = 7; fr_to_mont(&seven, &s); }
  { u64 e[4]; memcpy(e, R_LIMBS, 32); e[0] -= 1; /* (r-1) >> logn */
    for (int k = 0; k < logn; k++) { for (int i = 0; i < 3; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63); e[3] >>= 1; }
    fr_pow_limbs(&root, &seven, e, 4); }
  if (inverse) fr_inv(&root, &root);
  for (int s = 1; s <= logn; s++) {
    long m = 1L << s;
    fr_t wm = root; for (int k = s; k < logn; k++) fr_sqr(&wm, &wm);
    for (long k = 0; k < n; k += m) {
      fr_t w = fr_ONE;
      for (long j = 0; j < m / 2; j++) {
        fr_t t, u = a[k + j]; fr_mul(&t, &w, &a[k + j + m / 2]);
        fr_add(&a[k + j], &u, &t); fr_sub(&a[k + j + m / 2], &u, &t);
        fr_mul(&w, &w, &wm);
      }
    }
  }
  if (inverse) { fr_t ninv, nn; memset(&nn, 0, sizeof nn); nn.l[0] = (u64)n; fr_to_mont(&nn, &nn); fr_inv(&ninv, &nn);
    for (long i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv); }
}
static lpoly lp_mul(const lpoly *a, const lpoly *b, int use_ntt) {
  lpoly r = lp_new(a->lo + b->lo, a->len + b->len - 1);
  if (!use_ntt) { /* schoolbook: the reference's sparse convolution */
    for (long i = 0; i < a->len; i++) { if (fr_is_zero(&a->c[i])) continue;
      for (long j = 0; j < b->len; j++) { fr_t t; fr_mul(&t, &a->c[i], &b->c[j]); fr_add(&r.c[i + j], &r.c[i + j], &t); } }
    return r;
  }
  int logn = 0; while ((1L << logn) < r.len) logn++;
  long n = 1L << logn;
  fr_t *fa = calloc(n, sizeof(fr_t)), *fb = calloc(n, sizeof(fr_t));
  memcpy(fa, a->c, sizeof(fr_t) * a->len); memcpy(fb, b->c, sizeof(fr_t) * b->len);
  ntt(fa, logn, 0); ntt(fb, logn, 0);
  for (long i = 0; i < n; i++) fr_mul(&fa[i], &fa[i], &fb[i]);
  ntt(fa, logn, 1);
  memcpy(r.c, fa, sizeof(fr_t) * r.len); free(fa); free(fb);
  return r;
}

/* ------------------------------------------------------------------ error convention */
enum { ORC_OK = 0, ORC_D_TOO_SMALL = 1, ORC_SRS_INDEX = 2, ORC_BAD_ENCODING = 3, ORC_INEXACT = 4 };

/* commitPoly (CommitmentScheme.hs:20-33): F = sum v_e * B[e + d - max], alpha basis */
static int commit_poly(g1j_t *out, const srs_t *s, long maxm, const lpoly *f) {
  long shift = s->d - maxm;
  /* trim zero ends (normalised sparse form has no zero coefficients) */
  long a = 0, b = f->len;
  while (a < b && fr_is_zero(&f->c[a])) a++;
  while (b > a && fr_is_zero(&f->c[b - 1])) b--;
  if (a == b) { g1j_set_inf(out); return ORC_OK; }
  long e0 = f->lo + a + shift, e1 = f->lo + b - 1 + shift;
  if (e0 < -s->d || e1 > s->d) return ORC_SRS_INDEX;             /* index past vector end */
  if (e0 <= 0 && e1 >= 0 && !fr_is_zero(&f->c[-shift - f->lo])) return ORC_SRS_INDEX; /* e' = 0: index -1 */
  msm(out, s->ga + (e0 + s->d), f->c + a, b - a);
  return ORC_OK;
}
/* openPoly (CommitmentScheme.hs:36-48) */
static int open_poly(fr_t *fz_out, g1j_t *w_out, const srs_t *s, const fr_t *z, const lpoly *f) {
  fr_t zinv; int has_neg = f->lo < 0;
  if (fr_is_zero(z)) { if (has_neg) return ORC_INEXACT; zinv = *z; } else fr_inv(&zinv, z);
  fr_t fz; lp_eval(&fz, f, z, &zinv);
  /* (f(X) - f(z)) / (X - z): shift to an ordinary polynomial g = X^-lo (f - f(z)), Horner from the top */
  long lo = f->lo < 0 ? f->lo : 0, hi = f->lo + f->len - 1; if (hi < 0) hi = 0;
  lpoly g = lp_new(lo, hi - lo + 1);
  memcpy(g.c + (f->lo - lo), f->c, sizeof(fr_t) * f->len);
  fr_sub(&g.c[0 - lo], &g.c[0 - lo], &fz);
  long D = g.len - 1;
  lpoly q = lp_new(lo, D > 0 ? D : 0);
  fr_t carry; memset(&carry, 0, sizeof carry);
  for (long k = D; k >= 1; k--) { fr_t t; fr_mul(&t, &carry, z); fr_add(&carry, &g.c[k], &t); q.c[k - 1] = carry; }
  { fr_t t, rem; fr_mul(&t, &carry, z); fr_add(&rem, &g.c[0], &t); if (!fr_is_zero(&rem)) { lp_free(&g); lp_free(&q); return ORC_INEXACT; } }
  lp_free(&g);
  *fz_out = fz;
  long a = 0, b = q.len;
  while (a < b && fr_is_zero(&q.c[a])) a++;
  while (b > a && fr_is_zero(&q.c[b - 1])) b--;
  if (a == b) { g1j_set_inf(w_out); lp_free(&q); return ORC_OK; }
  long e0 = q.lo + a, e1 = q.lo + b - 1;
  if (e0 < -s->d || e1 > s->d) { lp_free(&q); return ORC_SRS_INDEX; }
  msm(w_out, s->g + (e0 + s->d), q.c + a, b - a);
  lp_free(&q);
  return ORC_OK;
}

/* ------------------------------------------------------------------ exported API (ctypes) */
#define API __attribute__((visibility("default")))

API void orc_set_mode(int msm_mode, int threads) { g_msm_mode = msm_mode; g_threads = threads < 1 ? 1 : threads; }

API int orc_g1_mul(uint8_t *out96, const uint8_t *p96, const uint8_t *k32) {
  ensure_init();
  g1a_t p; fr_t k; if (g1a_from_bytes(&p, p96) || fr_from_bytes(&k, k32)) return ORC_BAD_ENCODING;
  fr_t ks; fr_from_mont(&ks, &k);
  g1j_t r; g1_mul_raw(&r, &p, ks.l); g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); return ORC_OK;
}
API int orc_g1_add(uint8_t *out96, const uint8_t *p96, const uint8_t *q96) {
  ensure_init();
  g1a_t p, q; if (g1a_from_bytes(&p, p96) || g1a_from_bytes(&q, q96)) return ORC_BAD_ENCODING;
  g1j_t r; g1j_set_inf(&r); g1j_add_affine(&r, &r, &p); g1j_add_affine(&r, &r, &q);
  g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); return ORC_OK;
}
API int orc_g1_gen(uint8_t *out96) { ensure_init(); g1a_to_bytes(out96, &G1_GEN); return ORC_OK; }
API int orc_g1_on_curve(const uint8_t *p96) {
  ensure_init(); g1a_t p; if (g1a_from_bytes(&p, p96)) return 0; if (g1a_is_inf(&p)) return 1;
  fq_t l, r, four; fq_sqr(&l, &p.y); fq_sqr(&r, &p.x); fq_mul(&r, &r, &p.x);
  fq_add(&four, &fq_ONE, &fq_ONE); fq_add(&four, &four, &four); fq_add(&r, &r, &four); return fq_eq(&l, &r);
}
/* mode: 0 fold (reference-shaped), 1 pippenger */
API int orc_msm(uint8_t *out96, const uint8_t *pts96, const uint8_t *scal32, long n, int mode, int threads) {
  ensure_init();
  g1a_t *pts = malloc(sizeof(g1a_t) * (n ? n : 1)); fr_t *sc = malloc(sizeof(fr_t) * (n ? n : 1));
  for (long i = 0; i < n; i++) if (g1a_from_bytes(&pts[i], pts96 + 96 * i) || fr_from_bytes(&sc[i], scal32 + 32 * i)) { free(pts); free(sc); return ORC_BAD_ENCODING; }
  g1j_t r; if (mode == 0) msm_fold(&r, pts, sc, n); else msm_pippenger(&r, pts, sc, n, threads);
  g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); free(pts); free(sc); return ORC_OK;
}
/* same, but on an SRS slice: basis 0 = g^{x^e}, 1 = g^{alpha x^e}; points e0..e0+n-1 */
API int orc_msm_srs(uint8_t *out96, const void *srs, int basis, long e0, const uint8_t *scal32, long n, int mode, int threads) {
  ensure_init(); const srs_t *s = srs;
  if (e0 < -s->d || e0 + n - 1 > s->d) return ORC_SRS_INDEX;
  fr_t *sc = malloc(sizeof(fr_t) * (n ? n : 1));
  for (long i = 0; i < n; i++) if (fr_from_bytes(&sc[i], scal32 + 32 * i)) { free(sc); return ORC_BAD_ENCODING; }
  const g1a_t *pts = (basis ? s->ga : s->g) + (e0 + s->d);
  g1j_t r; if (mode == 0) msm_fold(&r, pts, sc, n); else msm_pippenger(&r, pts, sc, n, threads);
  g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); free(sc); return ORC_OK;
}

API void *orc_srs_new(long d, const uint8_t *x32, const uint8_t *alpha32, int threads) {
  ensure_init(); fb_table_init();
  srs_ctx c; if (fr_from_bytes(&c.x, x32) || fr_from_bytes(&c.alpha, alpha32)) return NULL;
  if (fr_is_zero(&c.x)) return NULL;           /* recip 0 in SRS.hs:29 */
  fr_inv(&c.xinv, &c.x);
  srs_t *s = malloc(sizeof *s); s->d = d;
  s->g = malloc(sizeof(g1a_t) * (2 * d + 1)); s->ga = malloc(sizeof(g1a_t) * (2 * d + 1));
  c.s = s;
  parallel_for(2 * d + 1, threads, srs_fill, &c);
  return s;
}
/* the record constructor `SRS{..}` (SRS.hs:11-22) from canonical bytes, (2d+1) x 96 per basis: lets a CPU baseline run on an SRS
 * that was generated elsewhere (set-up is not part of prove()).  Every point must be canonical and on the curve. */
typedef struct { srs_t *s; const uint8_t *b0, *b1; int bad; } srs_pts_ctx;
static void srs_from_bytes_range(void *vctx, long lo, long hi, int tid) {
  (void)tid; srs_pts_ctx *c = vctx;
  for (long i = lo; i < hi; i++)
    if (g1a_from_bytes(&c->s->g[i], c->b0 + 96 * i) || g1a_from_bytes(&c->s->ga[i], c->b1 + 96 * i)) c->bad = 1;
}
API void *orc_srs_from_points(long d, const uint8_t *b0, const uint8_t *b1, int threads) {
  ensure_init();
  srs_t *s = malloc(sizeof *s); s->d = d;
  s->g = malloc(sizeof(g1a_t) * (2 * d + 1)); s->ga = malloc(sizeof(g1a_t) * (2 * d + 1));
  srs_pts_ctx c = {s, b0, b1, 0};
  parallel_for(2 * d + 1, threads, srs_from_bytes_range, &c);
  if (c.bad) { free(s->g); free(s->ga); free(s); return NULL; }
  return s;
}
API void orc_srs_free(void *srs) { srs_t *s = srs; if (!s) return; free(s->g); free(s->ga); free(s); }
API long orc_srs_d(const void *srs) { return ((const srs_t *)srs)->d; }
/* copy points e0..e0+n-1 of a basis as canonical bytes */
API int orc_srs_points(const void *srs, int basis, long e0, long n, uint8_t *out) {
  const srs_t *s = srs; if (e0 < -s->d || e0 + n - 1 > s->d) return ORC_SRS_INDEX;
  const g1a_t *pts = (basis ? s->ga : s->g) + (e0 + s->d);
  for (long i = 0; i < n; i++) g1a_to_bytes(out + 96 * i, &pts[i]);
  return ORC_OK;
}

static int lp_from_sparse(lpoly *out, long n, const int64_t *exps, const uint8_t *coeffs) {
  if (n == 0) { *out = lp_new(0, 0); return 0; }
  long lo = exps[0], hi = exps[0];
  for (long i = 1; i < n; i++) { if (exps[i] < lo) lo = exps[i]; if (exps[i] > hi) hi = exps[i]; }
  *out = lp_new(lo, hi - lo + 1);
  for (long i = 0; i < n; i++) { fr_t c; if (fr_from_bytes(&c, coeffs + 32 * i)) { lp_free(out); return -1; } fr_add(&out->c[exps[i] - lo], &out->c[exps[i] - lo], &c); }
  return 0;
}
API int orc_commit_poly(const void *srs, long maxm, long n, const int64_t *exps, const uint8_t *coeffs, uint8_t *out96) {
  ensure_init(); lpoly f; if (lp_from_sparse(&f, n, exps, coeffs)) return ORC_BAD_ENCODING;
  g1j_t r; int rc = commit_poly(&r, srs, maxm, &f); lp_free(&f); if (rc) return rc;
  g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); return ORC_OK;
}
API int orc_open_poly(const void *srs, const uint8_t *z32, long n, const int64_t *exps, const uint8_t *coeffs, uint8_t *out_fz32, uint8_t *out96) {
  ensure_init(); lpoly f; fr_t z; if (fr_from_bytes(&z, z32)) return ORC_BAD_ENCODING;
  if (lp_from_sparse(&f, n, exps, coeffs)) return ORC_BAD_ENCODING;
  g1j_t r; fr_t fz; int rc = open_poly(&fz, &r, srs, &z, &f); lp_free(&f); if (rc) return rc;
  g1a_t a; g1j_to_affine(&a, &r); g1a_to_bytes(out96, &a); fr_to_bytes(out_fz32, &fz); return ORC_OK;
}

/* Fr vector ops for kernel-level parity tests (canonical bytes in/out) */
API int orc_ntt(uint8_t *data32, int logn, int inverse) {
  ensure_init(); long n = 1L << logn; fr_t *a = malloc(sizeof(fr_t) * n);
  for (long i = 0; i < n; i++) if (fr_from_bytes(&a[i], data32 + 32 * i)) { free(a); return ORC_BAD_ENCODING; }
  ntt(a, logn, inverse);
  for (long i = 0; i < n; i++) fr_to_bytes(data32 + 32 * i, &a[i]);
  free(a); return ORC_OK;
}
API int orc_poly_mul(const uint8_t *a32, long na, const uint8_t *b32, long nb, uint8_t *out32, int use_ntt) {
  ensure_init(); lpoly a = lp_new(0, na), b = lp_new(0, nb);
  for (long i = 0; i < na; i++) if (fr_from_bytes(&a.c[i], a32 + 32 * i)) return ORC_BAD_ENCODING;
  for (long i = 0; i < nb; i++) if (fr_from_bytes(&b.c[i], b32 + 32 * i)) return ORC_BAD_ENCODING;
  lpoly r = lp_mul(&a, &b, use_ntt);
  for (long i = 0; i < r.len; i++) fr_to_bytes(out32 + 32 * i, &r.c[i]);
  lp_free(&a); lp_free(&b); lp_free(&r); return ORC_OK;
}
API int orc_fr_mul(uint8_t *out, const uint8_t *a, const uint8_t *b) { ensure_init(); fr_t x, y; if (fr_from_bytes(&x, a) || fr_from_bytes(&y, b)) return ORC_BAD_ENCODING; fr_mul(&x, &x, &y); fr_to_bytes(out, &x); return 0; }
API int orc_fr_inv(uint8_t *out, const uint8_t *a) { ensure_init(); fr_t x; if (fr_from_bytes(&x, a)) return ORC_BAD_ENCODING; fr_inv(&x, &x); fr_to_bytes(out, &x); return 0; }

/* ------------------------------------------------------------------ prove (Protocol.hs:47-109, Signature.hs:38-72) */
/* s(X,y) for the dense weights: exps [-n, 2n].
 * X^-i: u_i(y) = sum_q wL[q][i] y^{q+n}; X^i: v_i(y) (wR); X^{i+n}: w_i(y) = -y^i - y^-i + sum_q wO[q][i] y^{q+n}
 * (Constraints.hs:39-49 under evalY, Utils.hs:20-21) */
static lpoly s_of_y(long n, long Q, const fr_t *wL, const fr_t *wR, const fr_t *wO, const fr_t *y) {
  lpoly s = lp_new(-n, 3 * n + 1);
  fr_t yinv; if (fr_is_zero(y)) yinv = *y; else fr_inv(&yinv, y);
  fr_t *yq = malloc(sizeof(fr_t) * Q);            /* y^{n+q}, q = 1..Q */
  fr_t yn; { u64 e[1] = {(u64)n}; fr_pow_limbs(&yn, y, e, 1); }
  fr_t p = yn; for (long q = 0; q < Q; q++) { fr_mul(&p, &p, y); yq[q] = p; }
  fr_t yp = fr_ONE, ym = fr_ONE;
  for (long i = 1; i <= n; i++) {
    fr_mul(&yp, &yp, y); fr_mul(&ym, &ym, &yinv);
    fr_t u, v, w, t; memset(&u, 0, sizeof u); v = u; w = u;
    for (long q = 0; q < Q; q++) {
      fr_mul(&t, &wL[q * n + i - 1], &yq[q]); fr_add(&u, &u, &t);
      fr_mul(&t, &wR[q * n + i - 1], &yq[q]); fr_add(&v, &v, &t);
      fr_mul(&t, &wO[q * n + i - 1], &yq[q]); fr_add(&w, &w, &t);
    }
    fr_sub(&w, &w, &yp); fr_sub(&w, &w, &ym);
    s.c[-i + n] = u; s.c[i + n] = v; s.c[i + n + n] = w;
  }
  free(yq);
  return s;
}
/* s(u,Y): exps [-n, n+Q] (evalX, Utils.hs:17-18) */
static lpoly s_of_u(long n, long Q, const fr_t *wL, const fr_t *wR, const fr_t *wO, const fr_t *u) {
  lpoly s = lp_new(-n, 2 * n + Q + 1);
  fr_t uinv; fr_inv(&uinv, u);
  fr_t un; { u64 e[1] = {(u64)n}; fr_pow_limbs(&un, u, e, 1); }
  fr_t up = fr_ONE, um = fr_ONE;
  for (long i = 1; i <= n; i++) {
    fr_mul(&up, &up, u); fr_mul(&um, &um, &uinv);
    fr_t uin; fr_mul(&uin, &up, &un);               /* u^{i+n} */
    fr_t neg; fr_neg(&neg, &uin);
    s.c[-i + n] = neg; s.c[i + n] = neg;            /* -u^{i+n} (Y^-i + Y^i) */
    for (long q = 0; q < Q; q++) {
      fr_t t, acc = s.c[n + 1 + q + n];
      fr_mul(&t, &um, &wL[q * n + i - 1]); fr_add(&acc, &acc, &t);
      fr_mul(&t, &up, &wR[q * n + i - 1]); fr_add(&acc, &acc, &t);
      fr_mul(&t, &uin, &wO[q * n + i - 1]); fr_add(&acc, &acc, &t);
      s.c[n + 1 + q + n] = acc;
    }
  }
  return s;
}

static void put_g1(uint8_t **o, const g1j_t *p) { g1a_t a; g1j_to_affine(&a, p); g1a_to_bytes(*o, &a); *o += 96; }
static void put_fr(uint8_t **o, const fr_t *x) { fr_to_bytes(*o, x); *o += 32; }

API long orc_proof_size(long Q) { return (7 + 4 * Q) * 96 + (5 + 2 * Q) * 32; }

/* weights: dense Q x n row-major canonical Fr; transcript: 8+2Q Fr in draw order
 * cns[4], y, z, ys[Q], zs[Q], u, v.  use_ntt = 0 -> schoolbook t(X,y) product. */
API int orc_prove(const void *srs_v, long n, long Q, const uint8_t *wL8, const uint8_t *wR8, const uint8_t *wO8,
                  const uint8_t *cs8, const uint8_t *aL8, const uint8_t *aR8, const uint8_t *aO8,
                  const uint8_t *tr8, uint8_t *out, int use_ntt) {
  ensure_init(); const srs_t *srs = srs_v;
  if (srs->d < 7 * n) return ORC_D_TOO_SMALL;                                  /* Protocol.hs:54-55 */
  int rc = ORC_OK;
  fr_t *wL = malloc(sizeof(fr_t) * Q * n), *wR = malloc(sizeof(fr_t) * Q * n), *wO = malloc(sizeof(fr_t) * Q * n);
  fr_t *cs = malloc(sizeof(fr_t) * Q), *tr = malloc(sizeof(fr_t) * (8 + 2 * Q));
  lpoly r1 = lp_new(-2 * n - 4, 3 * n + 5);
#define RD(dst, src) if (fr_from_bytes(&(dst), (src))) { rc = ORC_BAD_ENCODING; }
  for (long i = 0; i < Q * n; i++) { RD(wL[i], wL8 + 32 * i); RD(wR[i], wR8 + 32 * i); RD(wO[i], wO8 + 32 * i); }
  for (long q = 0; q < Q; q++) RD(cs[q], cs8 + 32 * q);
  for (long i = 0; i < 8 + 2 * Q; i++) RD(tr[i], tr8 + 32 * i);
  /* r'(X,1): aL at X^i, aR at X^-i, aO at X^{-i-n}, c_{n+i} at X^{-2n-i} (Constraints.hs:23-31, Protocol.hs:58-62) */
  for (long i = 1; i <= n; i++) { RD(r1.c[i - r1.lo], aL8 + 32 * (i - 1)); RD(r1.c[-i - r1.lo], aR8 + 32 * (i - 1)); RD(r1.c[-i - n - r1.lo], aO8 + 32 * (i - 1)); }
  for (long i = 1; i <= 4; i++) r1.c[-2 * n - i - r1.lo] = tr[i - 1];
  if (rc) { free(wL); free(wR); free(wO); free(cs); free(tr); lp_free(&r1); return rc; }
  const fr_t *y = &tr[4], *z = &tr[5], *ys = &tr[6], *zs = &tr[6 + Q], *u = &tr[6 + 2 * Q], *v = &tr[7 + 2 * Q];
  uint8_t *o = out;
  g1j_t P; fr_t a, b, tz, szy;
  lpoly sy = {0, 0, NULL}, bb = {0, 0, NULL}, t = {0, 0, NULL}, su = {0, 0, NULL};
  g1j_t Rc, Tc, Wa, Wb, Wt;

  if ((rc = commit_poly(&Rc, srs, n, &r1))) goto done;                         /* Protocol.hs:63 */
  /* r'(X,y) + s(X,y) over [-2n-4, 2n] */
  sy = s_of_y(n, Q, wL, wR, wO, y);
  bb = lp_new(-2 * n - 4, 4 * n + 5);
  { fr_t yinv; if (fr_is_zero(y)) yinv = *y; else fr_inv(&yinv, y);
    fr_t pw; fr_pow_signed(&pw, y, &yinv, r1.lo);
    for (long i = 0; i < r1.len; i++) { fr_mul(&bb.c[i], &r1.c[i], &pw); fr_mul(&pw, &pw, y); } }   /* r(X,y): c_e y^e */
  for (long i = 0; i < sy.len; i++) fr_add(&bb.c[sy.lo + i - bb.lo], &bb.c[sy.lo + i - bb.lo], &sy.c[i]);
  t = lp_mul(&r1, &bb, use_ntt);                                               /* Constraints.hs:61 */
  { fr_t ky; memset(&ky, 0, sizeof ky); fr_t yn; { u64 e[1] = {(u64)n}; fr_pow_limbs(&yn, y, e, 1); }
    fr_t p = yn; for (long q = 0; q < Q; q++) { fr_t tt; fr_mul(&p, &p, y); fr_mul(&tt, &cs[q], &p); fr_add(&ky, &ky, &tt); }
    fr_sub(&t.c[0 - t.lo], &t.c[0 - t.lo], &ky); }                             /* - k(y) (Constraints.hs:65,67-68) */
  if ((rc = commit_poly(&Tc, srs, srs->d, &t))) goto done;                     /* Protocol.hs:73 */
  if ((rc = open_poly(&a, &Wa, srs, z, &r1))) goto done;                       /* :79 */
  { fr_t yz; fr_mul(&yz, y, z); if ((rc = open_poly(&b, &Wb, srs, &yz, &r1))) goto done; }   /* :80 */
  if ((rc = open_poly(&tz, &Wt, srs, z, &t))) goto done;                       /* :81 */
  { fr_t zinv; if (fr_is_zero(z)) zinv = *z; else fr_inv(&zinv, z); lp_eval(&szy, &sy, z, &zinv); }  /* :83 */
  put_g1(&o, &Rc); put_g1(&o, &Tc); put_fr(&o, &a); put_g1(&o, &Wa); put_fr(&o, &b); put_g1(&o, &Wb); put_g1(&o, &Wt); put_fr(&o, &szy);
  /* hscProve (Signature.hs:38-72) */
  for (long j = 0; j < Q; j++) {                                               /* :40-45 */
    lpoly sj = s_of_y(n, Q, wL, wR, wO, &ys[j]); fr_t sjz;
    if ((rc = commit_poly(&P, srs, srs->d, &sj))) { lp_free(&sj); goto done; }
    put_g1(&o, &P);
    if ((rc = open_poly(&sjz, &P, srs, &zs[j], &sj))) { lp_free(&sj); goto done; }
    put_fr(&o, &sjz); put_g1(&o, &P); lp_free(&sj);
  }
  su = s_of_u(n, Q, wL, wR, wO, u);                                            /* :51 */
  g1j_t Cc; if ((rc = commit_poly(&Cc, srs, srs->d, &su))) goto done;          /* :52 */
  for (long j = 0; j < Q; j++) {                                               /* :53-57 */
    lpoly sj = s_of_y(n, Q, wL, wR, wO, &ys[j]); fr_t tmp, sjp; g1j_t Wp, Qj;
    rc = open_poly(&tmp, &Wp, srs, u, &sj); lp_free(&sj); if (rc) goto done;
    if ((rc = open_poly(&sjp, &Qj, srs, &ys[j], &su))) goto done;
    put_fr(&o, &sjp); put_g1(&o, &Wp); put_g1(&o, &Qj);
  }
  { fr_t tmp; g1j_t Qv; if ((rc = open_poly(&tmp, &Qv, srs, v, &su))) goto done;   /* :63 */
    put_g1(&o, &Qv); put_g1(&o, &Cc); put_fr(&o, u); put_fr(&o, v); }
done:
  lp_free(&r1); lp_free(&sy); lp_free(&bb); lp_free(&t); lp_free(&su);
  free(wL); free(wR); free(wO); free(cs); free(tr);
  return rc;
}
