/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.h header comment).
 * BN254 field / group / MSM / NTT restatement.  PARITY UNPINNED vs the Rust reference;
 * follows the public algorithms named at each function.
 */
#include "bn254.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

mont_ctx ORC_FR, ORC_FQ;
static int g_init_done = 0;
static fr_t FR_ROOT_OF_UNITY; /* order 2^28 */
static fr_t FR_ZETA;
static g1_affine G1_GEN;

/* ------------------------------------------------------------------ u256 */
int u256_cmp(const u256 *a, const u256 *b) {
  for (int i = 3; i >= 0; i--) {
    if (a->l[i] < b->l[i]) return -1;
    if (a->l[i] > b->l[i]) return 1;
  }
  return 0;
}
int u256_is_zero(const u256 *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
uint64_t u256_add(u256 *o, const u256 *a, const u256 *b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a->l[i] + b->l[i];
    o->l[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
uint64_t u256_sub(u256 *o, const u256 *a, const u256 *b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - borrow;
    o->l[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
void u256_shr(u256 *o, const u256 *a, unsigned s) {
  u256 t = {{0, 0, 0, 0}};
  if (s < 256) {
    unsigned w = s / 64, b = s % 64;
    for (unsigned i = 0; i + w < 4; i++) {
      t.l[i] = a->l[i + w] >> b;
      if (b && i + w + 1 < 4) t.l[i] |= a->l[i + w + 1] << (64 - b);
    }
  }
  *o = t;
}
void u256_shl(u256 *o, const u256 *a, unsigned s) {
  u256 t = {{0, 0, 0, 0}};
  if (s < 256) {
    unsigned w = s / 64, b = s % 64;
    for (int i = 3; i >= (int)w; i--) {
      t.l[i] = a->l[i - w] << b;
      if (b && i - (int)w - 1 >= 0) t.l[i] |= a->l[i - w - 1] >> (64 - b);
    }
  }
  *o = t;
}
unsigned u256_bits(const u256 *a) {
  for (int i = 3; i >= 0; i--)
    if (a->l[i]) return 64 * i + (64 - __builtin_clzll(a->l[i]));
  return 0;
}
void u256_set_u64(u256 *o, uint64_t v) {
  o->l[0] = v;
  o->l[1] = o->l[2] = o->l[3] = 0;
}
/* schoolbook shift-subtract; semantics of num_integer::Integer::div_mod_floor on BigUint */
void u256_divmod(u256 *q, u256 *r, const u256 *a, const u256 *b) {
  u256 quo = {{0, 0, 0, 0}}, rem = {{0, 0, 0, 0}};
  unsigned nb = u256_bits(a);
  for (int i = (int)nb - 1; i >= 0; i--) {
    uint64_t top = rem.l[3] >> 63;
    u256_shl(&rem, &rem, 1);
    rem.l[0] |= (a->l[i / 64] >> (i % 64)) & 1;
    if (top || u256_cmp(&rem, b) >= 0) {
      u256_sub(&rem, &rem, b);
      quo.l[i / 64] |= 1ULL << (i % 64);
    }
  }
  if (q) *q = quo;
  if (r) *r = rem;
}
void u256_mul_wide(uint64_t out[8], const u256 *a, const u256 *b) {
  memset(out, 0, 8 * sizeof(uint64_t));
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[i] * b->l[j] + out[i + j];
      out[i + j] = (uint64_t)c;
      c >>= 64;
    }
    out[i + 4] = (uint64_t)c;
  }
}

/* ------------------------------------------------------------------ Montgomery */
/* CIOS Montgomery multiplication (Koc et al.), R = 2^256 */
void mont_mul(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[i] * b->l[j] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t mm = t[0] * m->inv;
    c = (u128)mm * m->p.l[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)mm * m->p.l[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  u256 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || u256_cmp(&r, &m->p) >= 0) u256_sub(&r, &r, &m->p);
  *o = r;
}
void mont_add(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m) {
  u256 r;
  uint64_t c = u256_add(&r, a, b);
  if (c || u256_cmp(&r, &m->p) >= 0) u256_sub(&r, &r, &m->p);
  *o = r;
}
void mont_sub(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m) {
  u256 r;
  if (u256_sub(&r, a, b)) u256_add(&r, &r, &m->p);
  *o = r;
}
void mont_neg(u256 *o, const u256 *a, const mont_ctx *m) {
  if (u256_is_zero(a)) {
    *o = *a;
    return;
  }
  u256_sub(o, &m->p, a);
}
void mont_from_canonical(u256 *o, const u256 *a, const mont_ctx *m) {
  u256 t = *a;
  if (u256_cmp(&t, &m->p) >= 0) u256_divmod(NULL, &t, &t, &m->p);
  mont_mul(o, &t, &m->r2, m);
}
void mont_to_canonical(u256 *o, const u256 *a, const mont_ctx *m) {
  u256 one = {{1, 0, 0, 0}};
  mont_mul(o, a, &one, m);
}
void mont_pow(u256 *o, const u256 *a, const u256 *e, const mont_ctx *m) {
  u256 acc = m->r1, base = *a;
  unsigned nb = u256_bits(e);
  for (unsigned i = 0; i < nb; i++) {
    if ((e->l[i / 64] >> (i % 64)) & 1) mont_mul(&acc, &acc, &base, m);
    mont_mul(&base, &base, &base, m);
  }
  *o = acc;
}
/* Fermat inversion a^(p-2); 0 -> 0 */
void mont_inv(u256 *o, const u256 *a, const mont_ctx *m) {
  u256 e, two = {{2, 0, 0, 0}};
  u256_sub(&e, &m->p, &two);
  mont_pow(o, a, &e, m);
}

static void mont_ctx_setup(mont_ctx *m, const uint64_t p[4]) {
  memcpy(m->p.l, p, 32);
  /* inv = -p^{-1} mod 2^64 by Newton iteration */
  uint64_t x = 1;
  for (int i = 0; i < 7; i++) x *= 2 - p[0] * x;
  m->inv = (uint64_t)0 - x;
  /* r1 = 2^256 mod p by 256 modular doublings of 1; r2 = 2^512 mod p */
  u256 t = {{1, 0, 0, 0}};
  for (int i = 0; i < 512; i++) {
    uint64_t top = t.l[3] >> 63;
    u256_shl(&t, &t, 1);
    if (top || u256_cmp(&t, &m->p) >= 0) u256_sub(&t, &t, &m->p);
    if (i == 255) m->r1 = t;
  }
  m->r2 = t;
}

void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { mont_mul(o, a, b, &ORC_FR); }
void fr_add(fr_t *o, const fr_t *a, const fr_t *b) { mont_add(o, a, b, &ORC_FR); }
void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { mont_sub(o, a, b, &ORC_FR); }
void fr_neg(fr_t *o, const fr_t *a) { mont_neg(o, a, &ORC_FR); }
void fr_inv(fr_t *o, const fr_t *a) { mont_inv(o, a, &ORC_FR); }
void fr_from_u64(fr_t *o, uint64_t v) {
  u256 t;
  u256_set_u64(&t, v);
  mont_from_canonical(o, &t, &ORC_FR);
}
void fr_from_canonical(fr_t *o, const u256 *a) { mont_from_canonical(o, a, &ORC_FR); }
void fr_to_canonical(u256 *o, const fr_t *a) { mont_to_canonical(o, a, &ORC_FR); }
int fr_eq(const fr_t *a, const fr_t *b) { return u256_cmp(a, b) == 0; }
int fr_is_zero(const fr_t *a) { return u256_is_zero(a); }

void orc_fr_mul_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n) {
  for (size_t i = 0; i < n; i++) fr_mul(&o[i], &a[i], &b[i]);
}
void orc_fr_add_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n) {
  for (size_t i = 0; i < n; i++) fr_add(&o[i], &a[i], &b[i]);
}
void orc_fr_sub_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n) {
  for (size_t i = 0; i < n; i++) fr_sub(&o[i], &a[i], &b[i]);
}
void orc_fr_inv_batch(fr_t *o, const fr_t *a, size_t n) {
  for (size_t i = 0; i < n; i++) fr_inv(&o[i], &a[i]);
}
/* halo2 arithmetic::eval_polynomial: sum_i a[i] x^i by Horner's rule */
void orc_eval_poly(fr_t *out, const fr_t *a, size_t n, const fr_t *x) {
  fr_t acc;
  memset(&acc, 0, sizeof(acc));
  for (size_t i = n; i > 0; i--) {
    fr_mul(&acc, &acc, x);
    fr_add(&acc, &acc, &a[i - 1]);
  }
  *out = acc;
}
/* Grand product of the permutation / lookup arguments (halo2 plonk/permutation/prover.rs and
 * plonk/lookup/prover.rs, [UPSTREAM-RECALL]): z[0] = 1, z[i + 1] = z[i] * num[i] / den[i].  Denominators are inverted the
 * way halo2's batch_invert does: a zero stays zero, so the product is zero from there on.  One column of n entries. */
void orc_grand_product(fr_t *z, const fr_t *num, const fr_t *den, size_t n) {
  if (n == 0) return;
  /* the n - 1 denominators inverted with ONE field inversion (Montgomery's trick, what halo2's batch_invert does), zeros skipped */
  fr_t *pre = (fr_t *)malloc(sizeof(fr_t) * n);
  fr_t acc, t;
  fr_from_u64(&acc, 1);
  for (size_t i = 0; i + 1 < n; i++) {
    pre[i] = acc;
    if (!fr_is_zero(&den[i])) fr_mul(&acc, &acc, &den[i]);
  }
  fr_inv(&acc, &acc);
  for (size_t i = n - 1; i-- > 0;) {
    if (fr_is_zero(&den[i])) {
      memset(&pre[i], 0, sizeof(fr_t));
    } else {
      fr_mul(&t, &acc, &pre[i]);
      fr_mul(&acc, &acc, &den[i]);
      pre[i] = t; /* 1 / den[i] */
    }
  }
  fr_from_u64(&acc, 1);
  for (size_t i = 0; i < n; i++) {
    z[i] = acc;
    if (i + 1 == n) break;
    fr_mul(&t, &num[i], &pre[i]);
    fr_mul(&acc, &acc, &t);
  }
  free(pre);
}
void orc_fr_from_canonical_batch(fr_t *o, const u256 *a, size_t n) {
  for (size_t i = 0; i < n; i++) fr_from_canonical(&o[i], &a[i]);
}
void orc_fr_to_canonical_batch(u256 *o, const fr_t *a, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_canonical(&o[i], &a[i]);
}
void orc_fq_from_canonical_batch(fq_t *o, const u256 *a, size_t n) {
  for (size_t i = 0; i < n; i++) mont_from_canonical(&o[i], &a[i], &ORC_FQ);
}
void orc_fq_to_canonical_batch(u256 *o, const fq_t *a, size_t n) {
  for (size_t i = 0; i < n; i++) mont_to_canonical(&o[i], &a[i], &ORC_FQ);
}
void orc_fr_root_of_unity(fr_t *o, unsigned k) {
  orc_init();
  fr_t w = FR_ROOT_OF_UNITY;
  for (unsigned i = k; i < 28; i++) fr_mul(&w, &w, &w);
  *o = w;
}
void orc_fr_zeta(fr_t *o) {
  orc_init();
  *o = FR_ZETA;
}

/* ------------------------------------------------------------------ G1 */
#define FQM(o, a, b) mont_mul((o), (a), (b), &ORC_FQ)
#define FQA(o, a, b) mont_add((o), (a), (b), &ORC_FQ)
#define FQS(o, a, b) mont_sub((o), (a), (b), &ORC_FQ)

void g1_set_identity(g1_jac *o) { memset(o, 0, sizeof(*o)); }
int g1_is_identity(const g1_jac *a) { return u256_is_zero(&a->z); }
static int g1_affine_is_identity(const g1_affine *a) { return u256_is_zero(&a->x) && u256_is_zero(&a->y); }
void g1_from_affine(g1_jac *o, const g1_affine *a) {
  if (g1_affine_is_identity(a)) {
    g1_set_identity(o);
    return;
  }
  o->x = a->x;
  o->y = a->y;
  o->z = ORC_FQ.r1;
}
void g1_to_affine(g1_affine *o, const g1_jac *a) {
  if (g1_is_identity(a)) {
    memset(o, 0, sizeof(*o));
    return;
  }
  fq_t zi, zi2, zi3;
  mont_inv(&zi, &a->z, &ORC_FQ);
  FQM(&zi2, &zi, &zi);
  FQM(&zi3, &zi2, &zi);
  FQM(&o->x, &a->x, &zi2);
  FQM(&o->y, &a->y, &zi3);
}
/* dbl-2009-l (a = 0) */
void g1_double(g1_jac *o, const g1_jac *p) {
  if (g1_is_identity(p)) {
    *o = *p;
    return;
  }
  fq_t a, b, c, d, e, f, t, x3, y3, z3;
  FQM(&a, &p->x, &p->x);
  FQM(&b, &p->y, &p->y);
  FQM(&c, &b, &b);
  FQA(&t, &p->x, &b);
  FQM(&t, &t, &t);
  FQS(&t, &t, &a);
  FQS(&t, &t, &c);
  FQA(&d, &t, &t);
  FQA(&e, &a, &a);
  FQA(&e, &e, &a);
  FQM(&f, &e, &e);
  FQS(&x3, &f, &d);
  FQS(&x3, &x3, &d);
  FQS(&t, &d, &x3);
  FQM(&y3, &e, &t);
  FQA(&c, &c, &c);
  FQA(&c, &c, &c);
  FQA(&c, &c, &c);
  FQS(&y3, &y3, &c);
  FQM(&z3, &p->y, &p->z);
  FQA(&z3, &z3, &z3);
  o->x = x3;
  o->y = y3;
  o->z = z3;
}
/* add-2007-bl with exceptional cases handled */
void g1_add(g1_jac *o, const g1_jac *p, const g1_jac *q) {
  if (g1_is_identity(p)) {
    *o = *q;
    return;
  }
  if (g1_is_identity(q)) {
    *o = *p;
    return;
  }
  fq_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
  FQM(&z1z1, &p->z, &p->z);
  FQM(&z2z2, &q->z, &q->z);
  FQM(&u1, &p->x, &z2z2);
  FQM(&u2, &q->x, &z1z1);
  FQM(&s1, &p->y, &q->z);
  FQM(&s1, &s1, &z2z2);
  FQM(&s2, &q->y, &p->z);
  FQM(&s2, &s2, &z1z1);
  if (u256_cmp(&u1, &u2) == 0) {
    if (u256_cmp(&s1, &s2) == 0) {
      g1_double(o, p);
    } else {
      g1_set_identity(o);
    }
    return;
  }
  FQS(&h, &u2, &u1);
  FQA(&i, &h, &h);
  FQM(&i, &i, &i);
  FQM(&j, &h, &i);
  FQS(&r, &s2, &s1);
  FQA(&r, &r, &r);
  FQM(&v, &u1, &i);
  FQM(&x3, &r, &r);
  FQS(&x3, &x3, &j);
  FQS(&x3, &x3, &v);
  FQS(&x3, &x3, &v);
  FQS(&t, &v, &x3);
  FQM(&y3, &r, &t);
  FQM(&t, &s1, &j);
  FQA(&t, &t, &t);
  FQS(&y3, &y3, &t);
  FQA(&z3, &p->z, &q->z);
  FQM(&z3, &z3, &z3);
  FQS(&z3, &z3, &z1z1);
  FQS(&z3, &z3, &z2z2);
  FQM(&z3, &z3, &h);
  o->x = x3;
  o->y = y3;
  o->z = z3;
}
void g1_add_mixed(g1_jac *o, const g1_jac *p, const g1_affine *q) {
  g1_jac qj;
  g1_from_affine(&qj, q);
  g1_add(o, p, &qj);
}
void g1_neg_affine(g1_affine *o, const g1_affine *a) {
  o->x = a->x;
  mont_neg(&o->y, &a->y, &ORC_FQ);
}
void g1_scalar_mul(g1_jac *o, const g1_affine *p, const u256 *k) {
  g1_jac acc;
  g1_set_identity(&acc);
  unsigned nb = u256_bits(k);
  for (int i = (int)nb - 1; i >= 0; i--) {
    g1_double(&acc, &acc);
    if ((k->l[i / 64] >> (i % 64)) & 1) g1_add_mixed(&acc, &acc, p);
  }
  *o = acc;
}
int g1_affine_on_curve(const g1_affine *a) {
  if (g1_affine_is_identity(a)) return 1;
  fq_t y2, x3, three;
  u256 t3 = {{3, 0, 0, 0}};
  mont_from_canonical(&three, &t3, &ORC_FQ);
  FQM(&y2, &a->y, &a->y);
  FQM(&x3, &a->x, &a->x);
  FQM(&x3, &x3, &a->x);
  FQA(&x3, &x3, &three);
  return u256_cmp(&y2, &x3) == 0;
}
void orc_g1_generator(g1_affine *o) {
  orc_init();
  *o = G1_GEN;
}

/* fixed-base table for the generator: tab[w][d-1] = d * 2^(8w) * G, d in 1..255 */
static g1_affine *g_gen_tab = NULL;
static pthread_mutex_t g_tab_mu = PTHREAD_MUTEX_INITIALIZER;
static void gen_table_build(void) {
  pthread_mutex_lock(&g_tab_mu);
  if (!g_gen_tab) {
    g1_affine *tab = (g1_affine *)malloc(sizeof(g1_affine) * 32 * 255);
    g1_jac base;
    g1_from_affine(&base, &G1_GEN);
    for (int w = 0; w < 32; w++) {
      g1_jac acc = base;
      for (int d = 1; d <= 255; d++) {
        g1_to_affine(&tab[w * 255 + d - 1], &acc);
        g1_add(&acc, &acc, &base);
      }
      base = acc; /* 256 * previous base */
    }
    g_gen_tab = tab;
  }
  pthread_mutex_unlock(&g_tab_mu);
}
static void gen_mul(g1_jac *o, const u256 *k) {
  g1_jac acc;
  g1_set_identity(&acc);
  for (int w = 0; w < 32; w++) {
    unsigned d = (unsigned)((k->l[w / 8] >> (8 * (w % 8))) & 0xff);
    if (d) g1_add_mixed(&acc, &acc, &g_gen_tab[w * 255 + d - 1]);
  }
  *o = acc;
}
void orc_g1_mul_generator_batch(g1_affine *out, const u256 *s, size_t n) {
  orc_init();
  gen_table_build();
  for (size_t i = 0; i < n; i++) {
    g1_jac t;
    gen_mul(&t, &s[i]);
    g1_to_affine(&out[i], &t);
  }
}
/* Test SRS with a KNOWN tau (same construction as halo2 ParamsKZG::setup, which draws tau
 * from an RNG): g[i] = tau^i G; g_lagrange[i] = L_i(tau) G with
 * L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i)). */
void orc_srs_from_tau(g1_affine *g, g1_affine *gl, unsigned k, const u256 *tau_c) {
  orc_init();
  gen_table_build();
  size_t n = (size_t)1 << k;
  fr_t tau, acc = ORC_FR.r1, omega, tn, nf, wi = ORC_FR.r1;
  fr_from_canonical(&tau, tau_c);
  for (size_t i = 0; i < n; i++) {
    u256 c;
    fr_to_canonical(&c, &acc);
    g1_jac t;
    gen_mul(&t, &c);
    g1_to_affine(&g[i], &t);
    fr_mul(&acc, &acc, &tau);
  }
  tn = acc; /* tau^n */
  fr_sub(&tn, &tn, &ORC_FR.r1);
  orc_fr_root_of_unity(&omega, k);
  fr_from_u64(&nf, (uint64_t)n);
  for (size_t i = 0; i < n; i++) {
    fr_t den, li;
    fr_sub(&den, &tau, &wi);
    fr_mul(&den, &den, &nf);
    fr_inv(&den, &den);
    fr_mul(&li, &wi, &tn);
    fr_mul(&li, &li, &den);
    u256 c;
    fr_to_canonical(&c, &li);
    g1_jac t;
    gen_mul(&t, &c);
    g1_to_affine(&gl[i], &t);
    fr_mul(&wi, &wi, &omega);
  }
}

/* ------------------------------------------------------------------ MSM */
void orc_msm_naive(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n) {
  orc_init();
  g1_jac acc;
  g1_set_identity(&acc);
  for (size_t i = 0; i < n; i++) {
    u256 k;
    fr_to_canonical(&k, &scalars[i]);
    g1_jac t;
    g1_scalar_mul(&t, &bases[i], &k);
    g1_add(&acc, &acc, &t);
  }
  g1_to_affine(out, &acc);
}

/* serial Pippenger as halo2 `multiexp_serial`: unsigned c-bit windows over the canonical
 * little-endian scalar, 256/c + 1 segments from the top, 2^c - 1 buckets, running-sum reduce */
static void msm_serial(g1_jac *acc, const u256 *k, const g1_affine *bases, size_t n) {
  unsigned c;
  if (n < 4) c = 1;
  else if (n < 32) c = 3;
  else c = (unsigned)ceil(log((double)n));
  unsigned segments = 256 / c + 1;
  size_t nb = ((size_t)1 << c) - 1;
  g1_jac *buckets = (g1_jac *)malloc(sizeof(g1_jac) * nb);
  for (int seg = (int)segments - 1; seg >= 0; seg--) {
    for (unsigned d = 0; d < c; d++) g1_double(acc, acc);
    memset(buckets, 0, sizeof(g1_jac) * nb);
    unsigned skip = (unsigned)seg * c;
    for (size_t i = 0; i < n; i++) {
      if (skip >= 256) continue;
      u256 t;
      u256_shr(&t, &k[i], skip);
      size_t d = (size_t)(t.l[0] & (((uint64_t)1 << c) - 1));
      if (d) g1_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
    }
    g1_jac run;
    g1_set_identity(&run);
    for (size_t b = nb; b-- > 0;) {
      g1_add(&run, &run, &buckets[b]);
      g1_add(acc, acc, &run);
    }
  }
  free(buckets);
}
typedef struct {
  const u256 *k;
  const g1_affine *bases;
  size_t n;
  g1_jac acc;
} msm_job;
static void *msm_thread(void *arg) {
  msm_job *j = (msm_job *)arg;
  g1_set_identity(&j->acc);
  msm_serial(&j->acc, j->k, j->bases, j->n);
  return NULL;
}
void orc_msm_pippenger(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n, int threads) {
  orc_init();
  u256 *k = (u256 *)malloc(sizeof(u256) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) fr_to_canonical(&k[i], &scalars[i]);
  g1_jac total;
  g1_set_identity(&total);
  if (threads < 1) threads = 1;
  if (n > (size_t)threads && threads > 1) {
    size_t chunk = n / (size_t)threads;
    size_t nj = (n + chunk - 1) / chunk;
    msm_job *jobs = (msm_job *)calloc(nj, sizeof(msm_job));
    pthread_t *th = (pthread_t *)calloc(nj, sizeof(pthread_t));
    for (size_t t = 0; t < nj; t++) {
      size_t lo = t * chunk, hi = lo + chunk > n ? n : lo + chunk;
      jobs[t].k = k + lo;
      jobs[t].bases = bases + lo;
      jobs[t].n = hi - lo;
      pthread_create(&th[t], NULL, msm_thread, &jobs[t]);
    }
    for (size_t t = 0; t < nj; t++) {
      pthread_join(th[t], NULL);
      g1_add(&total, &total, &jobs[t].acc);
    }
    free(jobs);
    free(th);
  } else {
    msm_serial(&total, k, bases, n);
  }
  free(k);
  g1_to_affine(out, &total);
}
typedef struct {
  g1_affine *out;
  const fr_t *scalars;
  const g1_affine *bases;
  size_t n, c0, c1;
} msmb_job;
static void *msmb_thread(void *arg) {
  msmb_job *j = (msmb_job *)arg;
  for (size_t c = j->c0; c < j->c1; c++) orc_msm_pippenger(&j->out[c], j->scalars + c * j->n, j->bases, j->n, 1);
  return NULL;
}
void orc_msm_batch(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n, size_t n_cols, int threads) {
  orc_init();
  if (threads < 1) threads = 1;
  if ((size_t)threads > n_cols) threads = (int)(n_cols ? n_cols : 1);
  msmb_job *jobs = (msmb_job *)calloc((size_t)threads, sizeof(msmb_job));
  pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t] = (msmb_job){out, scalars, bases, n, n_cols * (size_t)t / (size_t)threads, n_cols * (size_t)(t + 1) / (size_t)threads};
    pthread_create(&th[t], NULL, msmb_thread, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
}

/* ------------------------------------------------------------------ NTT */
static size_t bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) {
    r = (r << 1) | (x & 1);
    x >>= 1;
  }
  return r;
}
/* halo2 `best_fft`: bit-reversal permutation then radix-2 DIT stages with a twiddle table */
void orc_ntt(fr_t *a, unsigned log_n, const fr_t *omega) {
  orc_init();
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitrev(k, log_n);
    if (k < rk) {
      fr_t t = a[k];
      a[k] = a[rk];
      a[rk] = t;
    }
  }
  size_t half = n / 2 ? n / 2 : 1;
  fr_t *tw = (fr_t *)malloc(sizeof(fr_t) * half);
  fr_t w = ORC_FR.r1;
  for (size_t i = 0; i < n / 2; i++) {
    tw[i] = w;
    fr_mul(&w, &w, omega);
  }
  for (unsigned s = 0; s < log_n; s++) {
    size_t m = (size_t)1 << s; /* half block */
    size_t stride = n / (2 * m);
    for (size_t blk = 0; blk < n; blk += 2 * m) {
      for (size_t j = 0; j < m; j++) {
        fr_t t, u = a[blk + j];
        fr_mul(&t, &a[blk + j + m], &tw[j * stride]);
        fr_add(&a[blk + j], &u, &t);
        fr_sub(&a[blk + j + m], &u, &t);
      }
    }
  }
  free(tw);
}
void orc_ntt_naive_dft(fr_t *out, const fr_t *a, unsigned log_n, const fr_t *omega) {
  orc_init();
  size_t n = (size_t)1 << log_n;
  fr_t wi = ORC_FR.r1;
  for (size_t i = 0; i < n; i++) {
    fr_t acc = {{0, 0, 0, 0}}, w = ORC_FR.r1;
    for (size_t j = 0; j < n; j++) {
      fr_t t;
      fr_mul(&t, &a[j], &w);
      fr_add(&acc, &acc, &t);
      fr_mul(&w, &w, &wi);
    }
    out[i] = acc;
    fr_mul(&wi, &wi, omega);
  }
}
void orc_lagrange_to_coeff(fr_t *a, unsigned k) {
  fr_t omega, omega_inv, ninv;
  orc_fr_root_of_unity(&omega, k);
  fr_inv(&omega_inv, &omega);
  fr_from_u64(&ninv, (uint64_t)1 << k);
  fr_inv(&ninv, &ninv);
  orc_ntt(a, k, &omega_inv);
  for (size_t i = 0; i < ((size_t)1 << k); i++) fr_mul(&a[i], &a[i], &ninv);
}
/* halo2 EvaluationDomain::coeff_to_extended: distribute_powers_zeta(into_coset) with the cyclic
 * pattern [1, ZETA, ZETA^2], zero-extend, forward NTT with the extended omega */
void orc_coeff_to_extended(fr_t *out, const fr_t *coeffs, unsigned k, unsigned ext) {
  orc_init();
  size_t n = (size_t)1 << k, ne = (size_t)1 << (k + ext);
  fr_t z2, omega_e;
  fr_mul(&z2, &FR_ZETA, &FR_ZETA);
  for (size_t i = 0; i < n; i++) {
    out[i] = coeffs[i];
    if (i % 3 == 1) fr_mul(&out[i], &out[i], &FR_ZETA);
    if (i % 3 == 2) fr_mul(&out[i], &out[i], &z2);
  }
  memset(out + n, 0, sizeof(fr_t) * (ne - n));
  orc_fr_root_of_unity(&omega_e, k + ext);
  orc_ntt(out, k + ext, &omega_e);
}
/* halo2 EvaluationDomain::extended_to_coeff, without the final truncation: inverse NTT over the extended domain, then
 * distribute_powers_zeta(out of the coset): coefficient i times [1, ZETA^2, ZETA][i mod 3]  [UPSTREAM-RECALL] */
void orc_extended_to_coeff(fr_t *a, unsigned k, unsigned ext) {
  orc_init();
  fr_t z2;
  fr_mul(&z2, &FR_ZETA, &FR_ZETA);
  orc_lagrange_to_coeff(a, k + ext); /* same arithmetic: inverse transform of size 2^(k+ext) with 1/N */
  for (size_t i = 0; i < ((size_t)1 << (k + ext)); i++) {
    if (i % 3 == 1) fr_mul(&a[i], &a[i], &z2);
    if (i % 3 == 2) fr_mul(&a[i], &a[i], &FR_ZETA);
  }
}
typedef struct {
  fr_t *cols, *ext;
  size_t c0, c1;
  unsigned k, extk;
  const fr_t *omega;
  int mode;
} ntt_job;
static void *ntt_thread(void *arg) {
  ntt_job *j = (ntt_job *)arg;
  size_t n = (size_t)1 << j->k;
  for (size_t c = j->c0; c < j->c1; c++) {
    if (j->mode == 0) {
      orc_ntt(j->cols + c * n, j->k, j->omega);
    } else {
      orc_lagrange_to_coeff(j->cols + c * n, j->k);
      orc_coeff_to_extended(j->ext + c * (n << j->extk), j->cols + c * n, j->k, j->extk);
    }
  }
  return NULL;
}
static void ntt_par(ntt_job proto, size_t n_cols, int threads) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n_cols) threads = (int)(n_cols ? n_cols : 1);
  ntt_job *jobs = (ntt_job *)calloc((size_t)threads, sizeof(ntt_job));
  pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t] = proto;
    jobs[t].c0 = n_cols * (size_t)t / (size_t)threads;
    jobs[t].c1 = n_cols * (size_t)(t + 1) / (size_t)threads;
    pthread_create(&th[t], NULL, ntt_thread, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
}
void orc_ntt_batch(fr_t *cols, size_t n_cols, unsigned log_n, const fr_t *omega, int threads) {
  orc_init();
  ntt_job p = {cols, NULL, 0, 0, log_n, 0, omega, 0};
  ntt_par(p, n_cols, threads);
}
void orc_lde_batch(fr_t *ext_out, fr_t *cols, size_t n_cols, unsigned k, unsigned ext, int threads) {
  orc_init();
  ntt_job p = {cols, ext_out, 0, 0, k, ext, NULL, 1};
  ntt_par(p, n_cols, threads);
}

/* ------------------------------------------------------------------ init */
static pthread_mutex_t g_init_mu = PTHREAD_MUTEX_INITIALIZER;
void orc_init(void) {
  if (__atomic_load_n(&g_init_done, __ATOMIC_ACQUIRE)) return;
  pthread_mutex_lock(&g_init_mu);
  if (!g_init_done) {
    /* SURVEY App. D [VERIFIED-HERE]: r, q */
    static const uint64_t R_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static const uint64_t Q_MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    mont_ctx_setup(&ORC_FR, R_MOD);
    mont_ctx_setup(&ORC_FQ, Q_MOD);
    /* ROOT_OF_UNITY = 7^((r-1) >> 28), order 2^28 */
    fr_t seven;
    u256 e, one = {{1, 0, 0, 0}}, t7 = {{7, 0, 0, 0}};
    mont_from_canonical(&seven, &t7, &ORC_FR);
    u256_sub(&e, &ORC_FR.p, &one);
    u256_shr(&e, &e, 28);
    mont_pow(&FR_ROOT_OF_UNITY, &seven, &e, &ORC_FR);
    /* halo2curves bn256 Fr::ZETA = from_raw([0x8b17ea66b99c90dd, 0x5bfc41088d8daaa7, 0xb3c4d79d41a91758, 0]) [UPSTREAM-RECALL] */
    u256 z = {{0x8b17ea66b99c90ddULL, 0x5bfc41088d8daaa7ULL, 0xb3c4d79d41a91758ULL, 0}};
    mont_from_canonical(&FR_ZETA, &z, &ORC_FR);
    u256 gx = {{1, 0, 0, 0}}, gy = {{2, 0, 0, 0}};
    mont_from_canonical(&G1_GEN.x, &gx, &ORC_FQ);
    mont_from_canonical(&G1_GEN.y, &gy, &ORC_FQ);
    __atomic_store_n(&g_init_done, 1, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_init_mu);
}
