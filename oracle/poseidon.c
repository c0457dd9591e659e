/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see poseidon.h).
 */
#include "poseidon.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Grain LFSR
 * Poseidon paper reference script generate_parameters_grain.sage: 80-bit state initialised with
 * field(2b)=1, sbox(4b)=0, n(12b), t(12b), R_F(10b), R_P(10b), thirty 1 bits; feedback taps
 * 62,51,38,23,13,0; first 160 outputs discarded; self-shrinking output (pairs). */
typedef struct {
  uint8_t s[80];
} grain_t;
static int grain_update(grain_t *g) {
  int nb = g->s[62] ^ g->s[51] ^ g->s[38] ^ g->s[23] ^ g->s[13] ^ g->s[0];
  memmove(g->s, g->s + 1, 79);
  g->s[79] = (uint8_t)nb;
  return nb;
}
static void grain_put(grain_t *g, int *pos, unsigned v, int bits) {
  for (int i = bits - 1; i >= 0; i--) g->s[(*pos)++] = (v >> i) & 1;
}
static void grain_init(grain_t *g, int n_bits, int t, int r_f, int r_p) {
  int pos = 0;
  grain_put(g, &pos, 1, 2);
  grain_put(g, &pos, 0, 4);
  grain_put(g, &pos, (unsigned)n_bits, 12);
  grain_put(g, &pos, (unsigned)t, 12);
  grain_put(g, &pos, (unsigned)r_f, 10);
  grain_put(g, &pos, (unsigned)r_p, 10);
  for (int i = 0; i < 30; i++) g->s[pos++] = 1;
  for (int i = 0; i < 160; i++) grain_update(g);
}
static int grain_next_bit(grain_t *g) {
  int b = grain_update(g);
  while (b == 0) {
    grain_update(g); /* discarded */
    b = grain_update(g);
  }
  return grain_update(g);
}
/* n_bits (=254) bits, most significant first, as a canonical integer */
static void grain_next_int(grain_t *g, u256 *o, int n_bits) {
  u256 v = {{0, 0, 0, 0}};
  for (int i = 0; i < n_bits; i++) {
    u256_shl(&v, &v, 1);
    v.l[0] |= (uint64_t)grain_next_bit(g);
  }
  *o = v;
}
static void grain_next_fe(grain_t *g, fr_t *o, int reject) {
  u256 v;
  for (;;) {
    grain_next_int(g, &v, 254);
    if (!reject || u256_cmp(&v, &ORC_FR.p) < 0) break;
  }
  fr_from_canonical(o, &v); /* reduces mod r when !reject */
}

/* ------------------------------------------------------------------ small dense matrices over Fr */
typedef struct {
  int n;
  fr_t m[PSD_MAX_T][PSD_MAX_T];
} mat_t;
static void mat_identity(mat_t *o, int n) {
  memset(o, 0, sizeof(*o));
  o->n = n;
  for (int i = 0; i < n; i++) o->m[i][i] = ORC_FR.r1;
}
static void mat_transpose(mat_t *o, const mat_t *a) {
  mat_t t;
  t.n = a->n;
  for (int i = 0; i < a->n; i++)
    for (int j = 0; j < a->n; j++) t.m[i][j] = a->m[j][i];
  *o = t;
}
static void mat_mul(mat_t *o, const mat_t *a, const mat_t *b) {
  mat_t t;
  memset(&t, 0, sizeof(t));
  t.n = a->n;
  for (int i = 0; i < a->n; i++)
    for (int j = 0; j < a->n; j++)
      for (int k = 0; k < a->n; k++) {
        fr_t p;
        fr_mul(&p, &a->m[i][k], &b->m[k][j]);
        fr_add(&t.m[i][j], &t.m[i][j], &p);
      }
  *o = t;
}
static void mat_mul_vec(fr_t *o, const mat_t *a, const fr_t *v) {
  fr_t r[PSD_MAX_T];
  for (int i = 0; i < a->n; i++) {
    memset(&r[i], 0, sizeof(fr_t));
    for (int j = 0; j < a->n; j++) {
      fr_t p;
      fr_mul(&p, &a->m[i][j], &v[j]);
      fr_add(&r[i], &r[i], &p);
    }
  }
  memcpy(o, r, sizeof(fr_t) * (size_t)a->n);
}
/* Gauss-Jordan */
static void mat_invert(mat_t *o, const mat_t *a) {
  int n = a->n;
  mat_t w = *a, inv;
  mat_identity(&inv, n);
  for (int c = 0; c < n; c++) {
    int piv = c;
    while (piv < n && fr_is_zero(&w.m[piv][c])) piv++;
    if (piv == n) abort();
    if (piv != c)
      for (int j = 0; j < n; j++) {
        fr_t t = w.m[c][j];
        w.m[c][j] = w.m[piv][j];
        w.m[piv][j] = t;
        t = inv.m[c][j];
        inv.m[c][j] = inv.m[piv][j];
        inv.m[piv][j] = t;
      }
    fr_t pi;
    fr_inv(&pi, &w.m[c][c]);
    for (int j = 0; j < n; j++) {
      fr_mul(&w.m[c][j], &w.m[c][j], &pi);
      fr_mul(&inv.m[c][j], &inv.m[c][j], &pi);
    }
    for (int r = 0; r < n; r++) {
      if (r == c || fr_is_zero(&w.m[r][c])) continue;
      fr_t f = w.m[r][c];
      for (int j = 0; j < n; j++) {
        fr_t t;
        fr_mul(&t, &f, &w.m[c][j]);
        fr_sub(&w.m[r][j], &w.m[r][j], &t);
        fr_mul(&t, &f, &inv.m[c][j]);
        fr_sub(&inv.m[r][j], &inv.m[r][j], &t);
      }
    }
  }
  *o = inv;
}

/* PSE poseidon MDSMatrix::factorise: M = M' * M'' with M' = [[1,0],[0,M_hat]] and
 * M'' = [[m00, row],[w_hat, I]]; returns (M', transpose(M'')) */
static void mat_factorise(const mat_t *m, mat_t *m_prime, mat_t *m_pp_T) {
  int n = m->n;
  mat_t mhat, mhat_inv, pp;
  mat_identity(&mhat, n - 1);
  fr_t w[PSD_MAX_T], w_hat[PSD_MAX_T];
  for (int i = 1; i < n; i++) {
    w[i - 1] = m->m[i][0];
    for (int j = 1; j < n; j++) mhat.m[i - 1][j - 1] = m->m[i][j];
  }
  mat_invert(&mhat_inv, &mhat);
  mat_mul_vec(w_hat, &mhat_inv, w);
  mat_identity(m_prime, n);
  for (int i = 1; i < n; i++)
    for (int j = 1; j < n; j++) m_prime->m[i][j] = mhat.m[i - 1][j - 1];
  mat_identity(&pp, n);
  for (int j = 0; j < n; j++) pp.m[0][j] = m->m[0][j];
  for (int i = 1; i < n; i++) pp.m[i][0] = w_hat[i - 1];
  mat_transpose(m_pp_T, &pp);
}

static void spec_build(psd_spec *s, int t, int r_f, int r_p) {
  orc_init();
  memset(s, 0, sizeof(*s));
  s->t = t;
  s->rate = t - 1;
  s->r_f = r_f;
  s->r_p = r_p;
  grain_t g;
  grain_init(&g, 254, t, r_f, r_p);
  int rounds = r_f + r_p;
  for (int r = 0; r < rounds; r++)
    for (int i = 0; i < t; i++) grain_next_fe(&g, &s->rc[r][i], 1);
  fr_t xs[PSD_MAX_T], ys[PSD_MAX_T];
  for (int i = 0; i < t; i++) grain_next_fe(&g, &xs[i], 0);
  for (int i = 0; i < t; i++) grain_next_fe(&g, &ys[i], 0);
  mat_t mds;
  mds.n = t;
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) {
      fr_t sum;
      fr_add(&sum, &xs[i], &ys[j]);
      fr_inv(&mds.m[i][j], &sum);
      s->mds[i][j] = mds.m[i][j];
    }
  /* ---- optimized constants (PSE poseidon Spec::calculate_optimized_constants) ---- */
  mat_t inv;
  mat_invert(&inv, &mds);
  int half = r_f / 2;
  memcpy(s->start[0], s->rc[0], sizeof(fr_t) * (size_t)t);
  for (int r = 1; r < half; r++) mat_mul_vec(s->start[r], &inv, s->rc[r]);
  fr_t acc[PSD_MAX_T];
  memcpy(acc, s->rc[half + r_p], sizeof(fr_t) * (size_t)t);
  for (int p = r_p - 1; p >= 0; p--) {
    /* pairs partial[p] with unoptimized constants of round half + p */
    fr_t tmp[PSD_MAX_T];
    mat_mul_vec(tmp, &inv, acc);
    s->partial[p] = tmp[0];
    memset(&tmp[0], 0, sizeof(fr_t));
    for (int i = 0; i < t; i++) fr_add(&acc[i], &tmp[i], &s->rc[half + p][i]);
  }
  mat_mul_vec(s->start[half], &inv, acc);
  for (int r = 0; r < half - 1; r++) mat_mul_vec(s->end[r], &inv, s->rc[half + r_p + 1 + r]);
  /* ---- sparse matrices (Spec::calculate_sparse_matrices) ---- */
  mat_t mds_T, accm;
  mat_transpose(&mds_T, &mds);
  accm = mds_T;
  for (int p = 0; p < r_p; p++) {
    mat_t mp, mppT;
    mat_factorise(&accm, &mp, &mppT);
    mat_mul(&accm, &mds_T, &mp);
    /* list is reversed afterwards: factorisation p is used at partial round r_p-1-p */
    int dst = r_p - 1 - p;
    for (int j = 0; j < t; j++) s->sparse_row[dst][j] = mppT.m[0][j];
    for (int i = 1; i < t; i++) s->sparse_col_hat[dst][i - 1] = mppT.m[i][0];
  }
  mat_t pre;
  mat_transpose(&pre, &accm);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) s->pre_sparse_mds[i][j] = pre.m[i][j];
}

static psd_spec *g_specs[8];
static int g_nspecs = 0;
static pthread_mutex_t g_spec_mu = PTHREAD_MUTEX_INITIALIZER;
const psd_spec *psd_get_spec(int t, int r_f, int r_p) {
  pthread_mutex_lock(&g_spec_mu);
  for (int i = 0; i < g_nspecs; i++)
    if (g_specs[i]->t == t && g_specs[i]->r_f == r_f && g_specs[i]->r_p == r_p) {
      pthread_mutex_unlock(&g_spec_mu);
      return g_specs[i];
    }
  if (g_nspecs == 8 || t > PSD_MAX_T || r_f + r_p > PSD_MAX_ROUNDS) abort();
  psd_spec *s = (psd_spec *)malloc(sizeof(psd_spec));
  spec_build(s, t, r_f, r_p);
  g_specs[g_nspecs++] = s;
  pthread_mutex_unlock(&g_spec_mu);
  return s;
}

static void pow5(fr_t *o, const fr_t *x) {
  fr_t x2, x4;
  fr_mul(&x2, x, x);
  fr_mul(&x4, &x2, &x2);
  fr_mul(o, &x4, x);
}
static void apply_dense(const psd_spec *s, fr_t *st, const fr_t m[PSD_MAX_T][PSD_MAX_T]) {
  fr_t r[PSD_MAX_T];
  for (int i = 0; i < s->t; i++) {
    memset(&r[i], 0, sizeof(fr_t));
    for (int j = 0; j < s->t; j++) {
      fr_t p;
      fr_mul(&p, &m[i][j], &st[j]);
      fr_add(&r[i], &r[i], &p);
    }
  }
  memcpy(st, r, sizeof(fr_t) * (size_t)s->t);
}
void psd_permute_naive(const psd_spec *s, fr_t *st) {
  int half = s->r_f / 2, rounds = s->r_f + s->r_p;
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < s->t; i++) fr_add(&st[i], &st[i], &s->rc[r][i]);
    if (r < half || r >= half + s->r_p) {
      for (int i = 0; i < s->t; i++) pow5(&st[i], &st[i]);
    } else {
      pow5(&st[0], &st[0]);
    }
    apply_dense(s, st, s->mds);
  }
}
/* PoseidonChip::permutation (snark-verifier / halo2-lib community-edition poseidon chip) */
void psd_permute_absorb(const psd_spec *s, fr_t *st, const fr_t *inputs, int n_in) {
  int t = s->t, half = s->r_f / 2;
  /* absorb_with_pre_constants */
  fr_add(&st[0], &st[0], &s->start[0][0]);
  for (int i = 0; i < n_in; i++) {
    fr_add(&st[1 + i], &st[1 + i], &inputs[i]);
    fr_add(&st[1 + i], &st[1 + i], &s->start[0][1 + i]);
  }
  for (int i = n_in + 1, k = 0; i < t; i++, k++) {
    fr_t c = s->start[0][i];
    if (k == 0) fr_add(&c, &c, &ORC_FR.r1);
    fr_add(&st[i], &st[i], &c);
  }
  for (int r = 1; r < half; r++) {
    for (int i = 0; i < t; i++) {
      pow5(&st[i], &st[i]);
      fr_add(&st[i], &st[i], &s->start[r][i]);
    }
    apply_dense(s, st, s->mds);
  }
  for (int i = 0; i < t; i++) {
    pow5(&st[i], &st[i]);
    fr_add(&st[i], &st[i], &s->start[half][i]);
  }
  apply_dense(s, st, s->pre_sparse_mds);
  for (int p = 0; p < s->r_p; p++) {
    pow5(&st[0], &st[0]);
    fr_add(&st[0], &st[0], &s->partial[p]);
    fr_t n0, r[PSD_MAX_T];
    memset(&n0, 0, sizeof(n0));
    for (int j = 0; j < t; j++) {
      fr_t q;
      fr_mul(&q, &s->sparse_row[p][j], &st[j]);
      fr_add(&n0, &n0, &q);
    }
    for (int i = 1; i < t; i++) {
      fr_mul(&r[i], &st[0], &s->sparse_col_hat[p][i - 1]);
      fr_add(&r[i], &r[i], &st[i]);
    }
    st[0] = n0;
    for (int i = 1; i < t; i++) st[i] = r[i];
  }
  for (int r = 0; r < half - 1; r++) {
    for (int i = 0; i < t; i++) {
      pow5(&st[i], &st[i]);
      fr_add(&st[i], &st[i], &s->end[r][i]);
    }
    apply_dense(s, st, s->mds);
  }
  for (int i = 0; i < t; i++) pow5(&st[i], &st[i]);
  apply_dense(s, st, s->mds);
}
void psd_hash(const psd_spec *s, fr_t *out, const fr_t *msg, size_t len) {
  fr_t st[PSD_MAX_T];
  memset(st, 0, sizeof(st));
  u256 cap = {{0, 1, 0, 0}}; /* 2^64 */
  fr_from_canonical(&st[0], &cap);
  size_t rate = (size_t)s->rate, off = 0;
  int padding_offset = 0;
  /* squeeze(): chunks(RATE); empty input => zero chunks => padding_offset stays 0 */
  while (off < len) {
    size_t c = len - off < rate ? len - off : rate;
    padding_offset = (int)(rate - c);
    psd_permute_absorb(s, st, msg + off, (int)c);
    off += c;
  }
  if (padding_offset == 0) psd_permute_absorb(s, st, NULL, 0);
  *out = st[1];
}

void orc_poseidon_permute_naive(int t, int r_f, int r_p, fr_t *state) { psd_permute_naive(psd_get_spec(t, r_f, r_p), state); }
/* optimized permutation of a raw state: equals naive when the absorb step adds only constants.
 * We emulate that by absorbing zero inputs at every rate position and undoing the +1 padding. */
void orc_poseidon_permute_opt(int t, int r_f, int r_p, fr_t *state) {
  const psd_spec *s = psd_get_spec(t, r_f, r_p);
  fr_t zeros[PSD_MAX_T];
  memset(zeros, 0, sizeof(zeros));
  psd_permute_absorb(s, state, zeros, s->rate);
}
void orc_poseidon_hash_many(int t, int r_f, int r_p, const fr_t *in, size_t n_msgs, size_t msg_len, fr_t *dig) {
  const psd_spec *s = psd_get_spec(t, r_f, r_p);
  for (size_t i = 0; i < n_msgs; i++) psd_hash(s, &dig[i], in + i * msg_len, msg_len);
}
/* value-level merkle_commitment (/root/reference/src/gadget/vectordb.rs:165-223) */
void orc_poseidon_merkle_root(int t, int r_f, int r_p, const fr_t *vectors, size_t n, size_t dim, fr_t *root) {
  const psd_spec *s = psd_get_spec(t, r_f, r_p);
  size_t leaves = 1;
  while (leaves < n) leaves <<= 1;
  fr_t *lv = (fr_t *)calloc(leaves, sizeof(fr_t));
  for (size_t i = 0; i < n; i++) psd_hash(s, &lv[i], vectors + i * dim, dim);
  while (leaves > 1) {
    for (size_t i = 0; i < leaves; i += 2) {
      fr_t pair[2] = {lv[i], lv[i + 1]}, h;
      psd_hash(s, &h, pair, 2);
      lv[i / 2] = h;
    }
    leaves >>= 1;
  }
  *root = lv[0];
  free(lv);
}
void orc_poseidon_spec_dump(int t, int r_f, int r_p, fr_t *rc, fr_t *mds) {
  const psd_spec *s = psd_get_spec(t, r_f, r_p);
  for (int r = 0; r < r_f + r_p; r++)
    for (int i = 0; i < t; i++) rc[r * t + i] = s->rc[r][i];
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) mds[i * t + j] = s->mds[i][j];
}
