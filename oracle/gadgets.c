/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gadgets.h).
 * Each function cites the reference lines it restates.  halo2-base / poseidon templates are
 * [UPSTREAM-RECALL] of halo2-lib v0.3 "community-edition" (absent from /root/reference).
 */
#include "gadgets.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ================================================================== Context */
typedef struct {
  size_t idx;
  fr_t den;
} inv_rec;
typedef struct {
  octx c;
  inv_rec *inv; /* deferred Assigned::Rational(1, x) cells — batch inverted like halo2 does */
  size_t ninv, capinv;
} octx_full;

octx *orc_ctx_new(int store, int keygen) {
  orc_init();
  octx_full *f = (octx_full *)calloc(1, sizeof(octx_full));
  f->c.store = store;
  f->c.keygen = keygen;
  return &f->c;
}
void orc_ctx_enable_plan(octx *c, unsigned k, unsigned minimum_rows) {
  c->plan = 1;
  c->max_rows = ((size_t)1 << k) - minimum_rows;
  c->row = 0;
}
void orc_ctx_free(octx *c) {
  octx_full *f = (octx_full *)c;
  free(c->advice);
  free(c->sel);
  free(c->lookup);
  free(c->bp);
  free(f->inv);
  free(f);
}
static void ctx_finalize(octx *c) {
  octx_full *f = (octx_full *)c;
  if (!f->ninv) return;
  /* Montgomery batch inversion */
  size_t n = f->ninv;
  fr_t *pre = (fr_t *)malloc(sizeof(fr_t) * n);
  fr_t acc = ORC_FR.r1;
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    fr_mul(&acc, &acc, &f->inv[i].den);
  }
  fr_inv(&acc, &acc);
  for (size_t i = n; i-- > 0;) {
    fr_t v;
    fr_mul(&v, &acc, &pre[i]);
    c->advice[f->inv[i].idx] = v;
    fr_mul(&acc, &acc, &f->inv[i].den);
  }
  free(pre);
  f->ninv = 0;
}
size_t orc_ctx_len(const octx *c) { return c->n; }
size_t orc_ctx_lookup_len(const octx *c) { return c->nl; }
const fr_t *orc_ctx_advice(const octx *c) {
  ctx_finalize((octx *)c);
  return c->advice;
}
const fr_t *orc_ctx_lookup(const octx *c) { return c->lookup; }
const uint8_t *orc_ctx_selectors(const octx *c) { return c->sel; }
int orc_ctx_err(const octx *c) { return c->err; }
size_t orc_ctx_break_points(const octx *c, size_t *out, size_t cap) {
  for (size_t i = 0; i < c->nbp && i < cap; i++) out[i] = c->bp[i];
  return c->nbp;
}

/* Context::assign_cell + the row walk of GateThreadBuilder::assign_all (break points) */
static inline void push(octx *c, const fr_t *v, int q) {
  if (c->store) {
    if (c->n == c->cap) {
      c->cap = c->cap ? c->cap * 2 : 4096;
      c->advice = (fr_t *)realloc(c->advice, sizeof(fr_t) * c->cap);
      if (c->keygen) c->sel = (uint8_t *)realloc(c->sel, c->cap);
    }
    c->advice[c->n] = *v;
    if (c->keygen) c->sel[c->n] = (uint8_t)q;
  }
  if (c->plan) {
    if ((q && c->row + 4 > c->max_rows) || c->row >= c->max_rows - 1) {
      if (c->nbp == c->capbp) {
        c->capbp = c->capbp ? c->capbp * 2 : 256;
        c->bp = (size_t *)realloc(c->bp, sizeof(size_t) * c->capbp);
      }
      c->bp[c->nbp++] = c->row;
      c->row = 0;
    }
    c->row++;
  }
  c->n++;
}
static inline void push_inverse_of(octx *c, const fr_t *x) {
  /* WitnessFraction(Rational(1, x)); Trivial(1) when x == 0 (GateChip::is_zero) */
  if (fr_is_zero(x) || !c->store) {
    push(c, &ORC_FR.r1, 0);
    return;
  }
  octx_full *f = (octx_full *)c;
  if (f->ninv == f->capinv) {
    f->capinv = f->capinv ? f->capinv * 2 : 1024;
    f->inv = (inv_rec *)realloc(f->inv, sizeof(inv_rec) * f->capinv);
  }
  f->inv[f->ninv].idx = c->n;
  f->inv[f->ninv].den = *x;
  f->ninv++;
  push(c, &ORC_FR.r1, 0); /* placeholder, patched by ctx_finalize */
}
static inline void lookup_push(octx *c, const fr_t *v) {
  if (c->store) {
    if (c->nl == c->capl) {
      c->capl = c->capl ? c->capl * 2 : 1024;
      c->lookup = (fr_t *)realloc(c->lookup, sizeof(fr_t) * c->capl);
    }
    c->lookup[c->nl] = *v;
  }
  c->nl++;
}
/* assign_region(inputs, gate_offsets) for short fixed templates; gates bitmask over positions */
static inline void region(octx *c, const fr_t *v, int n, unsigned gates) {
  for (int i = 0; i < n; i++) push(c, &v[i], (gates >> i) & 1);
}

/* ================================================================== chip state */
typedef struct {
  fpchip *f;
  octx *c;
} G;
static fr_t g_small[257];
static int g_small_init = 0;
static const fr_t *small(unsigned v) {
  if (!g_small_init) {
    for (unsigned i = 0; i <= 256; i++) fr_from_u64(&g_small[i], i);
    g_small_init = 1;
  }
  return &g_small[v];
}
#define ONE (&g->f->one)
#define ZERO (&g->f->zero)

/* ================================================================== GateChip (halo2-base flex_gate.rs) */
static fr_t g_add(G *g, const fr_t *a, const fr_t *b) { /* [a, b, 1, out] */
  fr_t v[4] = {*a, *b, *ONE};
  fr_add(&v[3], a, b);
  region(g->c, v, 4, 1);
  return v[3];
}
static fr_t g_sub(G *g, const fr_t *a, const fr_t *b) { /* [out, b, 1, a] */
  fr_t v[4];
  fr_sub(&v[0], a, b);
  v[1] = *b;
  v[2] = *ONE;
  v[3] = *a;
  region(g->c, v, 4, 1);
  return v[0];
}
static fr_t g_neg(G *g, const fr_t *a) { /* [a, out, 1, 0] */
  fr_t v[4] = {*a, *a, *ONE, *ZERO};
  fr_neg(&v[1], a);
  region(g->c, v, 4, 1);
  return v[1];
}
static fr_t g_mul(G *g, const fr_t *a, const fr_t *b) { /* [0, a, b, out] */
  fr_t v[4] = {*ZERO, *a, *b};
  fr_mul(&v[3], a, b);
  region(g->c, v, 4, 1);
  return v[3];
}
static fr_t g_mul_add(G *g, const fr_t *a, const fr_t *b, const fr_t *cc) { /* [c, a, b, out] */
  fr_t v[4] = {*cc, *a, *b};
  fr_mul(&v[3], a, b);
  fr_add(&v[3], &v[3], cc);
  region(g->c, v, 4, 1);
  return v[3];
}
static void g_assert_bit(G *g, const fr_t *x) { /* [0, x, x, x] */
  fr_t v[4] = {*ZERO, *x, *x, *x};
  region(g->c, v, 4, 1);
}
static fr_t g_not(G *g, const fr_t *a) { return g_sub(g, ONE, a); }
static fr_t g_and(G *g, const fr_t *a, const fr_t *b) { return g_mul(g, a, b); }
static fr_t g_or(G *g, const fr_t *a, const fr_t *b) { /* [1-b, 1, b, 1, b, a, 1-b, out] gates 0,4 */
  fr_t nb, out, ab;
  fr_sub(&nb, ONE, b);
  fr_mul(&ab, a, b);
  fr_add(&out, a, b);
  fr_sub(&out, &out, &ab);
  fr_t v[8] = {nb, *ONE, *b, *ONE, *b, *a, nb, out};
  region(g->c, v, 8, 0x11);
  return out;
}
static fr_t g_select(G *g, const fr_t *a, const fr_t *b, const fr_t *sel) { /* [a-b,1,b,a,b,sel,a-b,out] */
  fr_t d, out;
  fr_sub(&d, a, b);
  fr_mul(&out, &d, sel);
  fr_add(&out, &out, b);
  fr_t v[8] = {d, *ONE, *b, *a, *b, *sel, d, out};
  region(g->c, v, 8, 0x11);
  return out;
}
static fr_t g_is_zero(G *g, const fr_t *a) { /* [z, a, inv, 1, 0, a, z, 0] gates 0,4 */
  fr_t z = fr_is_zero(a) ? *ONE : *ZERO;
  push(g->c, &z, 1);
  push(g->c, a, 0);
  push_inverse_of(g->c, a);
  push(g->c, ONE, 0);
  push(g->c, ZERO, 1);
  push(g->c, a, 0);
  push(g->c, &z, 0);
  push(g->c, ZERO, 0);
  return z;
}
static fr_t g_is_equal(G *g, const fr_t *a, const fr_t *b) {
  fr_t d = g_sub(g, a, b);
  return g_is_zero(g, &d);
}
static fr_t load_constant(G *g, const fr_t *v) {
  push(g->c, v, 0);
  return *v;
}
static fr_t load_zero(G *g) { /* Context::load_zero caches the cell */
  if (!g->c->has_zero) {
    push(g->c, ZERO, 0);
    g->c->has_zero = 1;
  }
  return *ZERO;
}
/* GateInstructions::sum */
static fr_t g_sum(G *g, const fr_t *a, size_t n) {
  if (n == 0) return load_zero(g);
  push(g->c, &a[0], n > 1);
  fr_t s = a[0];
  for (size_t i = 1; i < n; i++) {
    fr_add(&s, &s, &a[i]);
    push(g->c, &a[i], 0);
    push(g->c, ONE, 0);
    push(g->c, &s, i + 1 < n); /* gate offsets 3i for i in 0..n-1 */
  }
  return s;
}
/* GateInstructions::inner_product; b_const: b are QuantumCell::Constant (enables starts-with-one) */
static fr_t g_inner_product(G *g, const fr_t *a, const fr_t *b, size_t n, int b_const) {
  fr_t s;
  size_t i0;
  size_t ngates;
  if (b_const && n > 0 && fr_eq(&b[0], ONE)) {
    s = a[0];
    i0 = 1;
    ngates = n - 1;
    push(g->c, &a[0], ngates > 0);
  } else {
    s = *ZERO;
    i0 = 0;
    ngates = n;
    push(g->c, ZERO, ngates > 0);
  }
  size_t gi = 1;
  for (size_t i = i0; i < n; i++, gi++) {
    fr_t p;
    fr_mul(&p, &a[i], &b[i]);
    fr_add(&s, &s, &p);
    push(g->c, &a[i], 0);
    push(g->c, &b[i], 0);
    push(g->c, &s, gi < ngates);
  }
  return s;
}
/* GateInstructions::select_by_indicator: running value takes a[i] when ind[i] != 0 */
static fr_t g_select_by_indicator(G *g, const fr_t *a, size_t stride, const fr_t *ind, size_t n) {
  fr_t s = *ZERO;
  push(g->c, ZERO, n > 0);
  for (size_t i = 0; i < n; i++) {
    const fr_t *ai = &a[i * stride];
    if (!fr_is_zero(&ind[i])) s = *ai;
    push(g->c, ai, 0);
    push(g->c, &ind[i], 0);
    push(g->c, &s, i + 1 < n);
  }
  return s;
}
/* GateInstructions::idx_to_indicator (v0.3): first element unrolled is_zero, then is_equal(idx, i) */
static void g_idx_to_indicator(G *g, const fr_t *idx, size_t len, fr_t *out) {
  for (size_t i = 0; i < len; i++) {
    if (i == 0) {
      out[0] = g_is_zero(g, idx);
    } else {
      fr_t ci;
      if (i <= 256) ci = *small((unsigned)i);
      else fr_from_u64(&ci, i);
      out[i] = g_is_equal(g, idx, &ci);
    }
  }
}
static fr_t g_select_from_idx(G *g, const fr_t *cells, size_t n, const fr_t *idx) {
  fr_t *ind = (fr_t *)malloc(sizeof(fr_t) * n);
  g_idx_to_indicator(g, idx, n, ind);
  fr_t r = g_select_by_indicator(g, cells, 1, ind, n);
  free(ind);
  return r;
}
/* GateInstructions::num_to_bits */
static void g_num_to_bits(G *g, const fr_t *a, unsigned range_bits, fr_t *bits) {
  u256 ac;
  fr_to_canonical(&ac, a);
  for (unsigned i = 0; i < range_bits; i++) bits[i] = ((ac.l[i / 64] >> (i % 64)) & 1) ? *ONE : *ZERO;
  g_inner_product(g, bits, g->f->pow2, range_bits, 1);
  for (unsigned i = 0; i < range_bits; i++) g_assert_bit(g, &bits[i]);
}

/* ================================================================== RangeChip (halo2-base range.rs) */
/* returns the value of the last cell pushed to cells_to_lookup */
static fr_t r_range_check(G *g, const fr_t *a, unsigned range_bits) {
  unsigned L = g->f->L;
  unsigned k = (range_bits + L - 1) / L;
  unsigned rem_bits = range_bits % L;
  fr_t last;
  if (k == 1) {
    lookup_push(g->c, a);
    last = *a;
  } else {
    u256 ac;
    fr_to_canonical(&ac, a);
    fr_t limbs[64], bases[64];
    for (unsigned i = 0; i < k; i++) {
      u256 t;
      u256_shr(&t, &ac, i * L);
      fr_from_u64(&limbs[i], t.l[0] & (((uint64_t)1 << L) - 1));
      bases[i] = g->f->pow2[i * L];
    }
    g_inner_product(g, limbs, bases, k, 1);
    for (unsigned i = 0; i < k; i++) lookup_push(g->c, &limbs[i]);
    last = limbs[k - 1];
  }
  if (rem_bits == 1) {
    g_assert_bit(g, &last);
  } else if (rem_bits > 1) {
    fr_t chk = g_mul(g, &last, &g->f->pow2[L - rem_bits]);
    lookup_push(g->c, &chk);
    last = chk;
  }
  return last;
}
static void r_check_less_than(G *g, const fr_t *a, const fr_t *b, unsigned num_bits) {
  /* [a + 2^n - b, b, 1, a + 2^n, -2^n, 1, a] gates 0,3 */
  fr_t sa, chk, np;
  fr_add(&sa, &g->f->pow2[num_bits], a);
  fr_sub(&chk, &sa, b);
  fr_neg(&np, &g->f->pow2[num_bits]);
  fr_t v[7] = {chk, *b, *ONE, sa, np, *ONE, *a};
  region(g->c, v, 7, 0x9);
  r_range_check(g, &chk, num_bits);
}
static void r_check_big_less_than_safe(G *g, const fr_t *a, const u256 *b) {
  unsigned L = g->f->L;
  unsigned range_bits = (u256_bits(b) + L - 1) / L * L;
  r_range_check(g, a, range_bits);
  fr_t bf;
  fr_from_canonical(&bf, b);
  r_check_less_than(g, a, &bf, range_bits);
}
static fr_t r_is_less_than(G *g, const fr_t *a, const fr_t *b, unsigned num_bits) {
  unsigned L = g->f->L;
  unsigned k = (num_bits + L - 1) / L;
  unsigned padded = k * L;
  fr_t sa, sh, np;
  fr_add(&sa, &g->f->pow2[padded], a);
  fr_sub(&sh, &sa, b);
  fr_neg(&np, &g->f->pow2[padded]);
  fr_t v[7] = {sh, *b, *ONE, sa, np, *ONE, *a};
  region(g->c, v, 7, 0x9);
  fr_t last = r_range_check(g, &sh, padded + L);
  return g_is_zero(g, &last);
}
static void r_div_mod(G *g, const fr_t *a, const u256 *b, unsigned a_num_bits, fr_t *div, fr_t *rem) {
  u256 ac, q, r;
  fr_to_canonical(&ac, a);
  u256_divmod(&q, &r, &ac, b);
  fr_t bf;
  fr_from_canonical(div, &q);
  fr_from_canonical(rem, &r);
  fr_from_canonical(&bf, b);
  fr_t v[4] = {*rem, bf, *div, *a};
  region(g->c, v, 4, 1);
  u256 bound, one = {{1, 0, 0, 0}};
  u256_shl(&bound, &one, a_num_bits);
  u256_divmod(&bound, NULL, &bound, b);
  u256_add(&bound, &bound, &one);
  r_check_big_less_than_safe(g, div, &bound);
  r_check_big_less_than_safe(g, rem, b);
}
static void r_div_mod_var(G *g, const fr_t *a, const fr_t *b, unsigned a_bits, unsigned b_bits, fr_t *div, fr_t *rem) {
  u256 ac, bc, q = {{0, 0, 0, 0}}, r = {{0, 0, 0, 0}};
  fr_to_canonical(&ac, a);
  fr_to_canonical(&bc, b);
  if (u256_is_zero(&bc)) g->c->err = 1; /* BigUint division by zero panics upstream */
  else u256_divmod(&q, &r, &ac, &bc);
  fr_from_canonical(div, &q);
  fr_from_canonical(rem, &r);
  fr_t v[4] = {*rem, *b, *div, *a};
  region(g->c, v, 4, 1);
  r_range_check(g, div, a_bits);
  r_check_less_than(g, rem, b, b_bits);
}

/* ================================================================== FixedPointChip */
static void quantize1(unsigned P, double x, fr_t *out) { /* fixed_point.rs:104-119 */
  int neg = !isnan(x) && signbit(x); /* f64::signum: -0.0 -> -1.0, NaN -> NaN */
  double ax = fabs(x);
  double y = round(ax * ldexp(1.0, (int)P)); /* round half away from zero, as f64::round */
  u128 q;
  if (isnan(y) || y <= 0.0) q = 0;
  else if (y >= 340282366920938463463374607431768211456.0) q = ~(u128)0; /* `as u128` saturates */
  else q = (u128)y;
  u256 c = {{(uint64_t)q, (uint64_t)(q >> 64), 0, 0}};
  fr_from_canonical(out, &c);
  if (neg) fr_neg(out, out);
}
void orc_fp_quantize(unsigned P, const double *x, fr_t *out, size_t n) {
  orc_init();
  for (size_t i = 0; i < n; i++) quantize1(P, x[i], &out[i]);
}
void orc_fp_dequantize(unsigned P, const fr_t *x, double *out, size_t n) { /* fixed_point.rs:121-136 */
  orc_init();
  u256 np, one = {{1, 0, 0, 0}}, t;
  u256_shl(&t, &one, 2 * P + 1);
  u256_sub(&np, &ORC_FR.p, &t);
  for (size_t i = 0; i < n; i++) {
    u256 c;
    fr_to_canonical(&c, &x[i]);
    double sign = 1.0;
    if (u256_cmp(&c, &np) > 0) {
      /* x_mut = bn254_max - x - 1 (field arithmetic) */
      fr_t bm, xm;
      u256 bmc;
      u256_sub(&bmc, &ORC_FR.p, &one);
      fr_from_canonical(&bm, &bmc);
      fr_sub(&xm, &bm, &x[i]);
      fr_t onef = ORC_FR.r1;
      fr_sub(&xm, &xm, &onef);
      fr_to_canonical(&c, &xm);
      sign = -1.0;
    }
    u128 lo = ((u128)c.l[1] << 64) | c.l[0];
    u128 sc = (u128)1 << P;
    double xi = (double)(lo / sc);
    double xf = (double)(lo % sc) / (double)sc;
    out[i] = sign * (xi + xf);
  }
}

static fpchip *g_chips[1024]; /* one per (P, L) a process ever asks for: at most 32 x 28 */
static int g_nchips = 0;
void orc_fp_init(fpchip *f, unsigned P, unsigned L) { /* fixed_point.rs:54-98, 138-187 */
  orc_init();
  memset(f, 0, sizeof(*f));
  f->P = P;
  f->L = L;
  f->one = ORC_FR.r1;
  f->pow2[0] = f->one;
  for (int i = 1; i < 254; i++) fr_add(&f->pow2[i], &f->pow2[i - 1], &f->pow2[i - 1]);
  f->scale = f->pow2[P];
  fr_neg(&f->negative_point, &f->pow2[2 * P + 1]);
  fr_to_canonical(&f->negative_point_c, &f->negative_point);
  static const double exp2c[13] = {3.6240421303547230336183979205877e-11, 4.1284327467833130245549169910389e-10,
                                   0.0000000071086385644026346316624185550542, 0.00000010172297085296590958930245291448,
                                   0.0000013215904023658396206789543841996, 0.000015252713316417140696221389106544,
                                   0.00015403531076657894204857389177279, 0.0013333558131297097698435464957392,
                                   0.0096181291078409107025643582456283, 0.055504108664804181586140094858174,
                                   0.24022650695910142332414229540187, 0.69314718055994529934452147700678, 1.0};
  static const double logc[15] = {-3.319586265362338e-08, 1.4957235315170112e-06, -3.1350053389526744e-05,
                                  0.00040554177582512901, -0.0036218342998850703, 0.023663846121538389,
                                  -0.11691877183255484, 0.44524062371564499, -1.3195777548208449,
                                  3.0518128028712077, -5.4904626000399528, 7.6298580090181591,
                                  -8.1653313719804235, 7.1389971101896279, -3.1937385492842112};
  for (int i = 0; i < 13; i++) quantize1(P, exp2c[i], &f->exp2_poly[i]);
  for (int i = 0; i < 15; i++) quantize1(P, logc[i], &f->log_poly[i]);
  quantize1(P, 0.5, &f->c_half);
  quantize1(P, 0.693147180559945309417232121458176568 /* 2.0f64.ln() */, &f->c_ln2);
  quantize1(P, 1.44269504088896340735992468100189214 /* std::f64::consts::LOG2_E */, &f->c_log2e);
  quantize1(P, 1.0, &f->c_one_q);
  /* generate_sin_poly (fixed_point.rs:189-211): "lolremez -d 14 -r 0:pi sin(x)", highest power first */
  static const double sinc[15] = {-1.1008071636607462e-11, 2.4208013888629323e-10, -3.8584805817996712e-10, -2.3786993104309845e-08,
                                  -2.9795813710683115e-09, 2.7608543130047009e-06, -6.4467066994122565e-09, -0.00019840680551418068,
                                  -3.839555844512214e-09, 0.0083333350601673614, -5.0943769725466814e-10, -0.16666666657583049,
                                  -8.5029878414113731e-12, 1.0000000000003146, -1.9323057584419828e-15};
  for (int i = 0; i < 15; i++) quantize1(P, sinc[i], &f->sin_poly[i]);
  const double pi = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
  quantize1(P, pi, &f->c_pi);
  quantize1(P, pi * 2.0, &f->c_two_pi);
  quantize1(P, 1.57079632679489661923132169163975144 /* FRAC_PI_2 */, &f->c_half_pi);
  quantize1(P, 2.0, &f->c_two);
}
static fpchip *chip_get(unsigned P, unsigned L) {
  for (int i = 0; i < g_nchips; i++)
    if (g_chips[i]->P == P && g_chips[i]->L == L) return g_chips[i];
  if (g_nchips == 1024) abort();
  fpchip *f = (fpchip *)malloc(sizeof(fpchip));
  orc_fp_init(f, P, L);
  g_chips[g_nchips++] = f;
  return f;
}

static fr_t fp_is_neg(G *g, const fr_t *a) { /* fixed_point.rs:523-539 */
  u256 b, one = {{1, 0, 0, 0}};
  u256_shl(&b, &one, 2 * g->f->P + 1);
  fr_t div, rem;
  r_div_mod(g, a, &b, 254, &div, &rem);
  fr_t is_pos = g_is_zero(g, &div);
  return g_not(g, &is_pos);
}
static fr_t fp_qabs(G *g, const fr_t *a) { /* :511-521 */
  fr_t rev = g_neg(g, a);
  fr_t n = fp_is_neg(g, a);
  return g_select(g, &rev, a, &n);
}
static fr_t fp_cond_neg(G *g, const fr_t *a, const fr_t *flag) { /* :541-556 */
  fr_t na = g_neg(g, a);
  return g_select(g, &na, a, flag);
}
static void fp_signed_div_scale(G *g, const fr_t *a, fr_t *div, fr_t *rem) { /* :974-1016 */
  unsigned P = g->f->P;
  u256 ac, t252, one = {{1, 0, 0, 0}}, q, r;
  fr_to_canonical(&ac, a);
  u256_shl(&t252, &one, 252);
  if (u256_cmp(&ac, &t252) > 0) {
    u256 aabs, cq, mask;
    u256_sub(&aabs, &ORC_FR.p, &ac); /* bn254_max - a + 1 */
    /* q = bn254_max - ceil(|a| / 2^P) + 1 = r - ceil(|a|/2^P) */
    u256_shl(&mask, &one, P);
    u256_sub(&mask, &mask, &one);
    u256 lowbits = {{aabs.l[0] & mask.l[0], aabs.l[1] & mask.l[1], 0, 0}};
    u256_shr(&cq, &aabs, P);
    if (!u256_is_zero(&lowbits)) u256_add(&cq, &cq, &one);
    u256_sub(&q, &ORC_FR.p, &cq);
    /* r = a - (2^P * q mod r) as integers */
    fr_t qf, bq;
    fr_from_canonical(&qf, &q);
    fr_mul(&bq, &g->f->scale, &qf);
    u256 bqc;
    fr_to_canonical(&bqc, &bq);
    if (u256_sub(&r, &ac, &bqc)) g->c->err = 1; /* BigUint subtraction underflow would panic */
  } else {
    u256_shr(&q, &ac, P);
    u256 mask;
    u256_shl(&mask, &one, P);
    u256_sub(&mask, &mask, &one);
    r = (u256){{ac.l[0] & mask.l[0], ac.l[1] & mask.l[1], 0, 0}};
  }
  fr_from_canonical(div, &q);
  fr_from_canonical(rem, &r);
  fr_t v[4] = {*rem, g->f->scale, *div, *a};
  region(g->c, v, 4, 1);
  u256 b;
  u256_shl(&b, &one, P);
  r_check_big_less_than_safe(g, rem, &b);
  u256 bound;
  u256_shl(&bound, &one, 3 * P);
  fr_t dabs = fp_qabs(g, div);
  r_check_big_less_than_safe(g, &dabs, &bound);
}
static fr_t fp_qmul(G *g, const fr_t *a, const fr_t *b) { /* :588-604 */
  fr_t ab = g_mul(g, a, b), q, r;
  fp_signed_div_scale(g, &ab, &q, &r);
  return q;
}
static fr_t fp_bit_xor(G *g, const fr_t *a, const fr_t *b) { /* :797-815 */
  fr_t a2 = g_add(g, ZERO, a);
  fr_t b2 = g_add(g, ZERO, b);
  g_assert_bit(g, &a2);
  g_assert_bit(g, &b2);
  fr_t ab = g_add(g, &a2, &b2);
  fr_t one = g_add(g, ONE, ZERO);
  return g_is_equal(g, &ab, &one);
}
static fr_t fp_qdiv(G *g, const fr_t *a, const fr_t *b) { /* :631-656 */
  unsigned P = g->f->P;
  fr_t sa = fp_is_neg(g, a);
  fr_t sb = fp_is_neg(g, b);
  fr_t aa = fp_qabs(g, a);
  fr_t ba = fp_qabs(g, b);
  fr_t ar = g_mul(g, &aa, &g->f->scale);
  fr_t q, r;
  r_div_mod_var(g, &ar, &ba, 4 * P, 2 * P, &q, &r);
  fr_t sx = fp_bit_xor(g, &sa, &sb);
  return fp_cond_neg(g, &q, &sx);
}
static fr_t fp_polynomial(G *g, const fr_t *x, const fr_t *coef, int m) { /* :658-686 */
  fr_t result = g_add(g, x, ZERO); /* dead qadd(x, 0) */
  fr_t last_y = *ZERO;
  for (int i = 0; i < m; i++) {
    fr_t y_add = g_add(g, &last_y, &coef[i]);
    if (i < m - 1) last_y = fp_qmul(g, x, &y_add);
    else result = y_add;
  }
  return result;
}
static void fp_check_power_of_two(G *g, const fr_t *p2, const fr_t *e) { /* :688-708 */
  unsigned nb = 2 * g->f->P;
  fr_t bits[128];
  g_num_to_bits(g, p2, nb, bits);
  fr_t s = g_sum(g, bits, nb);
  fr_t sm1 = g_sub(g, &s, ONE);
  g_is_zero(g, &sm1);
  fr_t bit = g_select_from_idx(g, bits, nb, e);
  fr_t bm1 = g_sub(g, &bit, ONE);
  g_is_zero(g, &bm1);
}
static fr_t fp_qlog2(G *g, const fr_t *a) { /* :736-795 */
  unsigned P = g->f->P;
  fr_t a_assigned = g_add(g, a, ZERO);
  fr_t is_neg = fp_is_neg(g, a);
  fr_t is_zero = g_is_zero(g, &a_assigned);
  g_or(g, &is_neg, &is_zero); /* assert_is_const(is_invalid, 0): no cells */
  u256 ac;
  fr_to_canonical(&ac, &a_assigned);
  unsigned nd = 1; /* fold(1, ...) keeps the index of the highest set bit */
  if (!u256_is_zero(&ac)) nd = u256_bits(&ac) - 1;
  fr_t pow1w = g_add(g, &g->f->pow2[nd], ZERO);
  fr_t ndf;
  fr_from_u64(&ndf, nd);
  fr_t exp1 = g_add(g, &ndf, ZERO);
  fp_check_power_of_two(g, &pow1w, &exp1);
  fr_t pow2w = g_mul(g, &pow1w, small(2));
  fr_t exp2 = g_add(g, &exp1, ONE);
  fp_check_power_of_two(g, &pow2w, &exp2);
  fr_t lt2 = r_is_less_than(g, a, &pow2w, 2 * P);
  fr_t gt1 = r_is_less_than(g, &pow1w, a, 2 * P);
  fr_t eq1 = g_is_equal(g, a, &pow1w);
  fr_t ge1 = g_or(g, &eq1, &gt1);
  g_and(g, &lt2, &ge1);
  fr_t pc;
  fr_from_u64(&pc, (uint64_t)P + 2);
  fr_t shift = g_sub(g, &pc, &exp2);
  fr_t shift_neg = fp_is_neg(g, &shift);
  fr_t shift_abs = fp_qabs(g, &shift);
  u256 sac;
  fr_to_canonical(&sac, &shift_abs);
  unsigned si = (unsigned)(sac.l[0] & 0xffffffffu);
  if (si >= 254) { /* Vec index out of bounds would panic upstream */
    g->c->err = 1;
    si = 0;
  }
  fr_t spw = g_add(g, &g->f->pow2[si], ZERO);
  fp_check_power_of_two(g, &spw, &shift_abs);
  fr_t a_ls = g_mul(g, a, &spw);
  fr_t a_rs, rr;
  r_div_mod_var(g, a, &spw, 2 * P, P + 1, &a_rs, &rr);
  fr_t a_norm = g_select(g, &a_rs, &a_ls, &shift_neg);
  fr_t log_norm = fp_polynomial(g, &a_norm, g->f->log_poly, 15);
  fr_t log_shift = g_neg(g, &shift);
  fr_t lsq = g_mul(g, &log_shift, &g->f->scale);
  return g_add(g, &log_norm, &lsq);
}
static fr_t fp_qexp2(G *g, const fr_t *a) { /* :710-734 */
  unsigned P = g->f->P;
  fr_t a_abs = fp_qabs(g, a);
  u256 sh, one = {{1, 0, 0, 0}};
  u256_shl(&sh, &one, P);
  fr_t ip, fpart;
  r_div_mod(g, &a_abs, &sh, 2 * P, &ip, &fpart);
  fr_t ip2 = g_select_from_idx(g, g->f->pow2, 254, &ip);
  fr_t yf = fp_polynomial(g, &fpart, g->f->exp2_poly, 13);
  fr_t res_pos = g_mul(g, &ip2, &yf);
  fr_t res_neg = fp_qdiv(g, &g->f->scale, &res_pos);
  fr_t n = fp_is_neg(g, a);
  return g_select(g, &res_neg, &res_pos, &n);
}
static fr_t fp_qlog(G *g, const fr_t *a) { /* :954-964 */
  fr_t l2e = load_constant(g, &g->f->c_log2e);
  fr_t l2a = fp_qlog2(g, a);
  return fp_qdiv(g, &l2a, &l2e);
}
static fr_t fp_qexp(G *g, const fr_t *a) { /* :876-886 */
  fr_t ln2 = load_constant(g, &g->f->c_ln2);
  fr_t x1 = fp_qdiv(g, a, &ln2);
  return fp_qexp2(g, &x1);
}
static fr_t fp_qpow(G *g, const fr_t *x, const fr_t *e) { /* :441-456 */
  fr_t lx = fp_qlog(g, x);
  fr_t al = fp_qmul(g, e, &lx);
  return fp_qexp(g, &al);
}
static fr_t fp_qsqrt(G *g, const fr_t *x) { /* :966-972 */
  fr_t half = load_constant(g, &g->f->c_half);
  return fp_qpow(g, x, &half);
}
static fr_t fp_qmin(G *g, const fr_t *a, const fr_t *b) { /* :936-952 */
  fr_t amb = g_sub(g, a, b);
  fr_t s = fp_is_neg(g, &amb);
  return g_select(g, a, b, &s);
}
static fr_t fp_qmax(G *g, const fr_t *a, const fr_t *b) { /* :918-934 */
  fr_t amb = g_sub(g, a, b);
  fr_t s = fp_is_neg(g, &amb);
  return g_select(g, b, a, &s);
}
/* ---- the rest of FixedPointInstructions: not reached from DistanceChip / VectorDBChip (examples/fixed_point.rs uses qsin) ---- */
static fr_t fp_sign(G *g, const fr_t *a) { /* :558-569 */
  fr_t neg_one = g_neg(g, ONE);
  fr_t is_neg = fp_is_neg(g, a);
  return g_select(g, &neg_one, ONE, &is_neg);
}
static fr_t fp_clip(G *g, const fr_t *a) { /* :571-586 */
  fr_t sign = fp_is_neg(g, a);
  fr_t a_abs = fp_qabs(g, a);
  u256 m, one = {{1, 0, 0, 0}};
  u256_shl(&m, &one, 2 * g->f->P); /* max_value = 2^(2P) */
  fr_t div, rem;
  r_div_mod(g, &a_abs, &m, 254, &div, &rem);
  return fp_cond_neg(g, &rem, &sign);
}
static fr_t fp_qmod(G *g, const fr_t *a, const fr_t *b) { /* :606-629; b must be positive (assert_is_const(b_sign, 0): no cells) */
  unsigned P = g->f->P;
  fr_t a_sign = fp_is_neg(g, a);
  (void)fp_is_neg(g, b);
  fr_t a_abs = fp_qabs(g, a);
  fr_t q, res_abs;
  r_div_mod_var(g, &a_abs, b, 4 * P, 2 * P, &q, &res_abs);
  fr_t comp = g_sub(g, b, &res_abs);
  return g_select(g, &comp, &res_abs, &a_sign);
}
static fr_t fp_qsin(G *g, const fr_t *a) { /* :817-841 */
  fr_t a_abs = fp_qabs(g, a);
  fr_t a_sign = fp_is_neg(g, a);
  fr_t a_mod = fp_qmod(g, &a_abs, &g->f->c_two_pi);
  fr_t a_mpi = g_sub(g, &a_mod, &g->f->c_pi);
  fr_t is_neg_a_mpi = fp_is_neg(g, &a_mpi);
  fr_t sin_a_mod = fp_polynomial(g, &a_mod, g->f->sin_poly, 15);
  fr_t sin_a_mpi_rev = fp_polynomial(g, &a_mpi, g->f->sin_poly, 15);
  fr_t sin_a_mpi = g_neg(g, &sin_a_mpi_rev);
  fr_t sin_a_abs = g_select(g, &sin_a_mod, &sin_a_mpi, &is_neg_a_mpi);
  return fp_cond_neg(g, &sin_a_abs, &a_sign);
}
static fr_t fp_qcos(G *g, const fr_t *a) { /* :843-852 */
  fr_t half_pi = load_constant(g, &g->f->c_half_pi);
  fr_t t = g_add(g, a, &half_pi);
  return fp_qsin(g, &t);
}
static fr_t fp_qtan(G *g, const fr_t *a) { /* :383-393 */
  fr_t s = fp_qsin(g, a);
  fr_t c = fp_qcos(g, a);
  return fp_qdiv(g, &s, &c);
}
static fr_t fp_sinh_cosh(G *g, const fr_t *a, int cosh) { /* :888-916 */
  fr_t ea = fp_qexp(g, a);
  fr_t na = g_neg(g, a);
  fr_t ena = fp_qexp(g, &na);
  fr_t nume = cosh ? g_add(g, &ea, &ena) : g_sub(g, &ea, &ena);
  fr_t two = load_constant(g, &g->f->c_two);
  return fp_qdiv(g, &nume, &two);
}
static fr_t fp_qtanh(G *g, const fr_t *a) { /* :407-417 */
  fr_t s = fp_sinh_cosh(g, a, 0);
  fr_t c = fp_sinh_cosh(g, a, 1);
  return fp_qdiv(g, &s, &c);
}
static fr_t fp_inner_product(G *g, const fr_t *a, const fr_t *b, size_t n) { /* :854-874 */
  fr_t res = g_add(g, ZERO, ZERO);
  for (size_t i = 0; i < n; i++) {
    fr_t t = fp_qmul(g, &a[i], &b[i]);
    res = g_add(g, &res, &t);
  }
  return res;
}

void orc_fp_op(octx *c, unsigned P, unsigned L, int op, const fr_t *a, const fr_t *b, fr_t *out) {
  G gg = {chip_get(P, L), c}, *g = &gg;
  fr_t r2;
  switch (op) {
    case ORC_OP_QADD: *out = g_add(g, a, b); break;
    case ORC_OP_QSUB: *out = g_sub(g, a, b); break;
    case ORC_OP_QMUL: *out = fp_qmul(g, a, b); break;
    case ORC_OP_QDIV: *out = fp_qdiv(g, a, b); break;
    case ORC_OP_NEG: *out = g_neg(g, a); break;
    case ORC_OP_QABS: *out = fp_qabs(g, a); break;
    case ORC_OP_IS_NEG: *out = fp_is_neg(g, a); break;
    case ORC_OP_QMIN: *out = fp_qmin(g, a, b); break;
    case ORC_OP_QMAX: *out = fp_qmax(g, a, b); break;
    case ORC_OP_QSQRT: *out = fp_qsqrt(g, a); break;
    case ORC_OP_QLOG2: *out = fp_qlog2(g, a); break;
    case ORC_OP_QEXP2: *out = fp_qexp2(g, a); break;
    case ORC_OP_QLOG: *out = fp_qlog(g, a); break;
    case ORC_OP_QEXP: *out = fp_qexp(g, a); break;
    case ORC_OP_QPOW: *out = fp_qpow(g, a, b); break;
    case ORC_OP_BIT_XOR: *out = fp_bit_xor(g, a, b); break;
    case ORC_OP_COND_NEG: *out = fp_cond_neg(g, a, b); break;
    case ORC_OP_SIGNED_DIV_SCALE: fp_signed_div_scale(g, a, out, &r2); break;
    case ORC_OP_SIGN: *out = fp_sign(g, a); break;
    case ORC_OP_CLIP: *out = fp_clip(g, a); break;
    case ORC_OP_QMOD: *out = fp_qmod(g, a, b); break;
    case ORC_OP_QSIN: *out = fp_qsin(g, a); break;
    case ORC_OP_QCOS: *out = fp_qcos(g, a); break;
    case ORC_OP_QTAN: *out = fp_qtan(g, a); break;
    case ORC_OP_QSINH: *out = fp_sinh_cosh(g, a, 0); break;
    case ORC_OP_QCOSH: *out = fp_sinh_cosh(g, a, 1); break;
    case ORC_OP_QTANH: *out = fp_qtanh(g, a); break;
    default: c->err = 2;
  }
}

/* ================================================================== DistanceChip (distance.rs) */
static fr_t dist_euclidean(G *g, const fr_t *a, const fr_t *b, size_t n) { /* :97-119 */
  fr_t *ab = (fr_t *)malloc(sizeof(fr_t) * n);
  for (size_t i = 0; i < n; i++) ab[i] = g_sub(g, &a[i], &b[i]);
  fr_t ds = fp_inner_product(g, ab, ab, n);
  free(ab);
  return fp_qsqrt(g, &ds);
}
static fr_t dist_cosine(G *g, const fr_t *a, const fr_t *b, size_t n) { /* :121-144 */
  fr_t ab = fp_inner_product(g, a, b, n);
  fr_t aa = fp_inner_product(g, a, a, n);
  fr_t bb = fp_inner_product(g, b, b, n);
  fr_t as = fp_qsqrt(g, &aa);
  fr_t bs = fp_qsqrt(g, &bb);
  fr_t den = fp_qmul(g, &as, &bs);
  fr_t sim = fp_qdiv(g, &ab, &den);
  fr_t one = load_constant(g, &g->f->c_one_q);
  return g_sub(g, &one, &sim);
}
static fr_t dist_manhattan(G *g, const fr_t *a, const fr_t *b, size_t n) { /* :177-195 */
  fr_t *d = (fr_t *)malloc(sizeof(fr_t) * n);
  for (size_t i = 0; i < n; i++) d[i] = g_sub(g, &a[i], &b[i]);
  for (size_t i = 0; i < n; i++) d[i] = fp_qabs(g, &d[i]);
  fr_t s = g_sum(g, d, n);
  free(d);
  return s;
}
static fr_t dist_hamming(G *g, const fr_t *a, const fr_t *b, size_t n) { /* :146-175 */
  fr_t *e = (fr_t *)malloc(sizeof(fr_t) * n);
  for (size_t i = 0; i < n; i++) e[i] = g_is_equal(g, &a[i], &b[i]);
  fr_t s = g_sum(g, e, n);
  free(e);
  fr_t len, sq;
  quantize1(g->f->P, (double)n, &len);
  push(g->c, &len, 0); /* load_witness */
  u256 sc;
  fr_to_canonical(&sc, &s);
  u128 lo = ((u128)sc.l[1] << 64) | sc.l[0];
  quantize1(g->f->P, (double)lo, &sq);
  push(g->c, &sq, 0);
  fr_t sim = fp_qdiv(g, &sq, &len);
  fr_t one = load_constant(g, &g->f->c_one_q);
  return g_sub(g, &one, &sim);
}
static fr_t distance(G *g, int metric, const fr_t *a, const fr_t *b, size_t n) {
  switch (metric) {
    case ORC_METRIC_EUCLIDEAN: return dist_euclidean(g, a, b, n);
    case ORC_METRIC_COSINE: return dist_cosine(g, a, b, n);
    case ORC_METRIC_MANHATTAN: return dist_manhattan(g, a, b, n);
    case ORC_METRIC_HAMMING: return dist_hamming(g, a, b, n);
  }
  g->c->err = 2;
  return *ZERO;
}
void orc_distance(octx *c, unsigned P, unsigned L, int metric, const fr_t *a, const fr_t *b, size_t dim, fr_t *out) {
  G gg = {chip_get(P, L), c};
  *out = distance(&gg, metric, a, b, dim);
}
void orc_inner_product(octx *c, unsigned P, unsigned L, const fr_t *a, const fr_t *b, size_t dim, fr_t *out) {
  G gg = {chip_get(P, L), c};
  *out = fp_inner_product(&gg, a, b, dim);
}
void orc_assign_witnesses(octx *c, const fr_t *v, size_t n) {
  for (size_t i = 0; i < n; i++) push(c, &v[i], 0);
}

/* ================================================================== VectorDBChip (vectordb.rs) */
void orc_nearest_vector(octx *c, unsigned P, unsigned L, int metric, const fr_t *query, const fr_t *vectors,
                        size_t n, size_t dim, fr_t *ind, fr_t *result) { /* :122-163 */
  G gg = {chip_get(P, L), c}, *g = &gg;
  fr_t *d = (fr_t *)malloc(sizeof(fr_t) * n);
  for (size_t i = 0; i < n; i++) d[i] = distance(g, metric, vectors + i * dim, query, dim); /* (v, query) */
  fr_t min = d[0];
  for (size_t i = 1; i < n; i++) min = fp_qmin(g, &min, &d[i]);
  for (size_t i = 0; i < n; i++) ind[i] = g_is_equal(g, &min, &d[i]);
  for (size_t j = 0; j < dim; j++) result[j] = g_select_by_indicator(g, vectors + j, dim, ind, n);
  free(d);
}
void orc_kmeans(octx *c, unsigned P, unsigned L, int metric, const fr_t *vectors, size_t n, size_t dim,
                size_t K, size_t I, fr_t *cent, fr_t *inds) { /* :225-362 */
  G gg = {chip_get(P, L), c}, *g = &gg;
  if (!(K < n)) {
    c->err = 3;
    return;
  }
  fr_t one = load_constant(g, &g->f->c_one_q);
  fr_t zero = load_zero(g);
  memcpy(cent, vectors, sizeof(fr_t) * K * dim);
  fr_t *dist = (fr_t *)malloc(sizeof(fr_t) * K);
  fr_t *sizes = (fr_t *)malloc(sizeof(fr_t) * K);
  fr_t *filt = (fr_t *)malloc(sizeof(fr_t) * n * dim);
  fr_t *mean = (fr_t *)malloc(sizeof(fr_t) * K * dim);
  for (size_t it = 0; it < I; it++) {
    for (size_t v = 0; v < n; v++) {
      for (size_t k = 0; k < K; k++) dist[k] = distance(g, metric, cent + k * dim, vectors + v * dim, dim); /* (c, v) */
      fr_t min = dist[0];
      for (size_t k = 1; k < K; k++) min = fp_qmin(g, &min, &dist[k]);
      for (size_t k = 0; k < K; k++) {
        fr_t eq = g_is_equal(g, &min, &dist[k]);
        inds[v * K + k] = g_select(g, &one, &zero, &eq);
      }
    }
    memcpy(sizes, inds, sizeof(fr_t) * K);
    for (size_t v = 1; v < n; v++)
      for (size_t k = 0; k < K; k++) sizes[k] = g_add(g, &sizes[k], &inds[v * K + k]);
    for (size_t k = 0; k < K; k++) {
      for (size_t v = 0; v < n; v++) {
        fr_t iz = g_is_zero(g, &inds[v * K + k]);
        for (size_t j = 0; j < dim; j++) filt[v * dim + j] = g_select(g, &zero, &vectors[v * dim + j], &iz);
      }
      fr_t *sum = mean + k * dim;
      memcpy(sum, filt, sizeof(fr_t) * dim);
      for (size_t v = 1; v < n; v++)
        for (size_t j = 0; j < dim; j++) sum[j] = g_add(g, &filt[v * dim + j], &sum[j]); /* qadd(vector_j, sum_j) */
      for (size_t j = 0; j < dim; j++) sum[j] = fp_qdiv(g, &sum[j], &sizes[k]);
      /* centroids[cluster_id] = mean — visible to later clusters only through next iteration's distances */
    }
    memcpy(cent, mean, sizeof(fr_t) * K * dim);
  }
  free(dist);
  free(sizes);
  free(filt);
  free(mean);
}

/* ---- PoseidonChip trace (poseidon chip of halo2-lib community-edition, [UPSTREAM-RECALL]) ---- */
typedef struct {
  G *g;
  const psd_spec *s;
  fr_t st[PSD_MAX_T];
} pchip;
static void pc_sbox(pchip *p, int i, const fr_t *cst) { /* x_power5_with_constant */
  G *g = p->g;
  fr_t x = p->st[i];
  fr_t x2 = g_mul(g, &x, &x);
  fr_t x4 = g_mul(g, &x2, &x2);
  p->st[i] = g_mul_add(g, &x, &x4, cst);
}
static void pc_mds(pchip *p, const fr_t m[PSD_MAX_T][PSD_MAX_T]) {
  fr_t r[PSD_MAX_T];
  for (int i = 0; i < p->s->t; i++) r[i] = g_inner_product(p->g, p->st, m[i], (size_t)p->s->t, 1);
  memcpy(p->st, r, sizeof(fr_t) * (size_t)p->s->t);
}
static void pc_permutation(pchip *p, const fr_t *in, int n_in) {
  G *g = p->g;
  const psd_spec *s = p->s;
  int t = s->t, half = s->r_f / 2;
  /* absorb_with_pre_constants */
  {
    fr_t v[2] = {p->st[0], s->start[0][0]};
    p->st[0] = g_sum(g, v, 2);
  }
  for (int i = 0; i < n_in; i++) {
    fr_t v[3] = {p->st[1 + i], in[i], s->start[0][1 + i]};
    p->st[1 + i] = g_sum(g, v, 3);
  }
  for (int i = n_in + 1, k = 0; i < t; i++, k++) {
    fr_t cst = s->start[0][i];
    if (k == 0) fr_add(&cst, &cst, &g->f->one);
    fr_t v[2] = {p->st[i], cst};
    p->st[i] = g_sum(g, v, 2);
  }
  for (int r = 1; r < half; r++) {
    for (int i = 0; i < t; i++) pc_sbox(p, i, &s->start[r][i]);
    pc_mds(p, s->mds);
  }
  for (int i = 0; i < t; i++) pc_sbox(p, i, &s->start[half][i]);
  pc_mds(p, s->pre_sparse_mds);
  for (int q = 0; q < s->r_p; q++) {
    pc_sbox(p, 0, &s->partial[q]);
    fr_t r[PSD_MAX_T];
    r[0] = g_inner_product(g, p->st, s->sparse_row[q], (size_t)t, 1);
    for (int i = 1; i < t; i++) r[i] = g_mul_add(g, &p->st[0], &s->sparse_col_hat[q][i - 1], &p->st[i]);
    memcpy(p->st, r, sizeof(fr_t) * (size_t)t);
  }
  for (int r = 0; r < half - 1; r++) {
    for (int i = 0; i < t; i++) pc_sbox(p, i, &s->end[r][i]);
    pc_mds(p, s->mds);
  }
  for (int i = 0; i < t; i++) pc_sbox(p, i, &g->f->zero);
  pc_mds(p, s->mds);
}
static fr_t pc_hash(pchip *p, const fr_t *msg, size_t len) { /* clear(); update(msg); squeeze() */
  memset(p->st, 0, sizeof(p->st));
  u256 cap = {{0, 1, 0, 0}};
  fr_from_canonical(&p->st[0], &cap);
  size_t rate = (size_t)p->s->rate, off = 0;
  int pad = 0;
  while (off < len) {
    size_t cn = len - off < rate ? len - off : rate;
    pad = (int)(rate - cn);
    pc_permutation(p, msg + off, (int)cn);
    off += cn;
  }
  if (pad == 0) pc_permutation(p, NULL, 0);
  return p->st[1];
}
void orc_poseidon_chip_new(octx *c, int t) { /* PoseidonChip::new: T load_constant cells [2^64, 0, ...] */
  orc_init();
  fr_t w;
  u256 cap = {{0, 1, 0, 0}};
  fr_from_canonical(&w, &cap);
  push(c, &w, 0);
  fr_t z;
  memset(&z, 0, sizeof(z));
  for (int i = 1; i < t; i++) push(c, &z, 0);
}
void orc_merkle_commitment(octx *c, int t, int r_f, int r_p, const fr_t *vectors, size_t n, size_t dim, fr_t *root) {
  /* vectordb.rs:165-223 */
  G gg = {chip_get(48, 13), c}; /* P, L irrelevant: only GateChip primitives are used */
  pchip p;
  p.g = &gg;
  p.s = psd_get_spec(t, r_f, r_p);
  size_t leaves = 1;
  while (leaves < n) leaves <<= 1;
  fr_t *lv = (fr_t *)calloc(leaves, sizeof(fr_t));
  for (size_t i = 0; i < n; i++) lv[i] = pc_hash(&p, vectors + i * dim, dim);
  if (leaves > n) load_zero(&gg);
  while (leaves > 1) {
    for (size_t i = 0; i < leaves; i += 2) {
      fr_t pair[2] = {lv[i], lv[i + 1]};
      lv[i / 2] = pc_hash(&p, pair, 2);
    }
    leaves >>= 1;
  }
  *root = lv[0];
  free(lv);
}

/* ================================================================== checks & layout */
size_t orc_check_gates(const octx *c, unsigned L) {
  if (!c->store || !c->keygen) return (size_t)-1;
  const fr_t *a = orc_ctx_advice(c);
  size_t bad = 0;
  for (size_t i = 0; i < c->n; i++) {
    if (!c->sel[i]) continue;
    if (i + 3 >= c->n) {
      bad++;
      continue;
    }
    fr_t t;
    fr_mul(&t, &a[i + 1], &a[i + 2]);
    fr_add(&t, &t, &a[i]);
    if (!fr_eq(&t, &a[i + 3])) bad++;
  }
  for (size_t i = 0; i < c->nl; i++) {
    u256 v;
    fr_to_canonical(&v, &c->lookup[i]);
    if (u256_bits(&v) > L) bad++;
  }
  return bad;
}
size_t orc_layout_columns(const fr_t *stream, size_t n_cells, const size_t *bp, size_t nbp, unsigned k,
                          fr_t *cols, size_t cap) {
  size_t rows = (size_t)1 << k, col = 0, row = 0, bi = 0;
  if (cap == 0) return 0;
  memset(cols, 0, sizeof(fr_t) * rows * cap);
  for (size_t i = 0; i < n_cells; i++) {
    if (row >= rows) return 0;
    cols[col * rows + row] = stream[i];
    if (bi < nbp && bp[bi] == row) {
      bi++;
      row = 0;
      col++;
      if (col >= cap) return 0;
      cols[col * rows] = stream[i];
    }
    row++;
  }
  return col + 1;
}
size_t orc_layout_lookup(const fr_t *lookup, size_t n_cells, unsigned k, unsigned minimum_rows, fr_t *cols, size_t cap) {
  size_t rows = (size_t)1 << k, max_rows = rows - minimum_rows, col = 0, off = 0;
  if (cap == 0) return 0;
  memset(cols, 0, sizeof(fr_t) * rows * cap);
  for (size_t i = 0; i < n_cells; i++) {
    if (off >= max_rows) {
      off = 0;
      col++;
      if (col >= cap) return 0;
    }
    cols[col * rows + off] = lookup[i];
    off++;
  }
  return n_cells ? col + 1 : 0;
}
