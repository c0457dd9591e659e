/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the BN254 arithmetic that the reference reaches through
 * un-vendored third-party crates (halo2curves bn256 via halo2_base::halo2_proofs,
 * /root/reference/src/scaffold/mod.rs:17-31, Cargo.toml:19-28).  The crates are
 * absent from /root/reference, so this file restates the PUBLIC algorithms
 * (Montgomery arithmetic, short-Weierstrass Jacobian group law, Pippenger MSM as in
 * halo2 `best_multiexp`, radix-2 NTT as in halo2 `best_fft`).
 *
 * PARITY UNPINNED against the real Rust reference (cannot be built here);
 * pinned instead by public BN254 constants and algebraic invariants
 * plus an independent Python big-int implementation (oracle/pyref.py).
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } u256;
typedef u256 fr_t; /* Montgomery form, R = 2^256, little-endian limbs (halo2curves layout) */
typedef u256 fq_t;

typedef struct {
  u256 p;       /* modulus */
  u256 r1;      /* R mod p   (Montgomery one) */
  u256 r2;      /* R^2 mod p */
  uint64_t inv; /* -p^{-1} mod 2^64 */
} mont_ctx;

extern mont_ctx ORC_FR, ORC_FQ;
void orc_init(void);

/* ---- 256-bit integer helpers (canonical integers) ---- */
int u256_cmp(const u256 *a, const u256 *b);
int u256_is_zero(const u256 *a);
uint64_t u256_add(u256 *o, const u256 *a, const u256 *b); /* returns carry */
uint64_t u256_sub(u256 *o, const u256 *a, const u256 *b); /* returns borrow */
void u256_shr(u256 *o, const u256 *a, unsigned s);
void u256_shl(u256 *o, const u256 *a, unsigned s);
unsigned u256_bits(const u256 *a); /* bit length, 0 for zero */
void u256_divmod(u256 *q, u256 *r, const u256 *a, const u256 *b); /* b != 0 */
void u256_mul_wide(uint64_t out[8], const u256 *a, const u256 *b);
void u256_set_u64(u256 *o, uint64_t v);

/* ---- field arithmetic (generic over mont_ctx) ---- */
void mont_mul(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m);
void mont_add(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m);
void mont_sub(u256 *o, const u256 *a, const u256 *b, const mont_ctx *m);
void mont_neg(u256 *o, const u256 *a, const mont_ctx *m);
void mont_from_canonical(u256 *o, const u256 *a, const mont_ctx *m);
void mont_to_canonical(u256 *o, const u256 *a, const mont_ctx *m);
void mont_pow(u256 *o, const u256 *a, const u256 *e, const mont_ctx *m);
void mont_inv(u256 *o, const u256 *a, const mont_ctx *m); /* 0 -> 0 */

/* Fr conveniences */
void fr_mul(fr_t *o, const fr_t *a, const fr_t *b);
void fr_add(fr_t *o, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *o, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *o, const fr_t *a);
void fr_inv(fr_t *o, const fr_t *a);
void fr_from_u64(fr_t *o, uint64_t v);
void fr_from_canonical(fr_t *o, const u256 *a);
void fr_to_canonical(u256 *o, const fr_t *a);
int fr_eq(const fr_t *a, const fr_t *b);
int fr_is_zero(const fr_t *a);
/* batch helpers exported for ctypes tests */
void orc_fr_mul_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n);
void orc_fr_add_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n);
void orc_fr_sub_batch(fr_t *o, const fr_t *a, const fr_t *b, size_t n);
void orc_fr_inv_batch(fr_t *o, const fr_t *a, size_t n);
void orc_eval_poly(fr_t *out, const fr_t *a, size_t n, const fr_t *x);
void orc_grand_product(fr_t *z, const fr_t *num, const fr_t *den, size_t n);
void orc_fr_from_canonical_batch(fr_t *o, const u256 *a, size_t n);
void orc_fr_to_canonical_batch(u256 *o, const fr_t *a, size_t n);
void orc_fq_from_canonical_batch(fq_t *o, const u256 *a, size_t n);
void orc_fq_to_canonical_batch(u256 *o, const fq_t *a, size_t n);
void orc_fr_root_of_unity(fr_t *o, unsigned k); /* primitive 2^k-th root: ROOT_OF_UNITY^(2^(28-k)) */
void orc_fr_zeta(fr_t *o);                      /* halo2curves bn256 Fr::ZETA (cube root of unity) */

/* ---- G1: y^2 = x^3 + 3 over Fq ---- */
typedef struct { fq_t x, y; } g1_affine;      /* identity encoded as (0,0), as halo2curves */
typedef struct { fq_t x, y, z; } g1_jac;      /* identity: z == 0 */

void g1_set_identity(g1_jac *o);
int g1_is_identity(const g1_jac *a);
void g1_from_affine(g1_jac *o, const g1_affine *a);
void g1_to_affine(g1_affine *o, const g1_jac *a);
void g1_double(g1_jac *o, const g1_jac *a);
void g1_add(g1_jac *o, const g1_jac *a, const g1_jac *b);
void g1_add_mixed(g1_jac *o, const g1_jac *a, const g1_affine *b);
void g1_neg_affine(g1_affine *o, const g1_affine *a);
void g1_scalar_mul(g1_jac *o, const g1_affine *p, const u256 *k_canonical);
int g1_affine_on_curve(const g1_affine *a);
void orc_g1_generator(g1_affine *o);
/* bases[i] = (s_i)*G for canonical scalars s_i (test SRS construction) */
void orc_g1_mul_generator_batch(g1_affine *out, const u256 *scalars_canonical, size_t n);
/* test SRS: monomial g[i] = tau^i G, lagrange gl[i] = L_i(tau) G over the 2^k domain */
void orc_srs_from_tau(g1_affine *g, g1_affine *g_lagrange, unsigned k, const u256 *tau_canonical);

/* ---- MSM ---- */
/* scalars in Montgomery form (as halo2curves &[Fr]); result canonical affine */
void orc_msm_naive(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n);
/* halo2 `best_multiexp` algorithm: per-thread chunks, serial Pippenger with c = ln(n) windows */
void orc_msm_pippenger(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n, int threads);
/* columns: n_cols pointers-free layout, scalars[col*n + i]; parallel over columns */
void orc_msm_batch(g1_affine *out, const fr_t *scalars, const g1_affine *bases, size_t n, size_t n_cols, int threads);

/* ---- NTT ---- */
/* halo2 `best_fft`: in place, natural order in / natural order out */
void orc_ntt(fr_t *a, unsigned log_n, const fr_t *omega);
void orc_ntt_naive_dft(fr_t *out, const fr_t *a, unsigned log_n, const fr_t *omega); /* O(n^2) */
/* EvaluationDomain::lagrange_to_coeff: iNTT incl. 1/n scaling */
void orc_lagrange_to_coeff(fr_t *a, unsigned k);
/* EvaluationDomain::coeff_to_extended: zeta-coset scaling, zero-pad to 2^(k+ext), forward NTT.
 * in: n=2^k coeffs; out: 2^(k+ext) evaluations */
void orc_coeff_to_extended(fr_t *out, const fr_t *coeffs, unsigned k, unsigned ext);
void orc_extended_to_coeff(fr_t *a, unsigned k, unsigned ext);
void orc_ntt_batch(fr_t *cols, size_t n_cols, unsigned log_n, const fr_t *omega, int threads);
void orc_lde_batch(fr_t *ext_out, fr_t *cols_inout, size_t n_cols, unsigned k, unsigned ext, int threads);

#ifdef __cplusplus
}
#endif
#endif
