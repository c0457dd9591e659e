/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * CPU restatement of the witness stream that the reference's gadgets emit:
 *   - /root/reference/src/gadget/fixed_point.rs   (FixedPointChip)
 *   - /root/reference/src/gadget/distance.rs      (DistanceChip)
 *   - /root/reference/src/gadget/vectordb.rs      (VectorDBChip)
 * on top of a restatement of the halo2-base v0.3 ("community-edition", Cargo.toml:22) primitive
 * cell templates (Context / GateChip / RangeChip) and the poseidon chip (Cargo.toml:24).
 * halo2-base and poseidon are NOT vendored in /root/reference: their templates are restated from
 * the published crate [UPSTREAM-RECALL] => cell ORDER parity is UNPINNED; cell VALUES of gadget
 * results are pinned to the reference's own f64 tolerance tests (tests/distances_test.rs,
 * tests/vectordb_test.rs: rel 1e-6) and the hand-derived constants in SURVEY App. E.
 */
#ifndef ORACLE_GADGETS_H
#define ORACLE_GADGETS_H
#include "bn254.h"
#include "poseidon.h"

#ifdef __cplusplus
extern "C" {
#endif

/* halo2-base Context restated: flat advice stream + cells_to_lookup queue (+ selector bits) */
typedef struct {
  fr_t *advice;     /* NULL when store == 0 (count only) */
  uint8_t *sel;     /* gate-start bits, only when keygen != 0 && store != 0 */
  size_t n, cap;
  fr_t *lookup;     /* values of cells_to_lookup, in queue order */
  size_t nl, capl;
  int store, keygen;
  int has_zero;     /* Context::load_zero caches one zero cell */
  int err;          /* reference would have panicked (division by zero, ...) */
  /* streaming break-point planner (GateThreadBuilder::assign_all row logic) */
  int plan;
  size_t max_rows, row;
  size_t *bp;
  size_t nbp, capbp;
} octx;

octx *orc_ctx_new(int store, int keygen);
void orc_ctx_enable_plan(octx *c, unsigned k, unsigned minimum_rows);
void orc_ctx_free(octx *c);
size_t orc_ctx_len(const octx *c);
size_t orc_ctx_lookup_len(const octx *c);
const fr_t *orc_ctx_advice(const octx *c);
const fr_t *orc_ctx_lookup(const octx *c);
const uint8_t *orc_ctx_selectors(const octx *c);
int orc_ctx_err(const octx *c);
size_t orc_ctx_break_points(const octx *c, size_t *out, size_t cap);

/* FixedPointChip<F, P> (fixed_point.rs:42-98) */
typedef struct {
  unsigned P, L;
  fr_t scale;          /* 2^P */
  fr_t negative_point; /* r - 2^(2P+1) */
  u256 negative_point_c;
  fr_t pow2[254];
  fr_t one, zero;
  fr_t exp2_poly[13], log_poly[15];
  fr_t c_half, c_ln2, c_log2e, c_one_q;
  fr_t sin_poly[15], c_pi, c_two_pi, c_half_pi, c_two; /* fixed_point.rs:189-211, 817-916 */
} fpchip;

void orc_fp_init(fpchip *f, unsigned P, unsigned L);
void orc_fp_quantize(unsigned P, const double *x, fr_t *out, size_t n);   /* fixed_point.rs:104-119 */
void orc_fp_dequantize(unsigned P, const fr_t *x, double *out, size_t n); /* fixed_point.rs:121-136 */

/* ---- single-op entry points (emit into ctx, return result) for unit parity tests ---- */
enum {
  ORC_OP_QADD = 0, ORC_OP_QSUB, ORC_OP_QMUL, ORC_OP_QDIV, ORC_OP_NEG, ORC_OP_QABS, ORC_OP_IS_NEG,
  ORC_OP_QMIN, ORC_OP_QSQRT, ORC_OP_QLOG2, ORC_OP_QEXP2, ORC_OP_QLOG, ORC_OP_QEXP, ORC_OP_QPOW,
  ORC_OP_BIT_XOR, ORC_OP_COND_NEG, ORC_OP_SIGNED_DIV_SCALE, ORC_OP_QMAX, ORC_OP_SIGN, ORC_OP_CLIP, ORC_OP_QMOD, ORC_OP_QSIN, ORC_OP_QCOS,
  ORC_OP_QTAN, ORC_OP_QSINH, ORC_OP_QCOSH, ORC_OP_QTANH
};
void orc_fp_op(octx *c, unsigned P, unsigned L, int op, const fr_t *a, const fr_t *b, fr_t *out);

enum { ORC_METRIC_EUCLIDEAN = 0, ORC_METRIC_COSINE = 1, ORC_METRIC_MANHATTAN = 2, ORC_METRIC_HAMMING = 3 };
/* DistanceChip (distance.rs:97-195): a,b already-assigned quantized vectors */
void orc_distance(octx *c, unsigned P, unsigned L, int metric, const fr_t *a, const fr_t *b, size_t dim, fr_t *out);
void orc_inner_product(octx *c, unsigned P, unsigned L, const fr_t *a, const fr_t *b, size_t dim, fr_t *out);

/* VectorDBChip (vectordb.rs:122-362).  `vectors` row-major n x dim. */
void orc_nearest_vector(octx *c, unsigned P, unsigned L, int metric, const fr_t *query, const fr_t *vectors,
                        size_t n, size_t dim, fr_t *indicator_out /*n*/, fr_t *result_out /*dim*/);
void orc_kmeans(octx *c, unsigned P, unsigned L, int metric, const fr_t *vectors, size_t n, size_t dim,
                size_t K, size_t I, fr_t *centroids_out /*K*dim*/, fr_t *indicators_out /*n*K*/);
/* PoseidonChip::new emits T load_constant cells; merkle_commitment emits the hash trace */
void orc_poseidon_chip_new(octx *c, int t);
void orc_merkle_commitment(octx *c, int t, int r_f, int r_p, const fr_t *vectors, size_t n, size_t dim, fr_t *root);
/* Context::assign_witnesses */
void orc_assign_witnesses(octx *c, const fr_t *v, size_t n);

/* MockProver-like check of the part of the circuit this oracle models: every selected row
 * satisfies a + b*c - d = 0 and every lookup cell is < 2^L.  Returns number of violations. */
size_t orc_check_gates(const octx *c, unsigned L);

/* Prover-side layout (halo2-base assign_threads_in): stream + break points -> columns of 2^k rows.
 * cols_out is n_cols x 2^k (zero filled).  Returns number of columns used, or 0 on overflow. */
size_t orc_layout_columns(const fr_t *stream, size_t n_cells, const size_t *bp, size_t nbp, unsigned k,
                          fr_t *cols_out, size_t n_cols_cap);
size_t orc_layout_lookup(const fr_t *lookup, size_t n_cells, unsigned k, unsigned minimum_rows,
                         fr_t *cols_out, size_t n_cols_cap);

#ifdef __cplusplus
}
#endif
#endif
