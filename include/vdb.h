/*
 * vdb.h — C ABI of the MI355X-native proving hot path for halo2-vectordb circuits.
 *
 * This is the drop-in boundary.  The reference (Rust) has no FFI of its own: the seam
 * is the set of upstream Rust functions its prover reaches through
 *   /root/reference/src/scaffold/mod.rs:296  gen_snark_shplonk -> create_proof   (MSM, NTT)
 *   /root/reference/src/scaffold/mod.rs:273  create_pk                            (MSM, NTT, keygen)
 *   /root/reference/src/scaffold/mod.rs:378-396  closure + RangeCircuitBuilder::prover (witness, layout)
 * A patched halo2-axiom / halo2-base (Cargo `[patch]`) forwards those calls here; INTEGRATION.md shows
 * the Rust `extern "C"` stubs.  Each entry point below names the interface it replaces.
 *
 * Conventions
 *  - vdb_fr  : BN254 scalar, 4 x u64 little-endian limbs, MONTGOMERY form (R = 2^256) — the in-memory
 *              layout of halo2curves::bn256::Fr, so `&[Fr]` crosses zero-copy.
 *  - vdb_g1  : affine point {x, y}, each a Montgomery Fq; identity = (0, 0) as halo2curves.
 *  - Every function returns 0 on success or a negative VDB_ERR_*; nothing aborts or throws across
 *    the ABI.  vdb_last_error() returns a thread-local message for the last failure.
 *  - Host pointers unless the name ends in _dev (then: device pointers in HBM of the bound GPU).
 *  - A process binds one GPU (vdb_init(device): one process per GPU) or several (vdb_init_devices(n): one context per
 *    device, the calling thread's device chosen with vdb_set_device).  Calls are not re-entrant per device context,
 *    matching the reference's single prover thread.  Entry points with host outputs block until the result is there;
 *    _dev entry points only queue work on the library's stream (in order), and the vdb_msm_batch_*_begin /
 *    vdb_msm_batch_end pair is explicitly asynchronous (see there).  vdb_sync() waits for everything queued.
 */
#ifndef VDB_H
#define VDB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } vdb_fr;
typedef struct { uint64_t l[4]; } vdb_fq;
typedef struct { vdb_fq x, y; } vdb_g1;
typedef struct vdb_srs vdb_srs;

enum {
  VDB_OK = 0,
  VDB_ERR_NOT_INIT = -1,
  VDB_ERR_HIP = -2,       /* HIP runtime failure (message has the hipError string) */
  VDB_ERR_ARG = -3,       /* bad argument (size mismatch, null pointer, unsupported parameter) */
  VDB_ERR_OOM = -4,
  VDB_ERR_DOMAIN = -5,    /* data-dependent failure the reference turns into a panic (div by zero, K >= N) */
  VDB_ERR_NO_DEVICE = -6
};

/* ---- b0 lifecycle -------------------------------------------------------------------------- */
/* Binds this process to HIP device `device` and creates streams/workspaces (whatever was bound before is released).
 * Fails loudly with VDB_ERR_NO_DEVICE when no GPU is visible: there is NO CPU fallback in this library. */
int vdb_init(int device);
/* binds devices 0 .. n_devices-1 in ONE process, each with its own context (streams, workspaces, twiddle and
 * gadget tables).  A host thread works on the device it selected with vdb_set_device (thread-local; device 0 until then);
 * device pointers, vdb_srs handles and deferred MSMs belong to the device they were created on.  The path shards by
 * column with no exchange between devices: the *_multi entry points below cut a batch of columns into one
 * block per device and return every device's results D2H; no RCCL communicator is needed for that.  The other supported
 * arrangement is one process per GPU (vdb_init(local_rank), torch.distributed / RCCL all_gather of the commitments:
 * bench.py). */
int vdb_init_devices(int n_devices);
int vdb_devices_bound(void);      /* number of bound devices */
int vdb_set_device(int device);   /* the calling thread works on this (bound) device from now on */
int vdb_current_device(void);     /* -1 when nothing is bound */
void vdb_shutdown(void);          /* releases every bound device */
const char *vdb_last_error(void);
int vdb_device_count(void);
const char *vdb_version(void);

/* device memory helpers for HBM-resident pipelines */
int vdb_malloc(void **dptr, size_t bytes);
int vdb_free(void *dptr);
int vdb_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int vdb_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
/* Frees the library's cached work buffers (MSM buckets and sort records — up to half of the free HBM —, NTT staging); they
 * are re-created on demand.  For callers that need the memory between two phases.  Waits for queued work. */
int vdb_scratch_release(void);
/* Bytes the cached work buffers hold at the moment (what vdb_scratch_release would give back). */
int vdb_scratch_held(size_t *bytes);
/* Wall time, bytes and number of the device allocations this process made through the library (vdb_malloc, the work buffers, keygen's
 * sort records) since start or the last reset: what a setup / keygen time consists of when HBM has to be mapped or cleared first. */
int vdb_alloc_stats(double *seconds, uint64_t *bytes, uint64_t *calls, int reset);
/* Upper bound for the MSM's work space from now on (0 = the default: half of the free HBM, between 8 and 96 GiB — one batch for
 * the whole k = 16 job).  A keygen that commits the fixed columns while the card is still empty sets a bound first: mapping 96 GiB
 * of fresh HBM for a two-second MSM and handing it back costs more than the MSM (keygen). */
int vdb_msm_set_scratch_cap(size_t bytes);
int vdb_mem_info(size_t *free_bytes, size_t *total_bytes); /* HBM of the calling thread's device (hipMemGetInfo) */
int vdb_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes); /* asynchronous on the library stream */
int vdb_memset_dev(void *dst_dev, int value, size_t bytes);
int vdb_sync(void);
/* HIP-event timing on the library's own stream (bench.py: torch.cuda.Event does not see it) */
int vdb_timer_start(void);
int vdb_timer_stop(float *ms_out);
/* per-kernel HIP-event timing: between begin and end every kernel launch of the library is bracketed by
 * events on its stream (serialising them); end returns {"kernel": {"ms": total, "launches": n}, ...} */
int vdb_profile_begin(void);
/* the same without serialising: events are recorded on the stream each launch goes to and read in vdb_profile_end, so
 * the kernels run (and overlap) exactly as they do untimed — what bench.py uses inside its timed region */
int vdb_profile_begin_deferred(void);
int vdb_profile_end(char *json_out, size_t cap);

/* ---- field helpers (tests / staging) -------------------------------------------------------- */
/* acc <- acc x + values[i], i = 0 .. n - 1, on the host (Montgomery in and out; needs no GPU): combining the evaluations of a
 * multi-open with powers of a challenge (halo2 poly/kzg/multiopen) */
int vdb_fr_horner(const vdb_fr *values, size_t n, const vdb_fr *x, vdb_fr *acc);
/* out[i] = (64 little-endian bytes wide[64 i ..]) mod r, Montgomery: Fr::random / Fr::from_u512 of halo2curves — how the
 * prover's blinding scalars are drawn from OS entropy (halo2 create_proof: Blind(Scalar::random(rng)), reached from
 * src/scaffold/mod.rs:296).  Device pointers; asynchronous on the library stream. */
int vdb_fr_from_wide_dev(const uint8_t *wide_dev, size_t n, vdb_fr *out_dev);
int vdb_fr_from_canonical(const vdb_fr *in, vdb_fr *out, size_t n);
int vdb_fr_to_canonical(const vdb_fr *in, vdb_fr *out, size_t n);
int vdb_fr_mul(const vdb_fr *a, const vdb_fr *b, vdb_fr *out, size_t n);
int vdb_fr_add(const vdb_fr *a, const vdb_fr *b, vdb_fr *out, size_t n);
int vdb_fr_sub(const vdb_fr *a, const vdb_fr *b, vdb_fr *out, size_t n);
/* replaces halo2 `BatchInvert` over `Assigned::Rational` denominators (poly::batch_invert_assigned);
 * 0 maps to 0 */
int vdb_fr_batch_invert(const vdb_fr *in, vdb_fr *out, size_t n);
/* micro-benchmark: `iters` dependent Montgomery multiplications in each of `threads` GPU threads;
 * reports Fr multiplications per second (kernel time only) */
int vdb_bench_fr_mul(size_t threads, size_t iters, double *mul_per_sec);

/* ---- a2/a3/a18 fixed-point staging: replaces FixedPointChip::{quantization, dequantization} and
 *      FixedPointVectorInstructions::{quantize_vector, dequantize_vector}
 *      (src/gadget/fixed_point.rs:104-136, src/gadget/fixed_point_vec.rs:29-51).  Host side, exact. ---- */
int vdb_fp_quantize(uint32_t precision_bits, const double *x, vdb_fr *out, size_t n);
int vdb_fp_dequantize(uint32_t precision_bits, const vdb_fr *x, double *out, size_t n);

/* ---- b5 witness streams.  Each call emits exactly the cells the Rust gadget pushes into
 *      halo2-base `Context.advice` (stream_out) and `cells_to_lookup` (lookup_out), in order, for
 *      already-assigned quantized inputs; sizes come from the matching *_size call.
 *      metric: 0 euclidean, 1 cosine, 2 manhattan, 3 hamming (DistanceChip, src/gadget/distance.rs:97-195; hamming = one minus the
 *      share of equal elements, its two load_witness cells unconstrained as in the reference, :165-169).
 *      selector_out (optional, 1 flag byte per advice cell): bit 0 marks gate starts — the keygen-side information
 *      from which vdb_layout_plan derives the break points that the reference pins in configs/<name>.json
 *      (src/scaffold/mod.rs:272, 285-287); bit 1 marks cells that hold a data-independent QuantumCell::Constant
 *      of the gate templates (used by vdb_msm_batch_masked_dev).  VDB_ERR_DOMAIN replaces the reference's panics. ---------- */
int vdb_wit_distance_size(int metric, uint32_t precision_bits, uint32_t lookup_bits, size_t n_pairs, size_t dim, uint64_t *cells, uint64_t *lookups);
/* DistanceChip::{euclidean,cosine,manhattan,hamming}_distance for n_pairs independent (a_i, b_i) */
int vdb_wit_distance(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *a, const vdb_fr *b, size_t n_pairs, size_t dim,
                     vdb_fr *stream_out, vdb_fr *lookup_out, uint8_t *selector_out, vdb_fr *result_out);
/* the same with inputs and outputs resident in HBM (the distances circuit of examples/distances.rs / examples/euclid.rs inside the hot
 * path: pipeline.DistancesHotPath); honours vdb_wit_set_window */
int vdb_wit_distance_dev(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *a_dev, const vdb_fr *b_dev, size_t n_pairs, size_t dim,
                         vdb_fr *stream_dev, vdb_fr *lookup_dev, uint8_t *selector_dev, vdb_fr *result_dev);
/* FixedPointInstructions, one call per instance (src/gadget/fixed_point.rs:213-460 — what the chips above are composed of, for circuits
 * that apply an operation to many values): n independent calls op(a_i [, b_i]); instance i's cells are cells/n consecutive cells of the
 * stream (its lookup cells likewise), in the order the gadget pushes them.  op: 0 qadd, 1 qsub, 2 qmul, 3 qdiv, 4 neg, 5 qabs, 6 is_neg,
 * 7 qmin, 8 qsqrt, 9 qlog2, 10 qexp2, 11 qlog, 12 qexp, 13 qpow, 14 bit_xor, 15 cond_neg (b: the flag), 16 signed_div_scale (result:
 * the quotient), 17 qmax, 18 sign (the field elements 1 / -1), 19 clip, 20 qmod (b positive), 21 qsin, 22 qcos, 23 qtan, 24 qsinh,
 * 25 qcosh, 26 qtanh.  b may be null for the unary ones.  VDB_ERR_DOMAIN where the reference panics (a division by zero).  The _dev
 * form honours vdb_wit_set_window. */
int vdb_wit_fp_op_size(int op, uint32_t precision_bits, uint32_t lookup_bits, size_t n, uint64_t *cells, uint64_t *lookups);
int vdb_wit_fp_op(int op, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *a, const vdb_fr *b, size_t n, vdb_fr *stream_out,
                  vdb_fr *lookup_out, uint8_t *selector_out, vdb_fr *result_out);
int vdb_wit_fp_op_dev(int op, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *a_dev, const vdb_fr *b_dev, size_t n, vdb_fr *stream_dev,
                      vdb_fr *lookup_dev, uint8_t *selector_dev, vdb_fr *result_dev);
/* VectorDBChip::nearest_vector (src/gadget/vectordb.rs:122-163): indicator_out n raw 0/1 field bits */
int vdb_wit_nearest_size(int metric, uint32_t precision_bits, uint32_t lookup_bits, size_t n, size_t dim, uint64_t *cells, uint64_t *lookups);
int vdb_wit_nearest(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *query, const vdb_fr *vectors, size_t n, size_t dim,
                    vdb_fr *stream_out, vdb_fr *lookup_out, uint8_t *selector_out, vdb_fr *indicator_out, vdb_fr *result_out);
/* the same on device-resident buffers (query_dev: dim, vectors_dev: n x dim; outputs stay in HBM); honours vdb_wit_set_window:
 * a rank stores the cells of its own columns, every rank computes every value (the n distances, the short minimum chain) */
int vdb_wit_nearest_dev(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *query_dev, const vdb_fr *vectors_dev, size_t n, size_t dim,
                        vdb_fr *stream_dev, vdb_fr *lookup_dev, uint8_t *selector_dev, vdb_fr *indicator_dev, vdb_fr *result_dev);
/* VectorDBChip::kmeans::<K, I> (src/gadget/vectordb.rs:225-362): centroids K x dim, indicators n x K
 * (quantized 1.0 / 0).  zero_cached: Context::load_zero already called earlier in this context. */
int vdb_wit_kmeans_size(int metric, uint32_t precision_bits, uint32_t lookup_bits, size_t n, size_t dim, size_t K, size_t I, int zero_cached,
                        uint64_t *cells, uint64_t *lookups);
int vdb_wit_kmeans(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *vectors, size_t n, size_t dim, size_t K, size_t I,
                   int zero_cached, vdb_fr *stream_out, vdb_fr *lookup_out, uint8_t *selector_out, vdb_fr *centroids_out, vdb_fr *indicators_out);
/* Multi-GPU: restricts what the following *_dev witness calls STORE to the advice cells [adv_lo, adv_hi) and lookup
 * cells [lookup_lo, lookup_hi) (coordinates of the stream / lookup pointers passed to the call): a rank that commits a
 * block of columns only needs those cells; all values (distances, centroids, ...) are still computed everywhere.
 * Pass (0, UINT64_MAX, 0, UINT64_MAX) to restore the full range. */
int vdb_wit_set_window(uint64_t adv_lo, uint64_t adv_hi, uint64_t lookup_lo, uint64_t lookup_hi);
int vdb_wit_kmeans_dev(int metric, uint32_t precision_bits, uint32_t lookup_bits, const vdb_fr *vectors_dev, size_t n, size_t dim, size_t K, size_t I,
                       int zero_cached, vdb_fr *stream_dev, vdb_fr *lookup_dev, uint8_t *selector_dev, vdb_fr *centroids_dev, vdb_fr *indicators_dev);
/* VectorDBChip::merkle_commitment with PoseidonChip<F,3,2>(R_F=8, R_P=57) (src/gadget/vectordb.rs:165-223):
 * the permutation trace cells; no lookups */
int vdb_wit_merkle_size(size_t n, size_t dim, int zero_cached, uint64_t *cells);
int vdb_wit_merkle(const vdb_fr *vectors, size_t n, size_t dim, int zero_cached, vdb_fr *stream_out, uint8_t *selector_out, vdb_fr *root_out);
int vdb_wit_merkle_dev(const vdb_fr *vectors_dev, size_t n, size_t dim, int zero_cached, vdb_fr *stream_dev, uint8_t *selector_dev, vdb_fr *root_dev);

/* ---- b4 stream -> columns: replaces halo2-base GateThreadBuilder::assign_all (break points, keygen)
 *      and assign_threads_in (prover) as driven by RangeCircuitBuilder::prover(builder, break_points)
 *      (src/scaffold/mod.rs:393-396).  Columns are 2^k rows; the cell on a break row is duplicated at
 *      row 0 of the next column.  Lookup cells fill dedicated columns of 2^k - minimum_rows cells. ---- */
int vdb_layout_plan(const uint8_t *selector, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t *break_points_out, uint64_t cap,
                    uint64_t *n_break_points);
int vdb_layout_plan_dev(const uint8_t *selector_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t *break_points_out, uint64_t cap,
                        uint64_t *n_break_points);
/* advice_cols_out: (n_bp + 1) x 2^k; lookup_cols_out: n_lookup_cols x 2^k (may be NULL) */
int vdb_layout_columns(const vdb_fr *stream, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, const vdb_fr *lookup, uint64_t n_lookup,
                       uint32_t k, uint32_t minimum_rows, vdb_fr *advice_cols_out, vdb_fr *lookup_cols_out, uint64_t n_lookup_cols);
/* device variants; blind_dev (optional): n_cols x n_blind scalars written to the last n_blind rows
 * (the prover's blinding rows) */
int vdb_layout_columns_dev(const vdb_fr *stream_dev, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, uint32_t k, vdb_fr *cols_dev,
                           const vdb_fr *blind_dev, uint32_t n_blind);
int vdb_layout_lookup_dev(const vdb_fr *lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, vdb_fr *cols_dev, uint64_t n_cols,
                          const vdb_fr *blind_dev, uint32_t n_blind);
/* Columns without the copy.  A vdb_colsrc describes one column that still lies in a witness stream: rows [0, len) are the
 * contiguous cells src[0 .. len), the last n_blind rows come from `blind` (may be NULL), every other row is zero — exactly
 * what vdb_layout_columns[_range]_dev / vdb_layout_lookup[_range]_dev would have written.  The commitment MSM
 * (vdb_msm_batch_src_dev_begin) and the first pass of lagrange_to_coeff (vdb_lagrange_to_coeff_src_dev, which writes the
 * coefficient columns) read through it, so the 64 B / cell of the layout copy disappear from the prover's step.  The
 * descriptors are data independent: build them once per circuit.  out_dev: (col_hi - col_lo) descriptors in device memory. */
typedef struct {
  const vdb_fr *src;
  uint64_t len;
  const vdb_fr *blind;
} vdb_colsrc;
int vdb_colsrc_build_dev(const vdb_fr *stream_dev, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, uint32_t k, uint64_t col_lo,
                         uint64_t col_hi, const vdb_fr *blind_dev, uint32_t n_blind, vdb_colsrc *out_dev);
int vdb_colsrc_build_lookup_dev(const vdb_fr *lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t col_lo, uint64_t col_hi,
                                const vdb_fr *blind_dev, uint32_t n_blind, vdb_colsrc *out_dev);
/* same, restricted to columns [col_lo, col_hi) (multi-GPU column shards); cols_dev holds just that range; blind_dev is
 * still indexed by absolute column */
int vdb_layout_columns_range_dev(const vdb_fr *stream_dev, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, uint32_t k, uint64_t col_lo,
                                 uint64_t col_hi, vdb_fr *cols_dev, const vdb_fr *blind_dev, uint32_t n_blind);
int vdb_layout_lookup_range_dev(const vdb_fr *lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t col_lo, uint64_t col_hi,
                                vdb_fr *cols_dev, const vdb_fr *blind_dev, uint32_t n_blind);
/* keygen side: column-layout image ((n_bp + 1) x 2^k bytes) of the constant-cell flags (bit 1 of the flag bytes) */
int vdb_layout_const_mask_dev(const uint8_t *flags_dev, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, uint32_t k, uint8_t *mask_dev);
/* out = mask ? in : 0 (keep_const = 1) or mask ? 0 : in (keep_const = 0), elementwise over n field elements */
int vdb_mask_select_dev(const vdb_fr *in_dev, const uint8_t *mask_dev, uint64_t n, int keep_const, vdb_fr *out_dev);

/* ---- b1 SRS: replaces halo2 ParamsKZG::{get_g, g_lagrange} as consumed by commit / commit_lagrange;
 *      the reference obtains the params with gen_srs(k) (src/scaffold/mod.rs:260) ------------------ */
/* Uploads the bases (caller keeps ownership of the host arrays; either may be NULL) and precomputes
 * the fixed-base window tables 2^(c*j) * G_i in HBM. */
int vdb_srs_load(uint32_t k, const vdb_g1 *g, const vdb_g1 *g_lagrange, vdb_srs **out);
/* the same with an explicit Pippenger window (2..14 bits; 0 = the default, 11 bits for k > 8, tuned for witness columns whose
 * scalars are mostly short).  Columns of full-width scalars (Poseidon traces, the product columns of the permutation and
 * lookup arguments, quotient and opening polynomials) commit ~15 % faster with 14 bits: 19 windows instead of 24. */
int vdb_srs_load_window(uint32_t k, const vdb_g1 *g, const vdb_g1 *g_lagrange, uint32_t window_bits, vdb_srs **out);
/* "unsafe" trusted setup with a caller-supplied tau, the construction ParamsKZG::setup performs behind
 * gen_srs(k) (src/scaffold/mod.rs:260-261: "unsafe" message); tau is a Montgomery Fr.  Test/bench SRS. */
int vdb_srs_setup_unsafe(uint32_t k, const vdb_fr *tau, vdb_g1 *g_out, vdb_g1 *g_lagrange_out);
void vdb_srs_free(vdb_srs *srs);
/* out[c] = parts[0][c] + ... + parts[m-1][c] in G1 (host arrays, m x n and n affine points; the identity is (0, 0)): the
 * combine step of a point-sharded MSM — each GPU commits its slice of the rows of every column against the matching slice of
 * the bases, the partial commitments are all-gathered and added.  RCCL has no curve reduction operator. */
int vdb_g1_sum(const vdb_g1 *parts, size_t m, size_t n, vdb_g1 *out);
int vdb_srs_device(const vdb_srs *srs, int *device); /* the device whose HBM holds the handle's tables */
int vdb_srs_info(const vdb_srs *srs, uint32_t *k, uint32_t *window_bits, uint32_t *windows);

/* ---- b2 MSM: replaces halo2 arithmetic::best_multiexp(&[Fr], &[G1Affine]) -> G1 and
 *      ParamsKZG::commit_lagrange / commit (reached from src/scaffold/mod.rs:296, :273).
 *      basis: 0 = monomial (g), 1 = lagrange (g_lagrange).  Result: exact group element as canonical
 *      affine, identity = (0,0).  n <= 2^k scalars use the first n bases. ------------------------- */
int vdb_msm(const vdb_srs *srs, int basis, const vdb_fr *scalars, size_t n, vdb_g1 *out);
int vdb_msm_batch(const vdb_srs *srs, int basis, const vdb_fr *const *cols, size_t n_cols, size_t n, vdb_g1 *out);
/* The same over every device bound by vdb_init_devices (columns are independent, the bases are replicated):
 * vdb_srs_load_all builds one handle per bound device (out[i] belongs to the i-th bound device, n_out = vdb_devices_bound());
 * vdb_msm_batch_multi gives device i the i-th contiguous block of the columns, one host thread per device, results D2H into
 * out[] in column order.  With one bound device they reduce to the calls above. */
int vdb_srs_load_all(uint32_t k, const vdb_g1 *g, const vdb_g1 *g_lagrange, uint32_t window_bits, vdb_srs **out, int n_out);
int vdb_msm_batch_multi(vdb_srs *const *srs, int n_srs, int basis, const vdb_fr *const *cols, size_t n_cols, size_t n, vdb_g1 *out);
/* scalars_dev: contiguous n_cols x n in HBM; out_host: n_cols points */
int vdb_msm_batch_dev(const vdb_srs *srs, int basis, const vdb_fr *scalars_dev, size_t n_cols, size_t n, vdb_g1 *out_host);
/* Deferred form of vdb_msm_batch[_masked]_dev (mask and constant points may both be NULL): _begin queues the whole MSM and
 * returns without waiting; the bucket folding of its last batch — short, latency-bound launches — runs on a second
 * stream, so work queued next (vdb_lagrange_to_coeff_dev / vdb_coeff_to_extended_dev on the same columns: the scalars
 * are no longer read) overlaps it.  _end waits FOR THE MSM ONLY and copies the n_cols commitments out: what was queued on the
 * library's stream after _begin may still be running when it returns (the caller feeds the commitments to its transcript
 * while the transforms run; vdb_sync() or any blocking call joins).  One deferred MSM at a time. */
int vdb_msm_batch_masked_dev_begin(const vdb_srs *srs, int basis, const vdb_fr *scalars_dev, size_t n_cols, size_t n, const uint8_t *skip_mask_dev,
                                   const vdb_g1 *const_points_dev);
/* same for columns described by vdb_colsrc (n rows each, the last n_blind of them blinding rows) */
int vdb_msm_batch_src_dev_begin(const vdb_srs *srs, int basis, const vdb_colsrc *src_dev, size_t n_cols, size_t n, uint32_t n_blind,
                                const uint8_t *skip_mask_dev, const vdb_g1 *const_points_dev);
int vdb_msm_batch_end(vdb_g1 *out_host, size_t n_cols);
/* per column: the number of (scalar, window) entries vdb_msm_batch[_masked]_dev would sort and accumulate for it (non-zero
 * signed digits of the cells not flagged in skip_mask_dev, which may be NULL).  A keygen-time statistic used to balance
 * column shards across GPUs.  counts_out: host, n_cols values. */
int vdb_msm_count_entries_dev(const vdb_srs *srs, const vdb_fr *scalars_dev, size_t n_cols, size_t n, const uint8_t *skip_mask_dev,
                              uint64_t *counts_out);
/* Prover-side commit with the constant cells factored out: cells flagged in skip_mask_dev (n_cols x n bytes, from
 * vdb_layout_const_mask_dev) are skipped and const_points_dev[col] — the keygen-time MSM of exactly those cells
 * (vdb_mask_select_dev(keep_const = 1) + vdb_msm_batch_dev) — is added.  Same group element as vdb_msm_batch_dev. */
int vdb_msm_batch_masked_dev(const vdb_srs *srs, int basis, const vdb_fr *scalars_dev, size_t n_cols, size_t n, const uint8_t *skip_mask_dev,
                             const vdb_g1 *const_points_dev, vdb_g1 *out_host);

/* ---- b3 NTT: replaces halo2 arithmetic::best_fft / EvaluationDomain::{lagrange_to_coeff,
 *      coeff_to_extended} (reached from src/scaffold/mod.rs:296) ------------------------------ */
#define VDB_NTT_INVERSE_SCALE 1 /* multiply the result by n^{-1} (EvaluationDomain::ifft) */
/* In place on each column; natural order in, natural order out; X[i] = sum_j a[j] omega^(ij). */
int vdb_ntt_batch(vdb_fr *const *cols, size_t n_cols, uint32_t log_n, const vdb_fr *omega, int flags);
/* the same with the columns cut into one contiguous block per bound device (vdb_init_devices), one host thread each */
int vdb_ntt_batch_multi(vdb_fr *const *cols, size_t n_cols, uint32_t log_n, const vdb_fr *omega, int flags);
/* contiguous device buffer of n_cols columns of 2^log_n elements */
int vdb_ntt_batch_dev(vdb_fr *cols_dev, size_t n_cols, uint32_t log_n, const vdb_fr *omega, int flags);
/* lagrange_to_coeff for the 2^k domain (omega = ROOT_OF_UNITY^(2^(28-k))) */
int vdb_lagrange_to_coeff(vdb_fr *const *cols, size_t n_cols, uint32_t k);
int vdb_lagrange_to_coeff_dev(vdb_fr *cols_dev, size_t n_cols, uint32_t k);
/* lagrange_to_coeff of columns described by vdb_colsrc: reads the witness stream, writes n_cols coefficient columns of
 * 2^k elements to coeff_dev (k > 10) */
int vdb_lagrange_to_coeff_src_dev(const vdb_colsrc *src_dev, vdb_fr *coeff_dev, size_t n_cols, uint32_t k, uint32_t n_blind);
/* Batched grand product of the permutation / lookup arguments (the first prover-round primitive after
 * commit + NTT): per column of n entries z[0] = 1, z[i + 1] = z[i] * num[i] / den[i] (i < n - 1).  A zero denominator
 * inverts to zero like halo2's batch_invert, i.e. z is zero from that row on.  One field inversion per column.
 * num_dev, den_dev, z_dev: n_cols x n, column-major contiguous, device memory. */
int vdb_grand_product_dev(const vdb_fr *num_dev, const vdb_fr *den_dev, size_t n_cols, size_t n, vdb_fr *z_dev);
/* out[c] = sum_i coeff[c][i] * x^i for n_cols coefficient-form polynomials of n coefficients (halo2 eval_polynomial, the
 * opening evaluations of the advice polynomials).  coeff_dev: device; x: one field element on the host; out_host: host. */
int vdb_eval_polys_dev(const vdb_fr *coeff_dev, size_t n_cols, size_t n, const vdb_fr *x, vdb_fr *out_host);
/* the same queued only: the n_cols values go to out_dev, nothing is waited for (the caller reads them with vdb_memcpy_d2h, which is
 * ordered behind the kernel, and may queue the next evaluation before it turns to the host side of the previous one) */
int vdb_eval_polys_dev_out(const vdb_fr *coeff_dev, size_t n_cols, size_t n, const vdb_fr *x, vdb_fr *out_dev);
/* Lookup argument, permuted columns (halo2 plonk/lookup/prover.rs permute_expression_pair) for range-table lookups: per
 * input column, over rows [0, usable_rows): permuted_input = the input values in ascending canonical order;
 * permuted_table = at the first row of every run of equal values that value, elsewhere the table's left-over values in
 * ascending order assigned from the last repeated row backwards; rows >= usable_rows are zeroed (the caller adds the
 * blinding rows).  All values must be canonical integers below 2^max_bits (max_bits <= 20: a counting sort) and every
 * input value must occur in the table, otherwise VDB_ERR_DOMAIN (the reference panics).
 * input_dev, permuted_*_dev: n_cols x n; table_dev: one column of n. */
int vdb_lookup_permute_dev(const vdb_fr *input_dev, const vdb_fr *table_dev, size_t n_cols, size_t n, size_t usable_rows, uint32_t max_bits,
                           vdb_fr *permuted_input_dev, vdb_fr *permuted_table_dev);
/* Lookup argument, running product (halo2 plonk/lookup/prover.rs commit_product): per input column z[0] = 1,
 * z[i + 1] = z[i] (A[i] + beta)(S[i] + gamma) / ((A'[i] + beta)(S'[i] + gamma)) for i < usable_rows, rows above usable_rows zero
 * (the caller blinds them).  A, A', S': n_cols x n; S (the table): one column of n.  z[usable_rows] is one exactly when
 * (A', S') is a valid permutation pair of (A, S). */
int vdb_lookup_product_dev(const vdb_fr *input_dev, const vdb_fr *table_dev, const vdb_fr *perm_input_dev, const vdb_fr *perm_table_dev, size_t n_cols, size_t n,
                           size_t usable_rows, const vdb_fr *beta, const vdb_fr *gamma, vdb_fr *z_dev);
/* Permutation argument.  halo2curves' Fr::DELTA = 7^(2^28). */
int vdb_fr_delta(vdb_fr *out);
/* keygen side (halo2 plonk/permutation/keygen.rs build_pk: the sigma columns in Lagrange form): sigma[c][row] =
 * delta^c' omega^row' for the cell (c', row') that the copy-constraint permutation sends (c, row) to;
 * mapping_dev: n_cols x 2^k words c' << 32 | row' (the identity where a cell is unconstrained). */
int vdb_permutation_sigma_dev(const uint64_t *mapping_dev, size_t n_cols, uint32_t k, const vdb_fr *delta, vdb_fr *sigma_dev);
/* The mapping kept with the proving key in 32 bits per cell (c' << k | row'; VDB_ERR_ARG when column and row do not fit), and the
 * sigma columns of a block of columns in Lagrange form straight from it: halo2's ProvingKey holds the permutation polynomials in
 * Lagrange, coefficient and coset form (plonk/permutation.rs ProvingKey: permutations, polys, cosets); here the coefficient form
 * is held, the cosets are made per block in the quotient, and the Lagrange form the product round reads costs one product per
 * cell from these 4 bytes instead of a forward transform per column.  packed_block_dev: the block's n_block_cols x 2^k words. */
int vdb_permutation_mapping_pack_dev(const uint64_t *mapping_dev, size_t n_cols, uint32_t k, uint32_t *packed_dev);
int vdb_permutation_sigma_packed_dev(const uint32_t *packed_block_dev, size_t n_block_cols, size_t n_cols_total, uint32_t k, const vdb_fr *delta,
                                     vdb_fr *sigma_block_dev);
/* prover side (plonk/permutation/prover.rs commit): the columns are taken in chunks of chunk_len (= constraint degree - 2);
 * per chunk one product column z with z[i + 1] = z[i] prod_c (v_c[i] + beta delta^c omega^i + gamma) / (v_c[i] + beta
 * sigma_c[i] + gamma) over the chunk's columns c, i < usable_rows; chunk j starts from the value chunk j - 1 ended on
 * (z_0[0] = 1); rows above usable_rows are zero (the caller blinds them).  The last chunk's z[usable_rows] is one
 * exactly when every cell equals the cell the permutation sends it to (up to the soundness error in beta, gamma).
 * cols_dev, sigma_dev: n_cols x 2^k; z_dev: ceil(n_cols / chunk_len) x 2^k. */
int vdb_permutation_product_dev(const vdb_fr *cols_dev, const vdb_fr *sigma_dev, size_t n_cols, uint32_t k, size_t usable_rows, size_t chunk_len,
                                const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *delta, vdb_fr *z_dev);
/* Permutation and lookup parts of the quotient numerator on the extended coset of 2^(k+ext_k) points (halo2
 * plonk/evaluation.rs evaluate_h), folded into acc_dev term by term as acc = acc * y + term, after the gates
 * (vdb_gate_eval_dev).  All *_ext_dev arrays are vdb_coeff_to_extended_dev images: the advice, sigma and product columns;
 * l0 (Lagrange basis of row 0), l_last (row usable_rows), l_active = 1 - (l_last + sum of the rows above usable_rows).
 * Permutation terms: l0 (1 - z_0); l_last (z_last^2 - z_last); l0 (z_i - z_{i-1}(w^-(n-usable_rows) X)) for i >= 1; per chunk
 * l_active (z_i(wX) prod (v + beta sigma + gamma) - z_i(X) prod (v + delta^c beta X + gamma)).
 * Lookup terms per input column: l0 (1 - z); l_last (z^2 - z); l_active (z(wX)(A' + beta)(S' + gamma) - z (A + beta)(S + gamma));
 * l0 (A' - S'); l_active (A' - S')(A' - A'(w^-1 X)). */
int vdb_permutation_eval_dev(const vdb_fr *adv_ext_dev, const vdb_fr *sigma_ext_dev, const vdb_fr *z_ext_dev, size_t n_cols, size_t chunk_len, uint32_t k,
                             uint32_t ext_k, size_t usable_rows, const vdb_fr *l0_ext_dev, const vdb_fr *l_last_ext_dev, const vdb_fr *l_active_ext_dev,
                             const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *delta, const vdb_fr *y, vdb_fr *acc_dev);
/* the same for the sets [set_lo, set_hi) only (the terms that involve nothing but the product columns come with set_lo == 0):
 * sigma_ext_block_dev holds the cosets of columns set_lo * chunk_len onwards, so that the sigma cosets can be produced block
 * by block from their coefficients instead of being kept resident; calls must be made in ascending order of the sets. */
int vdb_permutation_eval_range_dev(const vdb_fr *adv_ext_dev, const vdb_fr *sigma_ext_block_dev, const vdb_fr *z_ext_dev, size_t n_cols, size_t chunk_len,
                                   uint32_t k, uint32_t ext_k, size_t usable_rows, const vdb_fr *l0_ext_dev, const vdb_fr *l_last_ext_dev,
                                   const vdb_fr *l_active_ext_dev, const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *delta, const vdb_fr *y, vdb_fr *acc_dev,
                                   size_t set_lo, size_t set_hi);
/* Streaming forms for circuits whose columns do not all fit HBM at once (the rounds then work on blocks of columns):
 * vdb_permutation_product_range_dev: the running products of the chunks of ONE block of columns (cols_block / sigma_block hold
 * columns col0 .. col0 + n_block_cols, col0 a multiple of chunk_len), each starting from one; vdb_permutation_chain_dev then
 * puts all chunks of all blocks on one chain (z_c starts where z_{c-1} ended), as vdb_permutation_product_dev does in one go.
 * vdb_permutation_eval_parts_dev: the permutation part of the quotient numerator with block buffers — adv_ext_block holds the
 * cosets of columns adv_col0 .., z_ext_block those of product sets z_set0 .., z_first / z_last the cosets of the first and last
 * set; `head` bit 0: fold in l0 (1 - z_0) and l_last (z_last^2 - z_last); bit 1: sigma_ext_block holds the cosets of beta sigma(X)
 * (vdb_coeff_to_extended_scaled_dev) and the kernel skips its product by beta; [chain_lo, chain_hi): fold in l0 (z_i - z_{i-1}(..)) for these
 * sets (reads sets chain_lo - 1 .. chain_hi - 1); [set_lo, set_hi): fold in the product terms of these sets (reads their columns,
 * sigma_ext_block starting at column set_lo * chunk_len).  Terms must be folded in the order head, chain 1 .. n_sets - 1, products
 * 0 .. n_sets - 1 to agree with the resident form. */
int vdb_permutation_product_range_dev(const vdb_fr *cols_block_dev, const vdb_fr *sigma_block_dev, size_t n_block_cols, size_t col0, uint32_t k,
                                      size_t usable_rows, size_t chunk_len, const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *delta, vdb_fr *z_block_dev);
int vdb_permutation_chain_dev(vdb_fr *z_dev, size_t n_chunks, uint32_t k, size_t usable_rows);
int vdb_permutation_eval_parts_dev(const vdb_fr *adv_ext_block_dev, size_t adv_col0, const vdb_fr *sigma_ext_block_dev, const vdb_fr *z_ext_block_dev, size_t z_set0,
                                   const vdb_fr *z_first_ext_dev, const vdb_fr *z_last_ext_dev, size_t n_cols, size_t chunk_len, uint32_t k, uint32_t ext_k,
                                   size_t usable_rows, const vdb_fr *l0_ext_dev, const vdb_fr *l_last_ext_dev, const vdb_fr *l_active_ext_dev, const vdb_fr *beta,
                                   const vdb_fr *gamma, const vdb_fr *delta, const vdb_fr *y, vdb_fr *acc_dev, int head, size_t chain_lo, size_t chain_hi,
                                   size_t set_lo, size_t set_hi);
int vdb_lookup_eval_dev(const vdb_fr *input_ext_dev, const vdb_fr *table_ext_dev, const vdb_fr *perm_input_ext_dev, const vdb_fr *perm_table_ext_dev,
                        const vdb_fr *z_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr *l0_ext_dev, const vdb_fr *l_last_ext_dev,
                        const vdb_fr *l_active_ext_dev, const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *y, vdb_fr *acc_dev);
/* rows [from_row, n) of each of the n_cols columns <- src_dev (n_cols x (n - from_row) values): the blinding rows of the
 * columns the prover derives (permuted columns, product columns) */
int vdb_fill_rows_dev(vdb_fr *cols_dev, size_t n_cols, size_t n, size_t from_row, const vdb_fr *src_dev);
/* Opening proofs (halo2 poly/kzg/multiopen: both GWC and SHPLONK are built from these two steps and a commit).
 * vdb_poly_lincomb_dev: acc <- Horner over the n_cols polynomials, acc = acc * v + p_c, coefficient by coefficient (acc_dev:
 * n coefficients, read and written; zero it first).
 * vdb_kate_div_dev (arithmetic::kate_division): per polynomial of n coefficients q(X) = (p(X) - p(x)) / (X - x), n coefficients
 * written (the top one zero); rem_host (may be NULL) receives p(x).  Dividing by a product of linear factors — the vanishing
 * polynomial of a SHPLONK rotation set — is this call once per point. */
/* acc <- acc + a * x over n coefficients (SHPLONK's linearisation polynomial is a sum of polynomials with unrelated scalars) */
int vdb_poly_axpy_dev(vdb_fr *acc_dev, const vdb_fr *a, const vdb_fr *x_dev, size_t n);
/* x <- a * x over n elements.  The sharded prover rounds use it for what a rank does not hold: a fold acc <- acc y + term that
 * skips g terms of other ranks is acc <- y^g acc, and a rank's running products start from the product of all earlier ranks'. */
int vdb_poly_scale_dev(vdb_fr *x_dev, const vdb_fr *a, size_t n);
int vdb_poly_lincomb_dev(const vdb_fr *polys_dev, size_t n_cols, size_t n, const vdb_fr *v, vdb_fr *acc_dev);
int vdb_kate_div_dev(const vdb_fr *coeff_dev, size_t n_cols, size_t n, const vdb_fr *x, vdb_fr *quot_dev, vdb_fr *rem_host);
/* coeff_to_extended: zeta-coset scaling [1, ZETA, ZETA^2] cyclic, zero-extend to 2^(k+ext_k), forward NTT */
int vdb_coeff_to_extended(const vdb_fr *const *coeff_cols, vdb_fr *const *ext_cols, size_t n_cols, uint32_t k, uint32_t ext_k);
int vdb_coeff_to_extended_dev(const vdb_fr *coeff_dev, vdb_fr *ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k);
/* the coset image of scale * p(X): the scalar rides on the coset factors of the transform's first pass (a third of a product per
 * coefficient) — how the quotient takes beta sigma(X) instead of multiplying every point of every sigma coset by beta */
int vdb_coeff_to_extended_scaled_dev(const vdb_fr *coeff_dev, vdb_fr *ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr *scale);
/* EvaluationDomain::extended_to_coeff without the final truncation (the way back for h(X)): in place, per
 * column of 2^(k+ext_k) evaluations on the extended coset -> the 2^(k+ext_k) coefficients. */
int vdb_extended_to_coeff_dev(vdb_fr *ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k);
/* Gate part of the quotient numerator for halo2-base's vertical gate q * (a + b * c - d) with a, b, c, d in
 * four consecutive rows of one advice column: acc[j] <- Horner over the n_cols columns of (acc * y + q_c[j] * gate_c[j]) on
 * the extended coset of 2^(k+ext_k) points.  adv_ext_dev, sel_ext_dev: n_cols x 2^(k+ext_k) (vdb_coeff_to_extended_dev of
 * the advice and selector polynomials); acc_dev: 2^(k+ext_k), read and written (zero it before the first call). */
int vdb_gate_eval_dev(const vdb_fr *adv_ext_dev, const vdb_fr *sel_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr *y, vdb_fr *acc_dev);
/* The same on a sub-coset: the gate has degree 3, so its share of the quotient, (sum of the gate terms) / (X^n - 1), has degree
 * below 2 n and is determined by the coset of 2^(k+ext_k) points that lies inside the 2^(k+adv_ext_k) points the advice cosets were
 * made for (every 2^(adv_ext_k-ext_k)-th point: halo2's extended domain is sized for the highest-degree term, here the
 * permutation's, and evaluates every term on all of it).  adv_ext_dev: n_cols x 2^(k+adv_ext_k); sel_ext_dev: n_cols x
 * 2^(k+ext_k) (vdb_coeff_to_extended_dev with ext_k: half the transform); acc_dev: 2^(k+ext_k).  The caller divides by the
 * vanishing polynomial on the small coset, returns to coefficients there and adds them to the quotient's. */
int vdb_gate_eval_sub_dev(const vdb_fr *adv_ext_dev, uint32_t adv_ext_k, const vdb_fr *sel_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k,
                          const vdb_fr *y, vdb_fr *acc_dev);
/* EvaluationDomain::divide_by_vanishing_poly: h[j] /= (X^n - 1) at the j-th point of the extended coset, in place. */
int vdb_divide_by_vanishing_dev(vdb_fr *h_ext_dev, uint32_t k, uint32_t ext_k);
/* ---- The extended domain coset by coset (round 3).  halo2 evaluates the quotient's numerator on all 2^(k+2) points of the extended
 *      domain (EvaluationDomain::coeff_to_extended, reached from src/scaffold/mod.rs:296) although the quotient of the reference's
 *      circuits has degree below 3 n (constraint degree 4: three pieces) and the gates' share degree below 2 n: three of the four
 *      cosets g_t H of the 2^k-th roots of unity H determine it, two the gates' share.  "Slot" t of a polynomial is its image on
 *      g_t H, g_t = zeta w_{4n}^(bitrev2(t)): slot t, row r is point 4 r + bitrev2(t) of vdb_coeff_to_extended_dev's output, slots
 *      0 and 1 together the coset of 2 n points.  Arrays are [column][slot][row]; a rotation by one row is one step inside a slot.
 *      The quotient that comes out is the same polynomial, coefficient for coefficient.
 * vdb_coeff_to_cosets_dev: n_slots (1..4) cosets of each of n_cols polynomials of 2^k coefficients, times *scale_or_null if given
 *      (the scalar rides on the transform's input factors); one 2^k-point transform per slot instead of a 2^(k+2)-point one.
 * vdb_cosets_to_coeff_dev: the way back for ONE polynomial and the division by the vanishing polynomial in one: cosets_dev
 *      ([n_slots][2^k], overwritten) holds numerator values; coeff_dev receives the n_slots pieces of 2^k coefficients of
 *      numerator / (X^n - 1), which the caller asserts to have degree below n_slots 2^k (residues modulo X^n - g_t^n, then the
 *      Vandermonde system in g_t^n).
 * vdb_gate_eval_cosets_dev / vdb_lookup_eval_cosets_dev / vdb_permutation_eval_parts_cosets_dev: vdb_gate_eval_sub_dev /
 *      vdb_lookup_eval_dev / vdb_permutation_eval_parts_dev on such arrays — n_slots where those take ext_k; the advice cosets of the
 *      gate call may hold more slots per column (adv_slots) than the selectors' and the accumulator's n_slots. */
int vdb_coeff_to_cosets_dev(const vdb_fr *coeff_dev, vdb_fr *cosets_dev, size_t n_cols, uint32_t k, uint32_t n_slots, const vdb_fr *scale_or_null);
int vdb_cosets_to_coeff_dev(vdb_fr *cosets_dev, vdb_fr *coeff_dev, uint32_t k, uint32_t n_slots);
int vdb_gate_eval_cosets_dev(const vdb_fr *adv_cosets_dev, uint32_t adv_slots, const vdb_fr *sel_cosets_dev, size_t n_cols, uint32_t k, uint32_t n_slots,
                             const vdb_fr *y, vdb_fr *acc_dev);
int vdb_lookup_eval_cosets_dev(const vdb_fr *input_dev, const vdb_fr *table_dev, const vdb_fr *perm_input_dev, const vdb_fr *perm_table_dev, const vdb_fr *z_dev,
                               size_t n_cols, uint32_t k, uint32_t n_slots, const vdb_fr *l0_dev, const vdb_fr *l_last_dev, const vdb_fr *l_active_dev,
                               const vdb_fr *beta, const vdb_fr *gamma, const vdb_fr *y, vdb_fr *acc_dev);
int vdb_permutation_eval_parts_cosets_dev(const vdb_fr *adv_block_dev, size_t adv_col0, const vdb_fr *sigma_block_dev, const vdb_fr *z_block_dev, size_t z_set0,
                                          const vdb_fr *z_first_dev, const vdb_fr *z_last_dev, size_t n_cols, size_t chunk_len, uint32_t k, uint32_t n_slots,
                                          size_t usable_rows, const vdb_fr *l0_dev, const vdb_fr *l_last_dev, const vdb_fr *l_active_dev, const vdb_fr *beta,
                                          const vdb_fr *gamma, const vdb_fr *delta, const vdb_fr *y, vdb_fr *acc_dev, int head, size_t chain_lo, size_t chain_hi,
                                          size_t set_lo, size_t set_hi);
/* column-layout image of the gate selectors as field elements (keygen side; flags as for vdb_layout_const_mask_dev, bit 0):
 * q_dev: (n_bp + 1) x 2^k, one where a gate starts. */
int vdb_layout_selectors_dev(const uint8_t *flags_dev, uint64_t n_cells, const uint64_t *break_points, uint64_t n_bp, uint32_t k, vdb_fr *q_dev);
int vdb_fr_root_of_unity(uint32_t k, vdb_fr *out);

/* ---- f2 Fiat–Shamir transcript + proof byte stream: stands in for snark_verifier's PoseidonTranscript<NativeLoader, _>
 *      (src/scaffold/mod.rs:309-310) and the proof writer behind gen_snark_shplonk (src/scaffold/mod.rs:296).  Host code.
 *      Poseidon sponge of width t (the reference's transcript: t = 5, r_f = 8, r_p = 60 [UPSTREAM-RECALL]); values are
 *      buffered and absorbed at the next squeeze, which returns state[1]; `write_*` = `common_*` + append to the proof
 *      (points: 32 bytes, x little-endian, bit 6 of the last byte = y odd; scalars: 32 bytes little-endian canonical). */
typedef struct vdb_transcript vdb_transcript;
int vdb_transcript_new(uint32_t t, uint32_t r_f, uint32_t r_p, vdb_transcript **out);
/* where a compressed point carries "y is odd": bit 6 (default; halo2curves >= 0.4 as recalled) or bit 7 (halo2curves 0.3.x) of the
 * last byte — parity unpinned either way, the reference holds no proof bytes */
int vdb_transcript_set_sign_bit(vdb_transcript *tr, uint32_t bit);
void vdb_transcript_free(vdb_transcript *tr);
int vdb_transcript_common_scalar(vdb_transcript *tr, const vdb_fr *s);
int vdb_transcript_common_point(vdb_transcript *tr, const vdb_g1 *p);
int vdb_transcript_write_scalar(vdb_transcript *tr, const vdb_fr *s);
int vdb_transcript_write_point(vdb_transcript *tr, const vdb_g1 *p);
/* the same over arrays (one call per round instead of one per column) */
int vdb_transcript_write_points(vdb_transcript *tr, const vdb_g1 *p, size_t n);
int vdb_transcript_write_scalars(vdb_transcript *tr, const vdb_fr *s, size_t n);
int vdb_transcript_common_points(vdb_transcript *tr, const vdb_g1 *p, size_t n);
/* the public inputs of a proof: halo2's create_proof absorbs every instance value with common_scalar before anything else */
int vdb_transcript_common_scalars(vdb_transcript *tr, const vdb_fr *s, size_t n);
int vdb_transcript_squeeze(vdb_transcript *tr, vdb_fr *out);
/* absorbs the complete RATE-sized chunks written so far (same state as leaving them to the next squeeze): lets a caller run the
 * sponge's host work beside device work that is already queued */
int vdb_transcript_flush(vdb_transcript *tr);
int vdb_transcript_proof_len(const vdb_transcript *tr, size_t *len);
int vdb_transcript_proof_bytes(const vdb_transcript *tr, uint8_t *out, size_t cap);

/* ---- keygen: the permutation of the copy constraints, on the device.  Replaces the cycle construction of halo2's
 *      keygen_vk / keygen_pk (plonk/permutation/keygen.rs Assembly, reached from src/scaffold/mod.rs:273) for circuits of
 *      10^8 - 10^9 cells.  parent_dev[i], i < n_cells: the stream cell that advice cell i copies (an earlier cell, or i), or
 *      n_cells + r: row r of the constants' fixed column (QuantumCell::Constant, assert_is_const); OVERWRITTEN with the roots.
 *      break_points (host): rows per advice column as vdb_layout_plan gives them; lookup_src_dev[j]: the advice cell lookup cell
 *      j copies (lookup cell j sits at row j % lookup_rows of column n_adv + j / lookup_rows).  n_cols = advice + lookup
 *      columns; the fixed column is column n_cols, the INSTANCE column is column n_cols + 1: instance_cells_dev[i] is the
 *      stream cell the circuit makes public i-th (the closure's make_public vector, src/scaffold/mod.rs:378-400:
 *      RangeWithInstanceCircuitBuilder ties assigned_instances[i] to row i of its one instance column), n_instances <= 2^k.
 *      mapping_dev: (n_cols + 2) x 2^k words col << 32 | row, the input of vdb_permutation_sigma_dev.
 *      VDB_ERR_ARG (checked on the device, nothing is read out of bounds): an instance cell or a lookup source < 0 or >= n_cells,
 *      a copy map that holds a cycle. -------------------------------------------------------------------------------------- */
int vdb_permutation_mapping_dev(int64_t *parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t *break_points, uint64_t n_bp, uint32_t k,
                                const int64_t *lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                const int64_t *instance_cells_dev, uint64_t n_instances, uint64_t *mapping_dev);
/* The same with device memory lent for the sort records (4 x 8 B per grid position of a copy class + the sort's histograms): a keygen
 * passes the buffer its sigma columns and selectors fill afterwards; null or too small: the call allocates its own, as above. */
int vdb_permutation_mapping_ws_dev(int64_t *parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t *break_points, uint64_t n_bp, uint32_t k,
                                   const int64_t *lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                   const int64_t *instance_cells_dev, uint64_t n_instances, uint64_t *mapping_dev, void *work_dev, size_t work_bytes);
/* The constraint map itself built on the device.  halo2-base records one equality per `Existing` / `Constant` cell while the closure
 * runs (reached from src/scaffold/mod.rs:378-400); the gadgets' structure is data independent and repeats, so a caller that knows a
 * block's template (one distance, one assignment, one filter, one division: halo2_vectordb_amd/circuit_sym.py) places it at all its
 * stream offsets with one call instead of walking 10^9 cells on the host.  All pointers are device pointers.
 *   vdb_copymap_init_dev    copy_of[i] = i, const_idx[i] = -1, flags[i] = 0, lookup_src[j] = -1
 *   vdb_copymap_place_dev   m instances of a block of n_blk cells at stream offsets bases[m]: block cell c with code blk_src[c] >= 0
 *                           copies block cell blk_src[c] of the same instance, <= -10 copies the instance's external input number
 *                           -10 - blk_src[c] (ext[m][n_ext]: stream cells), anything else copies nothing; const_idx <- blk_cid[c]
 *                           (index of the fixed-column value, -1 none); flags <- blk_flags[c] (bit 0 gate start, bit 1 the tie is an
 *                           assert_is_const); the block's n_blk_lk lookup cells at lk_bases[m] get their source cells likewise
 *   vdb_copymap_finish_dev  parent_dev (may be NULL) <- the input of vdb_permutation_mapping_dev; counts cells tied to a constant that
 *                           also copy another cell (must be 0) and lookup cells without a source (must be 0) */
int vdb_copymap_init_dev(uint64_t n_cells, uint64_t n_lookup, int64_t *copy_of_dev, int64_t *const_idx_dev, uint8_t *flags_dev, int64_t *lookup_src_dev);
int vdb_copymap_place_dev(const int64_t *blk_src_dev, const int64_t *blk_cid_dev, const uint8_t *blk_flags_dev, uint64_t n_blk, const int64_t *blk_lk_dev,
                          uint64_t n_blk_lk, const int64_t *bases_dev, const int64_t *lk_bases_dev, const int64_t *ext_dev, uint64_t m, uint64_t n_ext,
                          uint64_t n_cells, uint64_t n_lookup, int64_t *copy_of_dev, int64_t *const_idx_dev, uint8_t *flags_dev, int64_t *lookup_src_dev);
int vdb_copymap_finish_dev(const int64_t *copy_of_dev, const int64_t *const_idx_dev, uint64_t n_cells, const int64_t *lookup_src_dev, uint64_t n_lookup,
                           int64_t *parent_dev, uint64_t *tied_not_root, uint64_t *lookups_without_source);
/* out_dev[i] = src_dev[idx_dev[i]]: the public cells read out of the witness stream in instance order (what
 * circuit.instances() returns, src/scaffold/mod.rs:265) without leaving HBM */
int vdb_gather_fr_dev(const vdb_fr *src_dev, const int64_t *idx_dev, size_t n, vdb_fr *out_dev);

/* ---- Mock stage: replaces MockProver::run(k, &circuit, instances).assert_satisfied() of the reference's Mock arm
 *      (src/scaffold/mod.rs:263-266): every gate row a + b c = d, every lookup cell against the range table, every copy
 *      constraint and every constant, checked on the witness where it lies in HBM (flat streams; device pointers).
 *      flags_dev: the flag byte per advice cell of a keygen-style run (bit 0 gate start, bit 1 constant cell);
 *      copy_of_dev[i] = the cell that cell i copies (i itself or negative: none); lookup_src_dev[j] = the advice cell
 *      lookup cell j copies; const_stream_dev: a stream of the same circuit whose constant cells hold the fixed values;
 *      const_idx_dev[i] = r >= 0: cell i is tied to entry r of const_table_dev (n_consts field elements) — Constant cells
 *      and assert_is_const alike.  Any of these may be null (that check is skipped; index and table go together).
 *      Counts and the first offending index per kind. ------ */
typedef struct {
  uint64_t gate_rows_violated, first_gate_row;
  uint64_t lookup_cells_out_of_table, first_lookup_cell;
  uint64_t copies_unequal, first_copy;
  uint64_t lookup_copies_unequal, first_lookup_copy;
  uint64_t constants_changed, first_constant;
  uint64_t instances_unequal, first_instance; /* filled by vdb_mock_check_instances_dev only */
} vdb_mock_report;
int vdb_mock_check_dev(const vdb_fr *stream_dev, uint64_t n_cells, const uint8_t *flags_dev, const vdb_fr *lookup_dev, uint64_t n_lookup,
                       uint32_t lookup_bits, const int64_t *copy_of_dev, const int64_t *lookup_src_dev, const vdb_fr *const_stream_dev,
                       const int64_t *const_idx_dev, const vdb_fr *const_table_dev, uint64_t n_consts, vdb_mock_report *out);
/* The `instances` argument of MockProver::run: stream cell instance_cells_dev[i] must hold instances_dev[i] (the copy between
 * the cell and row i of the instance column).  Sets out->instances_unequal / first_instance and leaves the other fields alone. */
int vdb_mock_check_instances_dev(const vdb_fr *stream_dev, uint64_t n_cells, const int64_t *instance_cells_dev, const vdb_fr *instances_dev,
                                 uint64_t n_instances, vdb_mock_report *out);

/* ---- b6 Poseidon: replaces poseidon::PoseidonChip<F,3,2> value semantics (T=3, RATE=2, R_F=8,
 *      R_P=57 as examples/merkle.rs:15-18; call sites src/gadget/vectordb.rs:180-182, 213-215) --- */
int vdb_poseidon_hash_many(const vdb_fr *inputs, size_t n_msgs, size_t msg_len, vdb_fr *digests);
int vdb_poseidon_merkle_root(const vdb_fr *vectors, size_t n, size_t dim, vdb_fr *root);
int vdb_poseidon_permute(vdb_fr *states /* n x 3 */, size_t n);

#ifdef __cplusplus
}
#endif
#endif
