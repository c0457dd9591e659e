/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * Poseidon (x^5, BN254 Fr) restatement: Grain-LFSR parameter generation (Poseidon paper,
 * generate_parameters_grain.sage), the PSE `poseidon` crate's optimized constants / sparse-MDS
 * factorisation, and the sponge used by the reference's `PoseidonChip<F,T,RATE>` call sites
 * (/root/reference/src/gadget/vectordb.rs:180-182, 213-215; examples/merkle.rs:15-18).
 * The crate itself is NOT in /root/reference (Cargo.toml:24, git branch community-edition).
 * Pinned by the public circomlib/Poseidon-paper known answers (tests/test_oracle_poseidon.py);
 * sponge framing (capacity 2^64, extra padding permutation) is [UPSTREAM-RECALL] => parity unpinned.
 */
#ifndef ORACLE_POSEIDON_H
#define ORACLE_POSEIDON_H
#include "bn254.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PSD_MAX_T 5
#define PSD_MAX_ROUNDS 80

typedef struct {
  int t, rate, r_f, r_p;
  fr_t rc[PSD_MAX_ROUNDS][PSD_MAX_T];           /* unoptimized round constants */
  fr_t mds[PSD_MAX_T][PSD_MAX_T];               /* Cauchy MDS */
  /* optimized form (PSE poseidon Spec) */
  fr_t start[PSD_MAX_ROUNDS][PSD_MAX_T];        /* r_f/2 + 1 rows */
  fr_t partial[PSD_MAX_ROUNDS];                 /* r_p */
  fr_t end[PSD_MAX_ROUNDS][PSD_MAX_T];          /* r_f/2 - 1 rows */
  fr_t pre_sparse_mds[PSD_MAX_T][PSD_MAX_T];
  fr_t sparse_row[PSD_MAX_ROUNDS][PSD_MAX_T];   /* per partial round: first row */
  fr_t sparse_col_hat[PSD_MAX_ROUNDS][PSD_MAX_T]; /* per partial round: t-1 entries */
} psd_spec;

/* cached spec for (t, r_f, r_p) */
const psd_spec *psd_get_spec(int t, int r_f, int r_p);

/* textbook permutation (unoptimized constants) — used to validate the optimized schedule */
void psd_permute_naive(const psd_spec *s, fr_t *state);
/* optimized permutation with absorb (`inputs` may be fewer than rate) exactly as the chip does */
void psd_permute_absorb(const psd_spec *s, fr_t *state, const fr_t *inputs, int n_inputs);
/* sponge hash as PoseidonChip::{clear,update,squeeze}: state [2^64,0,..], returns state[1] */
void psd_hash(const psd_spec *s, fr_t *out, const fr_t *msg, size_t len);

/* flat exports for ctypes */
void orc_poseidon_permute_naive(int t, int r_f, int r_p, fr_t *state);
void orc_poseidon_permute_opt(int t, int r_f, int r_p, fr_t *state);
void orc_poseidon_hash_many(int t, int r_f, int r_p, const fr_t *inputs, size_t n_msgs, size_t msg_len, fr_t *digests);
void orc_poseidon_merkle_root(int t, int r_f, int r_p, const fr_t *vectors, size_t n, size_t dim, fr_t *root);
/* dump spec constants in a flat layout (used by the product's table generator TEST, not by the product) */
void orc_poseidon_spec_dump(int t, int r_f, int r_p, fr_t *rc /*(r_f+r_p)*t*/, fr_t *mds /*t*t*/);

#ifdef __cplusplus
}
#endif
#endif
