// Poseidon (x^5, BN254 Fr, T=3, RATE=2) shared definitions: the optimized-schedule constant table
// that lives in HBM and the device permutation used both by the hash-only kernels (b6) and by the
// witness-trace kernels (b5).  Parameters as the reference's call sites use them:
// T=3, RATE=2, R_F=8, R_P=57 (/root/reference/examples/merkle.rs:15-18, tests/vectordb/mod.rs:7-10).
#pragma once
#include <vector>

#include "field.hpp"
#include "limb9.hpp"

namespace vdb {

constexpr int PSD_T = 3;
constexpr int PSD_RATE = 2;
constexpr int PSD_RF = 8;
constexpr int PSD_RP = 57;
constexpr int PSD_HALF = PSD_RF / 2;

// optimized schedule of the PSE `poseidon` Spec (constants folded through the MDS, sparse partial rounds)
struct PoseidonSpec {
  u256 start[PSD_HALF + 1][PSD_T];  // start[0] = pre-constants added in the absorb step
  u256 partial[PSD_RP];
  u256 end[PSD_HALF - 1][PSD_T];
  u256 mds[PSD_T][PSD_T];
  u256 pre_sparse[PSD_T][PSD_T];
  u256 sparse_row[PSD_RP][PSD_T];
  u256 sparse_col[PSD_RP][PSD_T - 1];
  u256 cap;  // initial state word 0: 2^64 (Montgomery)
  u256 one;
};

// the same schedule for any width (row-major vectors): the transcript's sponge (t = 5) runs it on the host
struct PoseidonOpt {
  int t, half, rp;
  std::vector<u256> start;       // (half + 1) x t; row 0 = the first round's constants, added before the first S-box layer
  std::vector<u256> partial;     // rp
  std::vector<u256> end;         // (half - 1) x t
  std::vector<u256> mds, pre_sparse;  // t x t
  std::vector<u256> sparse_row;  // rp x t
  std::vector<u256> sparse_col;  // rp x (t - 1)
};
void poseidon_build_opt(int t, int r_f, int r_p, PoseidonOpt* out);
// host: plain parameters (round constants, MDS) for any width
void poseidon_plain_params(int t, int r_f, int r_p, std::vector<u256>& rc, std::vector<u256>& mds);
// host: Grain-LFSR parameter generation + optimisation (poseidon_spec.cpp)
void poseidon_build_spec(PoseidonSpec* out);
// device-resident copy (created on first use)
int poseidon_spec_dev(const PoseidonSpec** dev_out, const PoseidonSpec** host_out);

__device__ __forceinline__ u256 psd_pow5(const u256& x) {
  u256 x2 = fr_mul(x, x);
  u256 x4 = fr_mul(x2, x2);
  return fr_mul(x4, x);
}
__device__ __forceinline__ void psd_dense(u256 st[PSD_T], const u256 m[PSD_T][PSD_T]) {
  u256 r[PSD_T];
#pragma unroll
  for (int i = 0; i < PSD_T; i++) {
    u256 acc = fr_mul(m[i][0], st[0]);
#pragma unroll
    for (int j = 1; j < PSD_T; j++) acc = fr_add(acc, fr_mul(m[i][j], st[j]));
    r[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st[i] = r[i];
}
// PoseidonChip::permutation value semantics: absorb n_in (<= RATE) inputs with the pre-constants and
// the +1 padding marker, then the optimized rounds
__device__ __forceinline__ void psd_permute_absorb_words(const PoseidonSpec* __restrict__ sp, u256 st[PSD_T], const u256* in, int n_in) {
  st[0] = fr_add(st[0], sp->start[0][0]);
#pragma unroll
  for (int i = 0; i < PSD_RATE; i++) {
    if (i < n_in) st[1 + i] = fr_add(fr_add(st[1 + i], in[i]), sp->start[0][1 + i]);
    else if (i == n_in) st[1 + i] = fr_add(st[1 + i], fr_add(sp->start[0][1 + i], sp->one));
    else st[1 + i] = fr_add(st[1 + i], sp->start[0][1 + i]);
  }
  for (int r = 1; r < PSD_HALF; r++) {
#pragma unroll
    for (int i = 0; i < PSD_T; i++) st[i] = fr_add(psd_pow5(st[i]), sp->start[r][i]);
    psd_dense(st, sp->mds);
  }
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st[i] = fr_add(psd_pow5(st[i]), sp->start[PSD_HALF][i]);
  psd_dense(st, sp->pre_sparse);
  for (int p = 0; p < PSD_RP; p++) {
    st[0] = fr_add(psd_pow5(st[0]), sp->partial[p]);
    u256 n0 = fr_mul(sp->sparse_row[p][0], st[0]);
#pragma unroll
    for (int j = 1; j < PSD_T; j++) n0 = fr_add(n0, fr_mul(sp->sparse_row[p][j], st[j]));
#pragma unroll
    for (int i = 1; i < PSD_T; i++) st[i] = fr_add(fr_mul(st[0], sp->sparse_col[p][i - 1]), st[i]);
    st[0] = n0;
  }
  for (int r = 0; r < PSD_HALF - 1; r++) {
#pragma unroll
    for (int i = 0; i < PSD_T; i++) st[i] = fr_add(psd_pow5(st[i]), sp->end[r][i]);
    psd_dense(st, sp->mds);
  }
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st[i] = psd_pow5(st[i]);
  psd_dense(st, sp->mds);
}
// ---- the same permutation for the VALUE chains (leaf sponges, tree levels: one lane walks 600 dependent products per permutation and
// nothing hides them) in nine-limb form.  A word x is held as x 2^261 mod p (Montgomery R' = 2^261, what a nine-limb product divides
// by), lazily: sums just add limbs, a product's first operand may carry limbs up to 6 x 2^29, its second is normalised.  The spec's
// constants (R = 2^256 form) enter as 32 c = l9_split32(c); squarings use the squaring core, a dense row is two products under one
// reduction plus one.  ~113 k vector instructions per permutation against ~186 k for the eight-word Montgomery products.
struct PsdL9 {
  L9 s[PSD_T];
};
__device__ __forceinline__ L9 psd9_in(const u256& x) { return l9_split32(x); }
__device__ __forceinline__ u256 psd9_out(const L9& x) {  // limbs below 6 x 2^29, any value below 2^261 -> canonical R = 2^256 form
  return l9_canon<Fr>(l9_mul<Fr>(x, l9_split(mont_one<Fr>())));
}
__device__ __forceinline__ L9 psd9_pow5(L9 x) {
  l9_renorm(x);
  const L9 x2 = l9_sqr<Fr>(x), x4 = l9_sqr<Fr>(x2);
  return l9_mul<Fr>(x4, x);
}
__device__ __forceinline__ void psd9_dense(PsdL9& st, const u256 m[PSD_T][PSD_T]) {   // st: limbs below 2 x 2^29 each
  L9 r[PSD_T];
#pragma unroll
  for (int i = 0; i < PSD_T; i++)
    r[i] = l9_add(l9_mul2<Fr>(st.s[0], psd9_in(m[i][0]), st.s[1], psd9_in(m[i][1])), l9_mul<Fr>(st.s[2], psd9_in(m[i][2])));
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st.s[i] = r[i];
}
__device__ __forceinline__ void psd9_permute_absorb(const PoseidonSpec* __restrict__ sp, PsdL9& st, const u256* in, int n_in) {
  // on entry the words are sums of at most two normalised values (a dense layer's output) or fresh
  st.s[0] = l9_add(st.s[0], psd9_in(sp->start[0][0]));
#pragma unroll
  for (int i = 0; i < PSD_RATE; i++) {
    if (i < n_in) st.s[1 + i] = l9_add(l9_add(st.s[1 + i], psd9_in(in[i])), psd9_in(sp->start[0][1 + i]));
    else if (i == n_in) st.s[1 + i] = l9_add(st.s[1 + i], psd9_in(fr_add(sp->start[0][1 + i], sp->one)));
    else st.s[1 + i] = l9_add(st.s[1 + i], psd9_in(sp->start[0][1 + i]));
  }
  for (int r = 1; r < PSD_HALF; r++) {
#pragma unroll
    for (int i = 0; i < PSD_T; i++) st.s[i] = l9_add(psd9_pow5(st.s[i]), psd9_in(sp->start[r][i]));
    psd9_dense(st, sp->mds);
  }
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st.s[i] = l9_add(psd9_pow5(st.s[i]), psd9_in(sp->start[PSD_HALF][i]));
  psd9_dense(st, sp->pre_sparse);
  for (int p = 0; p < PSD_RP; p++) {
    // words 1, 2 take one product more every round: a carry pass every second round keeps their limbs below 4 x 2^29 (their
    // values grow to at most ~70 p < 2^261 over the 57 rounds, which a product's first operand may be)
    if (p & 1) {
      l9_renorm(st.s[1]);
      l9_renorm(st.s[2]);
    }
    st.s[0] = l9_add(psd9_pow5(st.s[0]), psd9_in(sp->partial[p]));
    const L9 n0 = l9_add(l9_mul2<Fr>(st.s[0], psd9_in(sp->sparse_row[p][0]), st.s[1], psd9_in(sp->sparse_row[p][1])),
                         l9_mul<Fr>(st.s[2], psd9_in(sp->sparse_row[p][2])));
#pragma unroll
    for (int i = 1; i < PSD_T; i++) st.s[i] = l9_add(l9_mul<Fr>(st.s[0], psd9_in(sp->sparse_col[p][i - 1])), st.s[i]);
    st.s[0] = n0;
  }
  l9_renorm(st.s[1]);
  l9_renorm(st.s[2]);
  for (int r = 0; r < PSD_HALF - 1; r++) {
#pragma unroll
    for (int i = 0; i < PSD_T; i++) st.s[i] = l9_add(psd9_pow5(st.s[i]), psd9_in(sp->end[r][i]));
    psd9_dense(st, sp->mds);
  }
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st.s[i] = psd9_pow5(st.s[i]);
  psd9_dense(st, sp->mds);
}

// what the hash-only kernels and the value chains call: canonical words in and out, the nine-limb permutation in between
// (psd_permute_absorb_words is the same permutation on eight-word Montgomery products: kept as the readable form)
__device__ __forceinline__ void psd_permute_absorb(const PoseidonSpec* __restrict__ sp, u256 st[PSD_T], const u256* in, int n_in) {
  PsdL9 t;
#pragma unroll
  for (int i = 0; i < PSD_T; i++) t.s[i] = psd9_in(st[i]);
  psd9_permute_absorb(sp, t, in, n_in);
#pragma unroll
  for (int i = 0; i < PSD_T; i++) st[i] = psd9_out(t.s[i]);
}
// sponge: clear(); update(msg[0..len)); squeeze()  (elements `stride` apart)
__device__ __forceinline__ u256 psd_hash(const PoseidonSpec* __restrict__ sp, const u256* msg, size_t len, size_t stride) {
  u256 st[PSD_T];
  st[0] = sp->cap;
  st[1] = u256_zero();
  st[2] = u256_zero();
  size_t off = 0;
  int pad = 0;
  while (off < len) {
    int c = (len - off) < (size_t)PSD_RATE ? (int)(len - off) : PSD_RATE;
    u256 in[PSD_RATE];
    in[0] = ld256(msg + off * stride);
    in[1] = c > 1 ? ld256(msg + (off + 1) * stride) : u256_zero();
    pad = PSD_RATE - c;
    psd_permute_absorb(sp, st, in, c);
    off += (size_t)c;
  }
  if (pad == 0) psd_permute_absorb(sp, st, nullptr, 0);
  return st[1];
}

}  // namespace vdb
