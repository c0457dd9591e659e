// The transcript's permutation on AVX-512 IFMA (see hostperm.hpp): the state's t <= 8 field elements sit in the lanes of five
// vectors (one per 52-bit limb), in Montgomery form for R' = 2^260, lazily reduced (values stay below 2^260 ~ 84 r, limbs of every
// multiplicand below 2^52).  One vector product (25 x (vpmadd52luq, vpmadd52huq) + five reduction steps) costs what ONE scalar
// product costs in latency (~19 ns on an EPYC 9575F) and serves eight lanes: the full rounds' t S-boxes are three vector products,
// the dense mix t of them (independent), and in a partial round everything beside the S-box chain — the sum over the other state
// words, the column update — runs in the shadow of that chain, which is three products long in the form
// row_0 (x^5 + c) = (row_0 x) x^4 + row_0 c.  Built with -mavx512f -mavx512ifma -mavx512vl; picked at run time (transcript.hip).
// Same function as the portable build (tests/test_transcript_cpu.py holds the three builds against each other).
#include "hostperm.hpp"

#include <immintrin.h>

#include <cstdlib>
#include <cstring>
#include <new>

namespace vdb {
namespace {

typedef unsigned long long u64;
typedef unsigned __int128 u128;
constexpr u64 P52[5] = {0x1f593f0000001ull, 0x4879b9709143eull, 0x181585d2833e8ull, 0xa029b85045b68ull, 0x30644e72e131ull};
constexpr u64 INV52 = 0x1f593efffffffull, M52 = (1ull << 52) - 1;
constexpr u64 FR_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
// plain residues (radix 2^52): 2^264 mod r (into the R' domain from a value in 2^256 Montgomery form), 2^256 mod r (back), 2^260 mod r (one)
constexpr u64 K_IN[5] = {0x31f8c9ffffab6ull, 0xac31329faef6eull, 0x9e2a3495d7570ull, 0xe357276f48b70ull, 0xd791464ef86ull};
constexpr u64 K_OUT[5] = {0x6341c4ffffffbull, 0x959f60cd29ac9ull, 0x879462e36fc76ull, 0xdf2f666ea36f7ull, 0xe0a77c19a07ull};

struct V5 {
  __m512i l[5];
};
inline V5 v5_zero() {
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_setzero_si512();
  return r;
}
// a * b / 2^260 + addend (mod r): limbs of a, b below 2^52; the addend (any limbs below 2^58) joins the accumulator before the final carry pass.
// The result has limbs below 2^52 (the top one takes the rest) and is below a b / 2^260 + r + addend.
template <bool ADD>
inline __attribute__((always_inline)) V5 mm_impl(const V5& a, const V5& b, const V5* addend) {
  const __m512i z = _mm512_setzero_si512(), inv = _mm512_set1_epi64((long long)INV52), mask = _mm512_set1_epi64((long long)M52);
  __m512i p[5];
  for (int j = 0; j < 5; j++) p[j] = _mm512_set1_epi64((long long)P52[j]);
  __m512i acc[6] = {z, z, z, z, z, z};
#pragma GCC unroll 5
  for (int i = 0; i < 5; i++) {
#pragma GCC unroll 5
    for (int j = 0; j < 5; j++) {
      acc[j] = _mm512_madd52lo_epu64(acc[j], a.l[i], b.l[j]);
      acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], a.l[i], b.l[j]);
    }
    const __m512i m = _mm512_madd52lo_epu64(z, acc[0], inv);
#pragma GCC unroll 5
    for (int j = 0; j < 5; j++) {
      acc[j] = _mm512_madd52lo_epu64(acc[j], m, p[j]);
      acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], m, p[j]);
    }
    acc[1] = _mm512_add_epi64(acc[1], _mm512_srli_epi64(acc[0], 52));
    acc[0] = acc[1];
    acc[1] = acc[2];
    acc[2] = acc[3];
    acc[3] = acc[4];
    acc[4] = acc[5];
    acc[5] = z;
  }
  if (ADD)
    for (int k = 0; k < 5; k++) acc[k] = _mm512_add_epi64(acc[k], addend->l[k]);
  V5 r;
#pragma GCC unroll 4
  for (int k = 0; k < 4; k++) {
    acc[k + 1] = _mm512_add_epi64(acc[k + 1], _mm512_srli_epi64(acc[k], 52));
    r.l[k] = _mm512_and_si512(acc[k], mask);
  }
  r.l[4] = acc[4];
  return r;
}
// two independent products instruction by instruction in one stream: separate products overlap little in the reorder window (one is
// ~130 instructions long), a pair comes closer to the multiplier's throughput (the 12 accumulators and 20 operand limbs of a pair
// exceed the 32 vector registers: the allocator spills, which is what is left between this build and its arithmetic)
inline __attribute__((always_inline)) void mm_pair(const V5& a1, const V5& b1, const V5* add1, const V5& a2, const V5& b2, const V5* add2, V5& r1, V5& r2) {
  const __m512i z = _mm512_setzero_si512(), inv = _mm512_set1_epi64((long long)INV52), mask = _mm512_set1_epi64((long long)M52);
  __m512i acc[6] = {z, z, z, z, z, z}, bcc[6] = {z, z, z, z, z, z};
#pragma GCC unroll 5
  for (int i = 0; i < 5; i++) {
#pragma GCC unroll 5
    for (int j = 0; j < 5; j++) {
      acc[j] = _mm512_madd52lo_epu64(acc[j], a1.l[i], b1.l[j]);
      bcc[j] = _mm512_madd52lo_epu64(bcc[j], a2.l[i], b2.l[j]);
      acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], a1.l[i], b1.l[j]);
      bcc[j + 1] = _mm512_madd52hi_epu64(bcc[j + 1], a2.l[i], b2.l[j]);
    }
    const __m512i m1 = _mm512_madd52lo_epu64(z, acc[0], inv), m2 = _mm512_madd52lo_epu64(z, bcc[0], inv);
#pragma GCC unroll 5
    for (int j = 0; j < 5; j++) {
      const __m512i pj = _mm512_set1_epi64((long long)P52[j]);
      acc[j] = _mm512_madd52lo_epu64(acc[j], m1, pj);
      bcc[j] = _mm512_madd52lo_epu64(bcc[j], m2, pj);
      acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], m1, pj);
      bcc[j + 1] = _mm512_madd52hi_epu64(bcc[j + 1], m2, pj);
    }
    acc[1] = _mm512_add_epi64(acc[1], _mm512_srli_epi64(acc[0], 52));
    bcc[1] = _mm512_add_epi64(bcc[1], _mm512_srli_epi64(bcc[0], 52));
    for (int k = 0; k < 5; k++) acc[k] = acc[k + 1], bcc[k] = bcc[k + 1];
    acc[5] = z, bcc[5] = z;
  }
  if (add1)
    for (int k = 0; k < 5; k++) acc[k] = _mm512_add_epi64(acc[k], add1->l[k]);
  if (add2)
    for (int k = 0; k < 5; k++) bcc[k] = _mm512_add_epi64(bcc[k], add2->l[k]);
#pragma GCC unroll 4
  for (int k = 0; k < 4; k++) {
    acc[k + 1] = _mm512_add_epi64(acc[k + 1], _mm512_srli_epi64(acc[k], 52));
    bcc[k + 1] = _mm512_add_epi64(bcc[k + 1], _mm512_srli_epi64(bcc[k], 52));
    r1.l[k] = _mm512_and_si512(acc[k], mask);
    r2.l[k] = _mm512_and_si512(bcc[k], mask);
  }
  r1.l[4] = acc[4];
  r2.l[4] = bcc[4];
}
inline V5 mm(const V5& a, const V5& b) { return mm_impl<false>(a, b, nullptr); }
inline V5 mm_add(const V5& a, const V5& b, const V5& c) { return mm_impl<true>(a, b, &c); }
inline V5 v5_addl(const V5& a, const V5& b) {  // limb-wise, no carries: an addend for mm_add
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_add_epi64(a.l[k], b.l[k]);
  return r;
}
inline V5 v5_bcast(const V5& a, int lane) {
  const __m512i idx = _mm512_set1_epi64(lane);
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_permutexvar_epi64(idx, a.l[k]);
  return r;
}
inline V5 v5_hsum_lane0(const V5& a) {  // lane 0 <- sum of all lanes (limb-wise: an addend), other lanes zero
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_maskz_set1_epi64(1, _mm512_reduce_add_epi64(a.l[k]));
  return r;
}
inline V5 v5_blend0(const V5& lane0, const V5& rest) {
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_mask_blend_epi64(1, rest.l[k], lane0.l[k]);
  return r;
}
inline V5 v5_norm(const V5& a) {  // carry pass: limbs below 2^52 again
  const __m512i mask = _mm512_set1_epi64((long long)M52);
  __m512i c[5];
  for (int k = 0; k < 5; k++) c[k] = a.l[k];
  V5 r;
  for (int k = 0; k < 4; k++) {
    c[k + 1] = _mm512_add_epi64(c[k + 1], _mm512_srli_epi64(c[k], 52));
    r.l[k] = _mm512_and_si512(c[k], mask);
  }
  r.l[4] = c[4];
  return r;
}

// ---- scalar helpers for building the tables
// a (four 64-bit words, below r) -> 16 a mod r as five 52-bit limbs: the value in 2^256 Montgomery form becomes the 2^260 form
void to_rprime(const uint64_t* a, u64 out[5]) {
  u64 x[4] = {a[0], a[1], a[2], a[3]};
  for (int d = 0; d < 4; d++) {
    u64 y[4], c = 0;
    for (int i = 0; i < 4; i++) {
      const u64 t = (x[i] << 1) | c;
      c = x[i] >> 63;
      y[i] = t;
    }  // 2 x < 2^255: no carry out
    u64 s[4];
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      const u128 t = (u128)y[i] - FR_P[i] - (u64)bw;
      s[i] = (u64)t;
      bw = (t >> 64) & 1;
    }
    for (int i = 0; i < 4; i++) x[i] = bw ? y[i] : s[i];
  }
  out[0] = x[0] & M52;
  out[1] = ((x[0] >> 52) | (x[1] << 12)) & M52;
  out[2] = ((x[1] >> 40) | (x[2] << 24)) & M52;
  out[3] = ((x[2] >> 28) | (x[3] << 36)) & M52;
  out[4] = x[3] >> 16;
}
void set_lane(V5& v, int lane, const u64 limbs[5]) {
  alignas(64) u64 t[8];
  for (int k = 0; k < 5; k++) {
    _mm512_store_si512(t, v.l[k]);
    t[lane] = limbs[k];
    v.l[k] = _mm512_load_si512(t);
  }
}
V5 bcast_const(const u64 limbs[5]) {
  V5 r;
  for (int k = 0; k < 5; k++) r.l[k] = _mm512_set1_epi64((long long)limbs[k]);
  return r;
}

struct Tables {
  int t, half, rp;
  V5 k_in, k_out, one;
  V5* start;   // half + 1 (lane i: constant of state word i)
  V5* endc;    // half - 1
  V5* mds;     // t columns (lane i: M[i][j])
  V5* pre;     // t columns of the pre-sparse matrix
  V5* rvec;    // rp: lane j >= 1: sparse_row[p][j], lane 0: 0
  V5* r0l1;    // rp: lane 1: sparse_row[p][0] (the second factor of the product that makes x^2 in lane 0 and row_0 x in lane 1)
  V5* col;     // rp: lane j >= 1: sparse_col[p][j - 1], lane 0: 0
  V5* addc;    // rp: lane 0: sparse_row[p][0] * constant, lane 1: the round's constant
  V5* pool;
};

inline V5 pow5_add(const V5& x, const V5& c) {
  const V5 x2 = mm(x, x);
  const V5 x4 = mm(x2, x2);
  return mm_add(x4, x, c);
}
inline V5 dense(const V5* cols, const V5& s, int t) {
  // new word i = sum_j M[i][j] s_j: per j one vector product of column j with s_j in every lane; the products are independent
  V5 acc = v5_zero();
  int j = 0;
  for (; j + 1 < t; j += 2) {
    V5 r1, r2;
    mm_pair(cols[j], v5_bcast(s, j), nullptr, cols[j + 1], v5_bcast(s, j + 1), nullptr, r1, r2);
    acc = v5_addl(acc, v5_addl(r1, r2));
  }
  if (j < t) acc = v5_addl(acc, mm(cols[j], v5_bcast(s, j)));
  return v5_norm(acc);
}

}  // namespace

void* host_ifma_prepare(const HostPermView& o) {
  if (o.t < 2 || o.t > 8 || o.half < 1) return nullptr;
  const int t = o.t, half = o.half, rp = o.rp;
  const size_t n_vec = (size_t)(half + 1) + (half > 1 ? half - 1 : 0) + 2 * (size_t)t + 4 * (size_t)rp;
  Tables* T = new (std::nothrow) Tables();
  if (!T) return nullptr;
  T->pool = static_cast<V5*>(aligned_alloc(64, (n_vec ? n_vec : 1) * sizeof(V5)));
  if (!T->pool) {
    delete T;
    return nullptr;
  }
  for (size_t i = 0; i < n_vec; i++) T->pool[i] = v5_zero();
  T->t = t, T->half = half, T->rp = rp;
  V5* q = T->pool;
  T->start = q, q += half + 1;
  T->endc = q, q += half > 1 ? half - 1 : 0;
  T->mds = q, q += t;
  T->pre = q, q += t;
  T->rvec = q, q += rp;
  T->r0l1 = q, q += rp;
  T->col = q, q += rp;
  T->addc = q, q += rp;
  T->k_in = bcast_const(K_IN);
  T->k_out = bcast_const(K_OUT);
  {
    const uint64_t one_m[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};  // 2^256 mod r
    u64 l[5];
    to_rprime(one_m, l);
    T->one = bcast_const(l);
  }
  u64 l[5];
  for (int r = 0; r <= half; r++)
    for (int i = 0; i < t; i++) to_rprime(o.start + 4 * ((size_t)r * t + i), l), set_lane(T->start[r], i, l);
  for (int r = 0; r + 1 < half; r++)
    for (int i = 0; i < t; i++) to_rprime(o.end + 4 * ((size_t)r * t + i), l), set_lane(T->endc[r], i, l);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) {
      to_rprime(o.mds + 4 * ((size_t)i * t + j), l), set_lane(T->mds[j], i, l);
      to_rprime(o.pre_sparse + 4 * ((size_t)i * t + j), l), set_lane(T->pre[j], i, l);
    }
  for (int p = 0; p < rp; p++) {
    for (int j = 1; j < t; j++) {
      to_rprime(o.sparse_row + 4 * ((size_t)p * t + j), l), set_lane(T->rvec[p], j, l);
      to_rprime(o.sparse_col + 4 * ((size_t)p * (t - 1) + j - 1), l), set_lane(T->col[p], j, l);
    }
    u64 lc[5];
    to_rprime(o.sparse_row + 4 * ((size_t)p * t), l), set_lane(T->r0l1[p], 1, l);
    to_rprime(o.partial + 4 * (size_t)p, lc), set_lane(T->addc[p], 1, lc);
    V5 a = v5_zero(), b = v5_zero();
    set_lane(a, 0, l);
    set_lane(b, 0, lc);
    const V5 prod = mm(a, b);                // lane 0: row_0 c (in the R' form), below 1.1 r
    alignas(64) u64 tmp[8];
    for (int k = 0; k < 5; k++) {
      _mm512_store_si512(tmp, prod.l[k]);
      lc[k] = tmp[0];
    }
    set_lane(T->addc[p], 0, lc);
  }
  return T;
}
void host_ifma_free(void* tables) {
  Tables* T = static_cast<Tables*>(tables);
  if (!T) return;
  free(T->pool);
  delete T;
}

void host_permute_ifma(const void* tables, uint64_t* state) {
  const Tables& T = *static_cast<const Tables*>(tables);
  const int t = T.t, half = T.half;
  // in: words of four 64-bit limbs (2^256 Montgomery form, below r) -> lanes of 52-bit limbs, times 2^264 / 2^260, plus the first constants
  alignas(64) u64 lim[5][8];
  memset(lim, 0, sizeof lim);
  for (int i = 0; i < t; i++) {
    const u64* x = reinterpret_cast<const u64*>(state) + 4 * i;
    lim[0][i] = x[0] & M52;
    lim[1][i] = ((x[0] >> 52) | (x[1] << 12)) & M52;
    lim[2][i] = ((x[1] >> 40) | (x[2] << 24)) & M52;
    lim[3][i] = ((x[2] >> 28) | (x[3] << 36)) & M52;
    lim[4][i] = x[3] >> 16;
  }
  V5 st;
  for (int k = 0; k < 5; k++) st.l[k] = _mm512_load_si512(lim[k]);
  st = mm_add(st, T.k_in, T.start[0]);
  for (int r = 1; r <= half; r++) st = dense(r < half ? T.mds : T.pre, pow5_add(st, T.start[r]), t);
  // Partial rounds.  With x = word 0 and s = x^5 + c the new word 0 is row_0 s + sum_{j>=1} row_j word_j and word_j += s col_j.  Five
  // vector products per round, the independent ones sharing a product in different lanes (separate products do not overlap much:
  // one is ~130 instructions long and the reorder window holds three):
  //   P1 = x * [x | row_0]           lane 0: x^2, lane 1: row_0 x
  //   P2 = P1 * P1                   lane 0: x^4
  //   P3 = x^4 * [row_0 x | x] + [tp + row_0 c | c]     lane 0: the new word 0, lane 1: s      (row_0 s = (row_0 x) x^4 + row_0 c)
  //   P4 = s * col + words           lanes j >= 1: the new words
  //   P5 = row * words, summed over the lanes -> tp
  // The independent ones of consecutive rounds run as pairs: [P4 of the round before | P1], [P5 | P2], then the lane sum and P3.
  if (T.rp) {
    V5 bx = v5_bcast(st, 0), b1, p1, p2, t5, p3, upd;
    for (int k = 0; k < 5; k++) b1.l[k] = _mm512_mask_blend_epi64(2, bx.l[k], T.r0l1[0].l[k]);
    p1 = mm(bx, b1);
    for (int p = 0;; p++) {
      mm_pair(T.rvec[p], st, nullptr, p1, p1, nullptr, t5, p2);
      const V5 tp = v5_hsum_lane0(t5);
      const V5 x4 = v5_bcast(p2, 0), ub = v5_bcast(p1, 1);
      V5 b3;
      for (int k = 0; k < 5; k++) b3.l[k] = _mm512_mask_blend_epi64(2, ub.l[k], bx.l[k]);
      p3 = mm_add(x4, b3, v5_addl(tp, T.addc[p]));          // lane 0: the new word 0, lane 1: s = x^5 + c
      const V5 sb = v5_bcast(p3, 1);
      if (p + 1 == T.rp) {
        upd = mm_add(sb, T.col[p], st);
        st = v5_blend0(p3, upd);
        break;
      }
      bx = v5_bcast(p3, 0);
      for (int k = 0; k < 5; k++) b1.l[k] = _mm512_mask_blend_epi64(2, bx.l[k], T.r0l1[p + 1].l[k]);
      mm_pair(sb, T.col[p], &st, bx, b1, nullptr, upd, p1);
      st = v5_blend0(p3, upd);
      if ((p & 7) == 7) st = mm(st, T.one);   // the words 1 .. t-1 grow by up to 1.2 r per round: back below 1.2 r (word 0 too: same value)
    }
  }
  for (int r = 0; r + 1 < half; r++) st = dense(T.mds, pow5_add(st, T.endc[r]), t);
  st = dense(T.mds, pow5_add(st, v5_zero()), t);
  // out: times 2^256 / 2^260 (below 1.2 r), one conditional subtraction, back to four 64-bit limbs
  st = mm(st, T.k_out);
  for (int k = 0; k < 5; k++) _mm512_store_si512(lim[k], st.l[k]);
  for (int i = 0; i < t; i++) {
    u64 x[4];
    x[0] = lim[0][i] | (lim[1][i] << 52);
    x[1] = (lim[1][i] >> 12) | (lim[2][i] << 40);
    x[2] = (lim[2][i] >> 24) | (lim[3][i] << 28);
    x[3] = (lim[3][i] >> 36) | (lim[4][i] << 16);
    u64 s[4];
    u128 bw = 0;
    for (int w = 0; w < 4; w++) {
      const u128 d = (u128)x[w] - FR_P[w] - (u64)bw;
      s[w] = (u64)d;
      bw = (d >> 64) & 1;
    }
    u64* out = reinterpret_cast<u64*>(state) + 4 * i;
    for (int w = 0; w < 4; w++) out[w] = bw ? x[w] : s[w];
  }
}

}  // namespace vdb
