// See hostperm.hpp.  Built as hostperm_generic.o (no flags) and hostperm_mulx.o (-mbmi2 -madx -DVDB_HOSTPERM_MULX).
// The permutation is the optimised schedule of the PSE `poseidon` Spec (the one the chip runs on the device): the first round's
// constants, half - 1 full rounds with folded constants and the MDS matrix, one with the pre-sparse matrix, the partial rounds
// as sparse matrices (2 t - 1 products instead of t^2), the remaining full rounds.  Same function as the textbook schedule
// (tests/test_transcript_cpu.py holds it against an independent Python restatement of that one, in both builds).
// A proof of 2 x 10^4 columns absorbs ~2 x 10^5 values, i.e. ~5 x 10^4 permutations of 68 rounds at width 5: sums of products
// are reduced once (dot), single products by a four-step CIOS.
#include "hostperm.hpp"

#include <cstring>
#ifdef VDB_HOSTPERM_MULX
#include <immintrin.h>
#endif

namespace vdb {
namespace {

typedef unsigned long long u64;   // what the x86 intrinsics take
struct F4 {
  u64 l[4];
};
constexpr u64 FR_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr u64 FR_INV = 0xc2e1f593efffffffull;  // -r^-1 mod 2^64

#ifdef VDB_HOSTPERM_MULX
#define VDB_HOSTPERM_NAME(x) x##_mulx

// a * b[i] as a five-word row: four mulx and one carry chain
#define ROW(x, w, lo0, lo1, lo2, lo3, top)                                                  \
  {                                                                                          \
    u64 h0, h1, h2, h3;                                                                      \
    lo0 = _mulx_u64((x)[0], w, &h0);                                                         \
    lo1 = _mulx_u64((x)[1], w, &h1);                                                         \
    lo2 = _mulx_u64((x)[2], w, &h2);                                                         \
    lo3 = _mulx_u64((x)[3], w, &h3);                                                         \
    unsigned char c_ = _addcarry_u64(0, lo1, h0, &lo1);                                      \
    c_ = _addcarry_u64(c_, lo2, h1, &lo2);                                                   \
    c_ = _addcarry_u64(c_, lo3, h2, &lo3);                                                   \
    _addcarry_u64(c_, h3, 0, &top);                                                          \
  }

inline F4 f4_add(const F4& a, const F4& b) {
  u64 r0, r1, r2, r3, s0, s1, s2, s3;
  unsigned char c = _addcarry_u64(0, a.l[0], b.l[0], &r0);
  c = _addcarry_u64(c, a.l[1], b.l[1], &r1);
  c = _addcarry_u64(c, a.l[2], b.l[2], &r2);
  _addcarry_u64(c, a.l[3], b.l[3], &r3);          // a + b < 2 r < 2^255: no carry out
  c = _subborrow_u64(0, r0, FR_P[0], &s0);
  c = _subborrow_u64(c, r1, FR_P[1], &s1);
  c = _subborrow_u64(c, r2, FR_P[2], &s2);
  c = _subborrow_u64(c, r3, FR_P[3], &s3);
  return c ? F4{{r0, r1, r2, r3}} : F4{{s0, s1, s2, s3}};
}
// CIOS without the extra carry word: the modulus' top word is below 2^63 - 1, so t stays below 2 r through every step
inline F4 f4_mul(const F4& a, const F4& b) {
  u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma GCC unroll 4
  for (int i = 0; i < 4; i++) {
    u64 lo0, lo1, lo2, lo3, A, B;
    ROW(a.l, b.l[i], lo0, lo1, lo2, lo3, A)
    unsigned char c = _addcarry_u64(0, t0, lo0, &t0);
    c = _addcarry_u64(c, t1, lo1, &t1);
    c = _addcarry_u64(c, t2, lo2, &t2);
    c = _addcarry_u64(c, t3, lo3, &t3);
    _addcarry_u64(c, A, 0, &A);
    const u64 m = t0 * FR_INV;
    ROW(FR_P, m, lo0, lo1, lo2, lo3, B)
    c = _addcarry_u64(0, t0, lo0, &t0);            // clears word 0
    c = _addcarry_u64(c, t1, lo1, &t0);
    c = _addcarry_u64(c, t2, lo2, &t1);
    c = _addcarry_u64(c, t3, lo3, &t2);
    _addcarry_u64(c, A, B, &t3);
  }
  u64 s0, s1, s2, s3;
  unsigned char bw = _subborrow_u64(0, t0, FR_P[0], &s0);
  bw = _subborrow_u64(bw, t1, FR_P[1], &s1);
  bw = _subborrow_u64(bw, t2, FR_P[2], &s2);
  bw = _subborrow_u64(bw, t3, FR_P[3], &s3);
  return bw ? F4{{t0, t1, t2, t3}} : F4{{s0, s1, s2, s3}};
}
// sum_k a[k] b[k stride_b] with ONE Montgomery reduction: the n (<= 16) double-width products are summed row by row into a
// ten-word accumulator (the sum stays below 16 r^2 < 2^512), then reduced word by word
inline F4 f4_dot(const F4* a, const F4* b, int n, size_t stride_b = 1) {
  u64 T[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = n - 1; k >= 0; k--) {   // pair 0 last: in a partial round it is the one that waits for the S-box
    const u64* x = a[k].l;
    const u64* y = b[(size_t)k * stride_b].l;
#pragma GCC unroll 4
    for (int i = 0; i < 4; i++) {
      u64 lo0, lo1, lo2, lo3, A;
      ROW(x, y[i], lo0, lo1, lo2, lo3, A)
      unsigned char c = _addcarry_u64(0, T[i], lo0, &T[i]);
      c = _addcarry_u64(c, T[i + 1], lo1, &T[i + 1]);
      c = _addcarry_u64(c, T[i + 2], lo2, &T[i + 2]);
      c = _addcarry_u64(c, T[i + 3], lo3, &T[i + 3]);
      c = _addcarry_u64(c, T[i + 4], A, &T[i + 4]);
      for (int w = i + 5; w < 9; w++) c = _addcarry_u64(c, T[w], 0, &T[w]);
    }
  }
#pragma GCC unroll 4
  for (int i = 0; i < 4; i++) {  // T += m r 2^(64 i) clears word i
    const u64 m = T[i] * FR_INV;
    u64 lo0, lo1, lo2, lo3, B;
    ROW(FR_P, m, lo0, lo1, lo2, lo3, B)
    unsigned char c = _addcarry_u64(0, T[i], lo0, &T[i]);
    c = _addcarry_u64(c, T[i + 1], lo1, &T[i + 1]);
    c = _addcarry_u64(c, T[i + 2], lo2, &T[i + 2]);
    c = _addcarry_u64(c, T[i + 3], lo3, &T[i + 3]);
    c = _addcarry_u64(c, T[i + 4], B, &T[i + 4]);
    for (int w = i + 5; w < 9; w++) c = _addcarry_u64(c, T[w], 0, &T[w]);
  }
  // below n r + r: a few conditional subtractions, the word above 256 bits included
  u64 hi = T[8], r0 = T[4], r1 = T[5], r2 = T[6], r3 = T[7];
  for (;;) {
    u64 s0, s1, s2, s3;
    unsigned char bw = _subborrow_u64(0, r0, FR_P[0], &s0);
    bw = _subborrow_u64(bw, r1, FR_P[1], &s1);
    bw = _subborrow_u64(bw, r2, FR_P[2], &s2);
    bw = _subborrow_u64(bw, r3, FR_P[3], &s3);
    if (bw && !hi) break;
    hi -= bw;
    r0 = s0, r1 = s1, r2 = s2, r3 = s3;
  }
  return F4{{r0, r1, r2, r3}};
}

#else  // any x86-64 (or anything else with a 128-bit integer type)
#define VDB_HOSTPERM_NAME(x) x##_generic

inline bool geq_p(const u64 t[4]) {
  for (int i = 3; i >= 0; i--)
    if (t[i] != FR_P[i]) return t[i] > FR_P[i];
  return true;
}
inline void sub_p(u64 t[4]) {
  unsigned __int128 b = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)t[i] - FR_P[i] - (u64)b;
    t[i] = (u64)d;
    b = (d >> 64) & 1;
  }
}
inline F4 f4_add(const F4& a, const F4& b) {
  F4 r;
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.l[i] + b.l[i];
    r.l[i] = (u64)c;
    c >>= 64;
  }
  if (c || geq_p(r.l)) sub_p(r.l);  // a + b < 2 r < 2^255: no carry out in fact
  return r;
}
inline F4 f4_mul(const F4& a, const F4& b) {  // CIOS
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (u64)c;
    t[5] = (u64)(c >> 64);
    const u64 m = t[0] * FR_INV;
    c = (unsigned __int128)m * FR_P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (unsigned __int128)m * FR_P[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (u64)c;
    t[4] = t[5] + (u64)(c >> 64);
  }
  F4 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_p(r.l)) sub_p(r.l);
  return r;
}
// sum_k a[k] b[k stride_b] with ONE Montgomery reduction; product scanning: column c collects every x_i y_j with i + j = c of
// every pair into a three-word accumulator
inline F4 f4_dot(const F4* a, const F4* b, int n, size_t stride_b = 1) {
  u64 T[9];
  u64 a0 = 0, a1 = 0, a2 = 0;
  for (int c = 0; c < 7; c++) {
    const int ilo = c > 3 ? c - 3 : 0, ihi = c < 3 ? c : 3;
    for (int k = 0; k < n; k++) {
      const u64* x = a[k].l;
      const u64* y = b[(size_t)k * stride_b].l;
      for (int i = ilo; i <= ihi; i++) {
        const unsigned __int128 pr = (unsigned __int128)x[i] * y[c - i];
        const unsigned __int128 s = (unsigned __int128)a0 + (u64)pr;
        a0 = (u64)s;
        const unsigned __int128 s1 = (unsigned __int128)a1 + (u64)(pr >> 64) + (u64)(s >> 64);
        a1 = (u64)s1;
        a2 += (u64)(s1 >> 64);
      }
    }
    T[c] = a0;
    a0 = a1;
    a1 = a2;
    a2 = 0;
  }
  T[7] = a0;
  T[8] = a1;
  for (int i = 0; i < 4; i++) {  // T += m r 2^(64 i) clears word i
    const u64 m = T[i] * FR_INV;
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (unsigned __int128)m * FR_P[j] + T[i + j];
      T[i + j] = (u64)c;
      c >>= 64;
    }
    for (int w = i + 4; w < 9; w++) {
      c += T[w];
      T[w] = (u64)c;
      c >>= 64;
    }
  }
  u64 hi = T[8];
  F4 r = {{T[4], T[5], T[6], T[7]}};
  while (hi || geq_p(r.l)) {
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)r.l[i] - FR_P[i] - (u64)bw;
      r.l[i] = (u64)d;
      bw = (d >> 64) & 1;
    }
    hi -= (u64)bw;
  }
  return r;
}
#endif

inline F4 f4_pow5(const F4& x) {
  const F4 x2 = f4_mul(x, x);
  return f4_mul(f4_mul(x2, x2), x);
}

}  // namespace

void VDB_HOSTPERM_NAME(host_permute)(const HostPermView& o, uint64_t* state) {
  const int t = o.t, half = o.half;
  F4 st[16], nx[16];
  const F4* start = reinterpret_cast<const F4*>(o.start);
  const F4* mds = reinterpret_cast<const F4*>(o.mds);
  const F4* pre = reinterpret_cast<const F4*>(o.pre_sparse);
  const F4* srow = reinterpret_cast<const F4*>(o.sparse_row);
  const F4* scol = reinterpret_cast<const F4*>(o.sparse_col);
  const F4* end = reinterpret_cast<const F4*>(o.end);
  const F4* partial = reinterpret_cast<const F4*>(o.partial);
  memcpy(st, state, (size_t)t * 32);
  for (int i = 0; i < t; i++) st[i] = f4_add(st[i], start[i]);
  auto dense = [&](const F4* m) {
    for (int i = 0; i < t; i++) nx[i] = f4_dot(m + (size_t)i * t, st, t);
    for (int i = 0; i < t; i++) st[i] = nx[i];
  };
  for (int r = 1; r <= half; r++) {
    for (int i = 0; i < t; i++) st[i] = f4_add(f4_pow5(st[i]), start[(size_t)r * t + i]);
    dense(r < half ? mds : pre);
  }
  for (int p = 0; p < o.rp; p++) {
    st[0] = f4_add(f4_pow5(st[0]), partial[p]);
    const F4 n0 = f4_dot(srow + (size_t)p * t, st, t);
    for (int i = 1; i < t; i++) st[i] = f4_add(f4_mul(st[0], scol[(size_t)p * (t - 1) + i - 1]), st[i]);
    st[0] = n0;
  }
  for (int r = 0; r < half - 1; r++) {
    for (int i = 0; i < t; i++) st[i] = f4_add(f4_pow5(st[i]), end[(size_t)r * t + i]);
    dense(mds);
  }
  for (int i = 0; i < t; i++) st[i] = f4_pow5(st[i]);
  dense(mds);
  memcpy(state, st, (size_t)t * 32);
}

// acc <- acc x + v_i for i = 0 .. n - 1
void VDB_HOSTPERM_NAME(host_horner)(const uint64_t* values, size_t n, const uint64_t* x, uint64_t* acc) {
  F4 a, xx;
  memcpy(a.l, acc, 32);
  memcpy(xx.l, x, 32);
  for (size_t i = 0; i < n; i++) {
    F4 v;
    memcpy(v.l, values + 4 * i, 32);
    a = f4_add(f4_mul(a, xx), v);
  }
  memcpy(acc, a.l, 32);
}

}  // namespace vdb
