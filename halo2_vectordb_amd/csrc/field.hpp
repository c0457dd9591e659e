// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Representation: 8 x u32 little-endian limbs, Montgomery form with R = 2^256.  The byte layout is
// identical to halo2curves' `Fr`/`Fq` ([u64; 4] little-endian, Montgomery), i.e. what the
// reference's prover holds in memory behind `best_multiexp(&[Fr], &[G1Affine])` and `best_fft`
// (reached from /root/reference/src/scaffold/mod.rs:296), so buffers cross the C ABI zero-copy.
//
// CDNA4 has no 64x64 multiplier: the inner product step is v_mad_u64_u32 (32x32+64 -> 64), which on
// gfx950 issues at the rate of a 32-bit add (tools/valu_probe.hip).  All loops are fully unrolled so the
// operands stay in VGPRs (one element = 8 VGPRs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HD __host__ __device__ __forceinline__

namespace vdb {

struct alignas(16) u256 {
  uint32_t w[8];
};

struct FrParams {
  // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
  static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t INV = 0xefffffffu;  // -r^{-1} mod 2^32
  // 29-bit limbs of r and -r^{-1} mod 2^29 (mont_mul29)
  static constexpr uint32_t P29[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t INV29 = 0x0fffffffu;
};
struct FqParams {
  // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
  static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr uint32_t INV = 0xe4866389u;  // -q^{-1} mod 2^32
  static constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
  static constexpr uint32_t INV29 = 0x04866389u;
};

HD bool u256_is_zero(const u256& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.w[i];
  return o == 0;
}
HD bool u256_eq(const u256& a, const u256& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.w[i] ^ b.w[i];
  return o == 0;
}
// a >= b
HD bool u256_geq(const u256& a, const u256& b) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)a.w[i] - b.w[i] - borrow;
    borrow = (d >> 32) & 1;
  }
  return borrow == 0;
}
// o = a + b, returns carry
HD uint32_t u256_add(u256& o, const u256& a, const u256& b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.w[i] + b.w[i];
    o.w[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// o = a - b, returns borrow
HD uint32_t u256_sub(u256& o, const u256& a, const u256& b) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)a.w[i] - b.w[i] - borrow;
    o.w[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  return (uint32_t)borrow;
}
HD u256 u256_zero() {
  u256 z;
#pragma unroll
  for (int i = 0; i < 8; i++) z.w[i] = 0;
  return z;
}
HD u256 u256_from_u64(uint64_t v) {
  u256 z = u256_zero();
  z.w[0] = (uint32_t)v;
  z.w[1] = (uint32_t)(v >> 32);
  return z;
}
// Shifts by a runtime amount use static limb indices only: a runtime limb index (a.w[j]) would force the
// register array into scratch memory on the GPU.  Word shift by conditional moves (4, 2, 1 words), then bits.
// logical right shift by s in [0, 255]
HD u256 u256_shr(const u256& a, unsigned s) {
  u256 t = a;
  const unsigned ws = s >> 5, bs = s & 31;
  if (ws & 4) {
#pragma unroll
    for (int i = 0; i < 8; i++) t.w[i] = i + 4 < 8 ? t.w[i + 4 < 8 ? i + 4 : 0] : 0u;
  }
  if (ws & 2) {
#pragma unroll
    for (int i = 0; i < 8; i++) t.w[i] = i + 2 < 8 ? t.w[i + 2 < 8 ? i + 2 : 0] : 0u;
  }
  if (ws & 1) {
#pragma unroll
    for (int i = 0; i < 8; i++) t.w[i] = i + 1 < 8 ? t.w[i + 1 < 8 ? i + 1 : 0] : 0u;
  }
  if (bs) {
    u256 o;
#pragma unroll
    for (int i = 0; i < 7; i++) o.w[i] = (t.w[i] >> bs) | (t.w[i + 1] << (32 - bs));
    o.w[7] = t.w[7] >> bs;
    return o;
  }
  return t;
}
HD u256 u256_shl(const u256& a, unsigned s) {
  u256 t = a;
  const unsigned ws = s >> 5, bs = s & 31;
  if (ws & 4) {
#pragma unroll
    for (int i = 7; i >= 0; i--) t.w[i] = i - 4 >= 0 ? t.w[i - 4 >= 0 ? i - 4 : 0] : 0u;
  }
  if (ws & 2) {
#pragma unroll
    for (int i = 7; i >= 0; i--) t.w[i] = i - 2 >= 0 ? t.w[i - 2 >= 0 ? i - 2 : 0] : 0u;
  }
  if (ws & 1) {
#pragma unroll
    for (int i = 7; i >= 0; i--) t.w[i] = i - 1 >= 0 ? t.w[i - 1 >= 0 ? i - 1 : 0] : 0u;
  }
  if (bs) {
    u256 o;
#pragma unroll
    for (int i = 7; i >= 1; i--) o.w[i] = (t.w[i] << bs) | (t.w[i - 1] >> (32 - bs));
    o.w[0] = t.w[0] << bs;
    return o;
  }
  return t;
}
// a >> s for 0 < s < 32 with static limb indexing only (a runtime limb index would force the register array
// into scratch memory); used to walk a scalar digit by digit
HD u256 u256_shr_small(const u256& a, unsigned s) {
  u256 o;
#pragma unroll
  for (int i = 0; i < 7; i++) o.w[i] = (a.w[i] >> s) | (a.w[i + 1] << (32 - s));
  o.w[7] = a.w[7] >> s;
  return o;
}
// keep the low `bits` bits
HD u256 u256_low_bits(const u256& a, unsigned bits) {
  u256 o;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    unsigned lo = 32u * i;
    o.w[i] = bits >= lo + 32 ? a.w[i] : (bits <= lo ? 0u : a.w[i] & ((1u << (bits - lo)) - 1u));
  }
  return o;
}
// bit length (0 for zero)
HD unsigned u256_bits(const u256& a) {
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (a.w[i]) r = 32u * i + (32u - (unsigned)__builtin_clz(a.w[i]));
  return r;
}
HD uint32_t u256_bit(const u256& a, unsigned i) { return (a.w[i >> 5] >> (i & 31)) & 1u; }
// extract `len` (<= 32) bits starting at bit `pos`
HD uint32_t u256_extract(const u256& a, unsigned pos, unsigned len) {
  if (pos >= 256) return 0u;
  u256 t = u256_shr(a, pos);
  return len >= 32 ? t.w[0] : (t.w[0] & ((1u << len) - 1u));
}

template <class M>
HD u256 mod_p() {
  u256 p;
#pragma unroll
  for (int i = 0; i < 8; i++) p.w[i] = M::P[i];
  return p;
}

// Classic 32-bit-limb CIOS Montgomery product (kept as the cross-check of mont_mul and for the
// micro-benchmark): a*b*R^{-1} mod p with the "no final carry word" shortcut (top limb of p < 2^31).
template <class M>
HD u256 mont_mul32(const u256& a, const u256& b) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t A = (uint64_t)a.w[0] * b.w[i] + t[0];
    uint32_t m = (uint32_t)A * M::INV;
    uint64_t C = (uint64_t)m * M::P[0] + (uint32_t)A;
    A >>= 32;
    C >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      A += (uint64_t)a.w[j] * b.w[i] + t[j];
      C += (uint64_t)m * M::P[j] + (uint32_t)A;
      t[j - 1] = (uint32_t)C;
      A >>= 32;
      C >>= 32;
    }
    t[7] = (uint32_t)(A + C);
  }
  u256 r, p = mod_p<M>(), s;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = t[i];
  uint32_t borrow = u256_sub(s, r, p);
  return borrow ? r : s;
}

// Column-wise (product-scanning) core shared by every 29-bit-limb product: one 64-bit accumulator walks the 18
// columns; the carry of a column is the initial value of the next, so apart from the 162 multiply-adds a column costs
// one 64-bit shift (plus the two instructions that derive the reduction digit m_k in the low half).
// A: limbs below 6 * 2^29, B: limbs below 2^29  =>  every column stays below 2^64.  Returns (A * B + m * p) / 2^261
// as nine limbs (the top limb keeps all remaining bits).
// (Forcing each column into one dependent chain with inline-asm v_mad_u64_u32 was tried: the assembler pads every
// asm statement with an s_nop, which costs more than the 64-bit add per column the compiler spends on joining the
// two chains it builds to hide the latency of the reduction digit.)
HD void mad64(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
// On the device the three cores below are ONE inline-asm statement each (core29_*.inc, written by gen_core29.py): a single
// accumulator chain — 162 multiply-adds, 9 x (v_mul_lo_u32, v_and) for the reduction digits, 9 masks, 17 shifts = 206 vector
// instructions — where the compiler's schedule of the C++ below spends 243 (a second chain per column to hide the digit's latency,
// a 64-bit add to join the two, register moves).  Every one of these instructions issues at the same rate on gfx950 and the NTT and
// MSM kernels run at the issue ceiling, so the instruction count is the time: tools/core_probe.hip measures 169 against 148 G
// products/s, bit-identical.  The host build (and the reader) keeps the C++ form.
#define VDB_CORE29_OUT(o) "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8])
#define VDB_CORE29_IN(a) "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8])
#define VDB_CORE29_MOD(M)                                                                                                                          \
  "s"(M::P29[0]), "s"(M::P29[1]), "s"(M::P29[2]), "s"(M::P29[3]), "s"(M::P29[4]), "s"(M::P29[5]), "s"(M::P29[6]), "s"(M::P29[7]), "s"(M::P29[8]), \
      "s"(M::INV29)
#define VDB_CORE29_CLOBBER "vcc", "v30", "v31"
template <class M>
HD void mont_core29(uint32_t out[9], const uint32_t A[9], const uint32_t B[9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
#include "core29_mul.inc"
      : VDB_CORE29_OUT(out)
      : VDB_CORE29_IN(A), VDB_CORE29_IN(B), VDB_CORE29_MOD(M)
      : VDB_CORE29_CLOBBER);
#else
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t mq[9];
  uint32_t P[9];
#pragma unroll
  for (int j = 0; j < 9; j++) P[j] = M::P29[j];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mad64(acc, A[i], B[k - i]);
#pragma unroll
    for (int j = 1; j <= k; j++) mad64(acc, mq[k - j], P[j]);
    mq[k] = ((uint32_t)acc * M::INV29) & MASK;
    mad64(acc, mq[k], P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 18; k++) {
#pragma unroll
    for (int i = k - 8; i <= 8; i++) mad64(acc, A[i], B[k - i]);
#pragma unroll
    for (int j = k - 8; j <= 8; j++) mad64(acc, mq[k - j], P[j]);
    out[k - 9] = k < 17 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
    acc >>= 29;
  }
#endif
}

// Sum of two products under ONE reduction: (A1 * B1 + A2 * B2 + m * p) / 2^261.  Limbs of A1 + A2 together below
// 6 * 2^29 (column bound as in mont_core29), B1, B2 normalised.  Saves the 81 reduction multiply-adds of the second product.
template <class M>
HD void mont_core29_2(uint32_t out[9], const uint32_t A1[9], const uint32_t B1[9], const uint32_t A2[9], const uint32_t B2[9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm(
#include "core29_mul2.inc"
      : VDB_CORE29_OUT(out)
      : VDB_CORE29_IN(A1), VDB_CORE29_IN(B1), VDB_CORE29_IN(A2), VDB_CORE29_IN(B2), VDB_CORE29_MOD(M)
      : VDB_CORE29_CLOBBER);
#else
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t mq[9], P[9];
#pragma unroll
  for (int j = 0; j < 9; j++) P[j] = M::P29[j];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      mad64(acc, A1[i], B1[k - i]);
      mad64(acc, A2[i], B2[k - i]);
    }
#pragma unroll
    for (int j = 1; j <= k; j++) mad64(acc, mq[k - j], P[j]);
    mq[k] = ((uint32_t)acc * M::INV29) & MASK;
    mad64(acc, mq[k], P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 18; k++) {
#pragma unroll
    for (int i = k - 8; i <= 8; i++) {
      mad64(acc, A1[i], B1[k - i]);
      mad64(acc, A2[i], B2[k - i]);
    }
#pragma unroll
    for (int j = k - 8; j <= 8; j++) mad64(acc, mq[k - j], P[j]);
    out[k - 9] = k < 17 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
    acc >>= 29;
  }
#endif
}

// Product with a CONSTANT w (an NTT stage twiddle) whose quotient w' = floor(w 2^261 / p) was computed once: t = w v - q p with
// q = floor(w' v / 2^261) taken from the columns 7 .. 17 of w' v — at most two below floor(w v / p) (one for w', one for the dropped
// columns: their sum stays below 2^238 when v's limbs stay below 6.1 * 2^29) — and t from the low nine columns of
// w v + q (2^261 - p): exact, because 0 <= t < 3 p < 2^261.  143 multiply-adds and no chain of reduction digits against the
// Montgomery core's 162 + 9; 179 instructions as one asm statement.  V: limbs below 6.1 * 2^29, value below 2^261; W, WQ normalised.
// The result has exactly normalised limbs and is below 3 p; no factor 2^-261 (w is the plain residue, not a Montgomery form).
template <class M>
HD void shoup_core29(uint32_t out[9], const uint32_t V[9], const uint32_t W[9], const uint32_t WQ[9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t q[9];
  asm(
#include "core29_shoup.inc"
      : VDB_CORE29_OUT(out), VDB_CORE29_OUT(q)
      : VDB_CORE29_IN(V), VDB_CORE29_IN(W), VDB_CORE29_IN(WQ), "s"((0x20000000u) - M::P29[0]), "s"(0x1fffffffu - M::P29[1]), "s"(0x1fffffffu - M::P29[2]),
        "s"(0x1fffffffu - M::P29[3]), "s"(0x1fffffffu - M::P29[4]), "s"(0x1fffffffu - M::P29[5]), "s"(0x1fffffffu - M::P29[6]), "s"(0x1fffffffu - M::P29[7]),
        "s"(0x1fffffffu - M::P29[8])
      : VDB_CORE29_CLOBBER);
#else
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t q[9], NP[9];
#pragma unroll
  for (int k = 0; k < 9; k++) NP[k] = (k == 0 ? 0x20000000u : 0x1fffffffu) - M::P29[k];  // limbs of 2^261 - p (P29[0] != 0)
  uint64_t acc = 0;
#pragma unroll
  for (int k = 7; k < 18; k++) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) mad64(acc, WQ[i], V[k - i]);
    if (k >= 9) q[k - 9] = k < 17 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
    acc >>= 29;
  }
  acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      mad64(acc, W[i], V[k - i]);
      mad64(acc, q[i], NP[k - i]);
    }
    out[k] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
#endif
}

// Squaring variant: the 36 off-diagonal products are taken once against the doubled operand (45 multiply-adds instead
// of 81 for the product half).  A: limbs below 2^29 + 8 (normalised) so that a doubled column still fits 64 bits.
template <class M>
HD void mont_sqr_core29(uint32_t out[9], const uint32_t A[9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t D[9];
#pragma unroll
  for (int j = 0; j < 9; j++) D[j] = A[j] << 1;
  asm(
#include "core29_sqr.inc"
      : VDB_CORE29_OUT(out)
      : VDB_CORE29_IN(A), VDB_CORE29_IN(D), VDB_CORE29_MOD(M)
      : VDB_CORE29_CLOBBER);
#else
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t mq[9], A2[9], P[9];
#pragma unroll
  for (int j = 0; j < 9; j++) {
    P[j] = M::P29[j];
    A2[j] = A[j] << 1;
  }
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; 2 * i < k; i++) mad64(acc, A[i], A2[k - i]);
    if ((k & 1) == 0) mad64(acc, A[k / 2], A[k / 2]);
#pragma unroll
    for (int j = 1; j <= k; j++) mad64(acc, mq[k - j], P[j]);
    mq[k] = ((uint32_t)acc * M::INV29) & MASK;
    mad64(acc, mq[k], P[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 18; k++) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; i++) mad64(acc, A[i], A2[k - i]);
    if ((k & 1) == 0 && k / 2 <= 8) mad64(acc, A[k / 2], A[k / 2]);
#pragma unroll
    for (int j = k - 8; j <= 8; j++) mad64(acc, mq[k - j], P[j]);
    out[k - 9] = k < 17 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
    acc >>= 29;
  }
#endif
}

// THE Montgomery product of this library: 29-bit limbs (canonical in, canonical out, R = 2^256).
// Measured 118.7 G mul/s on MI355X against 90.2 G mul/s for mont_mul32 (vdb_bench_fr_mul).
// On gfx950 v_mad_u64_u32 issues at the rate of a plain 32-bit add, so the cost of the classic 32-bit-limb
// CIOS is dominated by carry handling and by staging {limb, 0} register pairs.  With 29-bit limbs a 64-bit
// column can absorb all 18 partial products (18 * 2^58 < 2^64) with no carry logic at all: every step is
// one multiply-add into a 64-bit accumulator.  Nine radix-2^29 reduction steps divide by 2^261; feeding
// a * 2^5 (a free change of limb offsets) makes the result a * b * 2^-256, i.e. the usual R.
template <class M>
HD u256 mont_mul(const u256& a, const u256& b) {
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t A[9], B[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    // limb k of (a << 5): bits [29k - 5, 29k + 24) of a
    int pos = 29 * k - 5;
    if (pos < 0) {
      A[k] = (a.w[0] << 5) & MASK;
    } else {
      int w = pos >> 5, o = pos & 31;
      uint32_t lo = w < 8 ? a.w[w < 8 ? w : 0] : 0u;
      uint32_t hi = w + 1 < 8 ? a.w[w + 1 < 8 ? w + 1 : 0] : 0u;
      A[k] = (o ? ((lo >> o) | (hi << (32 - o))) : lo) & MASK;
    }
    int pb = 29 * k, wb = pb >> 5, ob = pb & 31;
    uint32_t lob = b.w[wb];
    uint32_t hib = wb + 1 < 8 ? b.w[wb + 1 < 8 ? wb + 1 : 0] : 0u;
    B[k] = (ob ? ((lob >> ob) | (hib << (32 - ob))) : lob) & MASK;
  }
  uint32_t L[9];
  mont_core29<M>(L, A, B);
  u256 r;
  r.w[0] = L[0] | (L[1] << 29);
  r.w[1] = (L[1] >> 3) | (L[2] << 26);
  r.w[2] = (L[2] >> 6) | (L[3] << 23);
  r.w[3] = (L[3] >> 9) | (L[4] << 20);
  r.w[4] = (L[4] >> 12) | (L[5] << 17);
  r.w[5] = (L[5] >> 15) | (L[6] << 14);
  r.w[6] = (L[6] >> 18) | (L[7] << 11);
  r.w[7] = (L[7] >> 21) | (L[8] << 8);
  u256 s, p = mod_p<M>();
  uint32_t borrow = u256_sub(s, r, p);
  return borrow ? r : s;
}
// [0, 2p) -> [0, p)
template <class M>
HD u256 lazy_canon(const u256& a) {
  u256 s, p = mod_p<M>();
  uint32_t borrow = u256_sub(s, a, p);
  return borrow ? a : s;
}

template <class M>
HD u256 mont_sqr(const u256& a) {
  return mont_mul<M>(a, a);
}
template <class M>
HD u256 mod_add(const u256& a, const u256& b) {
  u256 r, s, p = mod_p<M>();
  u256_add(r, a, b);  // no carry: a, b < p < 2^254
  uint32_t borrow = u256_sub(s, r, p);
  return borrow ? r : s;
}
template <class M>
HD u256 mod_sub(const u256& a, const u256& b) {
  u256 r, s, p = mod_p<M>();
  uint32_t borrow = u256_sub(r, a, b);
  u256_add(s, r, p);
  return borrow ? s : r;
}
template <class M>
HD u256 mod_neg(const u256& a) {
  u256 r, p = mod_p<M>();
  u256_sub(r, p, a);
  return u256_is_zero(a) ? a : r;
}
template <class M>
HD u256 mod_dbl(const u256& a) {
  return mod_add<M>(a, a);
}
template <class M>
HD u256 mont_one() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = M::R1[i];
  return r;
}
template <class M>
HD u256 mont_r2() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = M::R2[i];
  return r;
}
template <class M>
HD u256 to_mont(const u256& canonical) {
  return mont_mul<M>(canonical, mont_r2<M>());
}
// a / R: on the device the reduction half of the core alone ((a << 5) + m p) / 2^261 — the product core in its asm form cannot
// drop the 72 multiply-adds by the zero limbs of the factor 1, which the compiler used to fold away (the MSM's sort converts every
// non-zero scalar of a column this way)
template <class M>
HD u256 from_mont(const u256& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t A[9], L[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int pos = 29 * k - 5;        // limb k of (a << 5), as in mont_mul
    if (pos < 0) {
      A[k] = (a.w[0] << 5) & MASK;
    } else {
      int w = pos >> 5, o = pos & 31;
      uint32_t lo = w < 8 ? a.w[w < 8 ? w : 0] : 0u;
      uint32_t hi = w + 1 < 8 ? a.w[w + 1 < 8 ? w + 1 : 0] : 0u;
      A[k] = (o ? ((lo >> o) | (hi << (32 - o))) : lo) & MASK;
    }
  }
  asm(
#include "core29_redc.inc"
      : VDB_CORE29_OUT(L)
      : VDB_CORE29_IN(A), VDB_CORE29_MOD(M)
      : VDB_CORE29_CLOBBER);
  u256 r;
  r.w[0] = L[0] | (L[1] << 29);
  r.w[1] = (L[1] >> 3) | (L[2] << 26);
  r.w[2] = (L[2] >> 6) | (L[3] << 23);
  r.w[3] = (L[3] >> 9) | (L[4] << 20);
  r.w[4] = (L[4] >> 12) | (L[5] << 17);
  r.w[5] = (L[5] >> 15) | (L[6] << 14);
  r.w[6] = (L[6] >> 18) | (L[7] << 11);
  r.w[7] = (L[7] >> 21) | (L[8] << 8);
  u256 s, p = mod_p<M>();
  uint32_t borrow = u256_sub(s, r, p);
  return borrow ? r : s;
#else
  u256 one = u256_from_u64(1);
  return mont_mul<M>(a, one);
#endif
}
// a^e (e canonical integer), left-to-right; not constant time (nothing here is secret-dependent
// beyond what the reference's own vartime code does)
template <class M>
HD u256 mont_pow(const u256& a, const u256& e) {
  u256 acc = mont_one<M>();
  int nb = (int)u256_bits(e);
  if (nb == 0) return acc;
  u256 ee = u256_shl(e, 256u - (unsigned)nb);  // top exponent bit at bit 255; then shift left one bit per step
  for (int i = 0; i < nb; i++) {
    acc = mont_sqr<M>(acc);
    if (ee.w[7] >> 31) acc = mont_mul<M>(acc, a);
    u256 t;
#pragma unroll
    for (int k = 7; k >= 1; k--) t.w[k] = (ee.w[k] << 1) | (ee.w[k - 1] >> 31);
    t.w[0] = ee.w[0] << 1;
    ee = t;
  }
  return acc;
}
// Fermat inverse; 0 -> 0
template <class M>
HD u256 mont_inv(const u256& a) {
  u256 e = mod_p<M>(), two = u256_from_u64(2), em2;
  u256_sub(em2, e, two);
  return mont_pow<M>(a, em2);
}

using Fr = FrParams;
using Fq = FqParams;

HD u256 fr_mul(const u256& a, const u256& b) { return mont_mul<Fr>(a, b); }
HD u256 fr_add(const u256& a, const u256& b) { return mod_add<Fr>(a, b); }
HD u256 fr_sub(const u256& a, const u256& b) { return mod_sub<Fr>(a, b); }
HD u256 fr_neg(const u256& a) { return mod_neg<Fr>(a); }
HD u256 fq_mul(const u256& a, const u256& b) { return mont_mul<Fq>(a, b); }
HD u256 fq_sqr(const u256& a) { return mont_mul<Fq>(a, a); }
HD u256 fq_add(const u256& a, const u256& b) { return mod_add<Fq>(a, b); }
HD u256 fq_sub(const u256& a, const u256& b) { return mod_sub<Fq>(a, b); }
HD u256 fq_neg(const u256& a) { return mod_neg<Fq>(a); }

// 16-byte vector load/store of one element (two dwordx4 per lane)
__device__ __forceinline__ u256 ld256(const u256* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  u256 r;
  r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
  r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
  return r;
}
__device__ __forceinline__ void st256(u256* p, const u256& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
  q[1] = make_uint4(v.w[4], v.w[5], v.w[6], v.w[7]);
}

}  // namespace vdb
