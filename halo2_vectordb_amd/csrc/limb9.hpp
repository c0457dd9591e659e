// Field elements as nine 29-bit limbs ("L9"): the register / LDS form used inside the NTT tiles and the MSM accumulator.
//
// value = sum l[k] * 2^(29k).  "Normalised": l[0..7] < 2^29 (+ a few units after a parallel carry pass; exactly below
// 2^29 after l9_carry or a product).  Between normalisations limbs may grow to 7 * 2^29 (sums of a few normalised
// values); a product accepts a first operand with limbs below 6.1 * 2^29 and a normalised second operand
// (mont_core29, field.hpp).  Add and subtract propagate no carries: 9, resp. 18, plain 32-bit operations.
// Values are only ever defined modulo p; what is tracked (in comments, at every use) is an upper bound in multiples
// of p, because a product of values A, B returns something below A * B / 2^261 + p.
#pragma once
#include <math.h>

#include "field.hpp"

namespace vdb {

struct L9 {
  uint32_t l[9];
};
HD L9 l9_split(const u256& a) {
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int pb = 29 * k, wb = pb >> 5, ob = pb & 31;
    uint32_t lo = a.w[wb];
    uint32_t hi = wb + 1 < 8 ? a.w[wb + 1 < 8 ? wb + 1 : 0] : 0u;
    r.l[k] = ob ? ((lo >> ob) | (hi << (32 - ob))) : lo;
    if (k < 8) r.l[k] &= 0x1fffffffu;
  }
  return r;
}
// 32 * a as nine limbs (a < 2^256, so 32 a < 2^261): the second operand that keeps a nine-limb product of two values in
// ordinary Montgomery form (R = 2^256) in that form: (x R)(32 y R) / 2^261 = x y R
HD L9 l9_split32(const u256& a) {
  L9 r;
  r.l[0] = (a.w[0] << 5) & 0x1fffffffu;
#pragma unroll
  for (int k = 1; k < 9; k++) {
    int pb = 29 * k - 5, wb = pb >> 5, ob = pb & 31;
    uint32_t lo = a.w[wb];
    uint32_t hi = wb + 1 < 8 ? a.w[wb + 1 < 8 ? wb + 1 : 0] : 0u;
    r.l[k] = ob ? ((lo >> ob) | (hi << (32 - ob))) : lo;
    if (k < 8) r.l[k] &= 0x1fffffffu;
  }
  return r;
}
// exactly normalised limbs of a value below 2^256 -> eight words
HD u256 l9_pack(const L9& L) {
  u256 r;
  r.w[0] = L.l[0] | (L.l[1] << 29);
  r.w[1] = (L.l[1] >> 3) | (L.l[2] << 26);
  r.w[2] = (L.l[2] >> 6) | (L.l[3] << 23);
  r.w[3] = (L.l[3] >> 9) | (L.l[4] << 20);
  r.w[4] = (L.l[4] >> 12) | (L.l[5] << 17);
  r.w[5] = (L.l[5] >> 15) | (L.l[6] << 14);
  r.w[6] = (L.l[6] >> 18) | (L.l[7] << 11);
  r.w[7] = (L.l[7] >> 21) | (L.l[8] << 8);
  return r;
}
// one parallel carry pass: limbs below 2^32 in, limbs below 2^29 + 8 out (top limb takes what is left)
HD void l9_renorm(L9& x) {
  uint32_t c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = x.l[k] >> 29;
#pragma unroll
  for (int k = 0; k < 8; k++) x.l[k] &= 0x1fffffffu;
#pragma unroll
  for (int k = 1; k < 9; k++) x.l[k] += c[k - 1];
}
// full carry propagation: limbs 0..7 exactly below 2^29
HD void l9_carry(L9& x) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    x.l[k + 1] += x.l[k] >> 29;
    x.l[k] &= 0x1fffffffu;
  }
}
HD L9 l9_add(const L9& a, const L9& b) {
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + b.l[k];
  return r;
}
// a - t + K p, limb-wise.  `ckp` is K p written with limbs c[0] = v_0 + 2^29, c[k] = v_k + 2^29 - 1 (0 < k < 8),
// c[8] = v_8 - 1 (l9_offset_limbs), which dominate the limbs of any t with limbs 0..7 below 2^29 and value below
// (K - 1) p, so no limb ever goes negative.
HD L9 l9_sub(const L9& a, const L9& t, const uint32_t (&ckp)[9]) {
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = a.l[k] + (ckp[k] - t.l[k]);
  return r;
}
// K p - t
HD L9 l9_neg(const L9& t, const uint32_t (&ckp)[9]) {
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = ckp[k] - t.l[k];
  return r;
}
// a * b / 2^261 (+ less than p): exactly normalised limbs
template <class M>
HD L9 l9_mul(const L9& a, const L9& b) {
  L9 r;
  mont_core29<M>(r.l, a.l, b.l);
  return r;
}
// w * v mod p (below 3 p, exactly normalised limbs) for a constant w given as its plain residue with wq = floor(w 2^261 / p)
template <class M>
HD L9 l9_mul_shoup(const L9& v, const L9& w, const L9& wq) {
  L9 r;
  shoup_core29<M>(r.l, v.l, w.l, wq.l);
  return r;
}
// (a1 * b1 + a2 * b2) / 2^261 (+ less than p) with one reduction; limbs of a1 and a2 together below 6 * 2^29
template <class M>
HD L9 l9_mul2(const L9& a1, const L9& b1, const L9& a2, const L9& b2) {
  L9 r;
  mont_core29_2<M>(r.l, a1.l, b1.l, a2.l, b2.l);
  return r;
}
// a * a / 2^261 (+ less than p) for a normalised a (limbs below 2^29 + 8)
template <class M>
HD L9 l9_sqr(const L9& a) {
  L9 r;
  mont_sqr_core29<M>(r.l, a.l);
  return r;
}
// exactly normalised limbs of a value below 2p -> canonical eight words
template <class M>
HD u256 l9_canon(const L9& t) {
  return lazy_canon<M>(l9_pack(t));
}
// all limbs zero, or equal to p: the two representations of 0 below 2p (exactly normalised input)
template <class M>
HD bool l9_is_zero_mod(const L9& t) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    z |= t.l[k];
    e |= t.l[k] ^ M::P29[k];
  }
  return z == 0 || e == 0;
}
// host: the dominating limbs of K p for l9_sub; returns the largest of limbs 0..7 in units of 2^29 (the growth a
// difference adds to a limb)
template <class M>
inline double l9_offset_limbs(uint32_t K, uint32_t out[9]) {
  uint64_t carry = 0;
  for (int k = 0; k < 9; k++) {
    uint64_t v = (uint64_t)K * M::P29[k] + carry;
    out[k] = (uint32_t)(v & 0x1fffffffu);
    carry = v >> 29;
  }
  out[8] += (uint32_t)(carry << 29);
  out[0] += 1u << 29;
  for (int k = 1; k < 8; k++) out[k] += (1u << 29) - 1;
  out[8] -= 1;
  double cmax = 0;
  for (int k = 0; k < 8; k++) cmax = fmax(cmax, (double)out[k] / (double)(1u << 29));
  return cmax;
}

}  // namespace vdb
