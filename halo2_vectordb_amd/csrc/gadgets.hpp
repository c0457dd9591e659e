// Witness-stream generators for the reference's gadgets, written once as host+device code.
//
// What is emitted: exactly the sequence of cells the Rust gadgets push into halo2-base's
// `Context.advice` (and `cells_to_lookup`), in the same order — SURVEY §8 a1-a28, b5:
//   FixedPointChip   /root/reference/src/gadget/fixed_point.rs   (line refs at each function)
//   DistanceChip     /root/reference/src/gadget/distance.rs
//   VectorDBChip     /root/reference/src/gadget/vectordb.rs
// on top of the halo2-base v0.3 GateChip/RangeChip cell templates ([UPSTREAM-RECALL]).
//
// GPU execution model.  Cell counts are data independent, so every sub-gadget sits at a static
// offset of the stream.  A generator runs over a *window* [lo, hi) of absolute stream positions:
// cells outside the window are not stored, and whole sub-gadgets that lie outside are skipped
// through a value-only fast path (`skip()`), using the size table `Sizes` that the host obtains by
// running this same code in counting mode.  Kernels give each wavefront one window of 64 different
// instances, so all lanes follow the same control flow (no divergence) while a long sequential
// gadget (e.g. qsqrt, ~20k cells) is cut into many windows that run concurrently.
//
// Values are Montgomery Fr (u256) throughout, the byte layout the prover consumes.
#pragma once
#include "field.hpp"

// mid-level generators are real functions on the device: the call tree is deep (qsqrt -> qlog2 -> qmul
// -> signed_div_scale -> qabs -> is_neg -> range checks) and full inlining would explode code size.
// Each `X` below is a thin inline member that calls the out-of-line `X__ool(context by value, args by value)`,
// which runs the inlined body `X__body` on its private copy of the context (see EmitOut).
#define HDN __host__ __device__ __noinline__

namespace vdb {

struct Sizes {  // [0] advice cells, [1] lookup cells
  uint32_t is_neg[2], qabs[2], sds[2], qmul[2], qdiv[2], qmin[2], cpow2[2], poly13[2], poly15[2], qlog2[2], qexp2[2], qlog[2], qexp[2],
      qsqrt[2], sfi254[2];
};

// constants of one FixedPointChip<P> + RangeChip(L) instance, resident in HBM
struct FpTables {
  uint32_t P, L;
  u256 pow2[254];        // 2^i
  u256 one;
  u256 scale;            // 2^P
  u256 exp2_poly[13], log_poly[15];  // fixed_point.rs:138-187, quantized
  u256 c_half, c_ln2, c_log2e, c_one_q;
  u256 sin_poly[15], c_pi, c_two_pi, c_half_pi, c_two;   // fixed_point.rs:189-211 and the constants of qsin / qcos / qsinh
  u256 small[260];       // i
  u256 small_inv[260];   // 1/i (i >= 1)
  const u256* limb_tab;  // Montgomery form of 0 .. 2^L - 1 (device pointer)
  Sizes sz;
};

// What does not change over the kernels of one call: where the streams lie, the rank's window, the chip's tables, the deferred-
// inversion list.  It lives in CONSTANT memory on the device (g_winv, written by the host entry before the call's kernels:
// witness.hip set_winv) and in a thread-local on the host (counting runs), NOT in the context: the out-of-line generators take the
// context by value, and a 132-byte struct travels through scratch memory — stores before every call, loads after, and on gfx9 a
// load's wait is a wait for every cell store queued before it (one in-order vmcnt).  With these fields read through scalar loads the
// context is ten dwords and travels in registers.
struct WInv {
  u256* adv;          // advice stream base (absolute indexing)
  uint8_t* sel;       // optional gate-start bits (keygen run), may be null
  u256* lk;           // lookup stream base
  // rank window (multi-GPU): only cells of the columns this rank commits are stored; everything else runs
  // through the value-only paths.  Advice positions [rlo, rhi), lookup positions [rllo, rlhi).
  uint64_t rlo, rhi, rllo, rlhi;
  const FpTables* T;
  // deferred inversions (halo2's Assigned::Rational + batch_invert): instead of a 380-product Fermat chain on
  // the critical path, (cell position, denominator) is appended here and k_inv_fixup patches the cell later
  uint64_t* inv_pos;
  u256* inv_val;
  uint32_t* inv_cnt;
  uint32_t inv_cap;
};
static __constant__ WInv g_winv;
inline thread_local WInv h_winv{};
HD const WInv& winv() {
#if defined(__HIP_DEVICE_COMPILE__)
  return g_winv;
#else
  return h_winv;
#endif
}

struct WCtx {
  uint64_t pos, lpos; // absolute positions of the next advice / lookup cell
  uint64_t lo, hi;    // emit window over advice positions (the slice of a sequential gadget this wavefront owns)
  bool count_only;    // host sizing run: count cells, store nothing, never skip
  int err;

  HD bool in_window(uint64_t p) const { return p >= lo && p < hi; }
  HD bool in_rank(uint64_t p) const { return p >= winv().rlo && p < winv().rhi; }
  // `cst`: the cell holds a data-independent constant of the gate template (QuantumCell::Constant); recorded in
  // bit 1 of the keygen-side flag byte so the prover's MSM can take those cells from a precomputed point
  HD void push(const u256& v, bool gate, bool cst = false) {
    if (!count_only && in_window(pos) && in_rank(pos)) {
      winv().adv[pos] = v;
      if (winv().sel) winv().sel[pos] = (uint8_t)((gate ? 1 : 0) | (cst ? 2 : 0));
    }
    pos++;
  }
  // a lookup cell belongs to the window that owns the advice cell pushed just before it
  HD void lookup(const u256& v) {
    if (!count_only && pos > 0 && in_window(pos - 1) && lpos >= winv().rllo && lpos < winv().rlhi) winv().lk[lpos] = v;
    lpos++;
  }
  // keygen-style runs only (sel != null): bit 2 of the flag byte marks the advice cell at `p` as the cell a lookup cell copies
  // (cells_to_lookup holds copies of advice cells; the j-th marked cell in stream order is the source of lookup cell j)
  HD void mark_lookup_source(uint64_t p) {
    if (winv().sel && !count_only && in_window(p) && in_rank(p)) winv().sel[p] |= 4;
  }
  // true when a sub-gadget of `cells` advice cells (and `lks` lookup cells) starting here cannot touch the windows
  HD bool skip(uint32_t cells, uint32_t lks = 0) const {
    if (count_only) return false;
    const bool seg_out = pos + cells <= lo || pos > hi;
    const bool rank_out = (pos + cells <= winv().rlo || pos >= winv().rhi) && (lks == 0 || lpos + lks <= winv().rllo || lpos >= winv().rlhi);
    return seg_out || rank_out;
  }
  HD bool skip2(const uint32_t sz[2]) const { return skip(sz[0], sz[1]); }
  HD void advance(const uint32_t sz[2]) {
    pos += sz[0];
    lpos += sz[1];
  }
};

// What an out-of-line generator hands back.  The generators take the context BY VALUE and return the fields they
// advance: a context passed by reference would live in scratch memory and, because every cell store may alias it, be
// re-read from there around every single push (measured: 3 loads per store in the distance kernels).
struct EmitOut {
  u256 v;
  uint64_t pos, lpos;
  int err;
  HD void take(const WCtx& c) {
    pos = c.pos;
    lpos = c.lpos;
    err = c.err;
  }
  HD void give(WCtx& c) const {
    c.pos = pos;
    c.lpos = lpos;
    c.err = err;
  }
};
struct EmitOut2 {
  u256 v, w;
  uint64_t pos, lpos;
  int err;
  HD void take(const WCtx& c) {
    pos = c.pos;
    lpos = c.lpos;
    err = c.err;
  }
  HD void give(WCtx& c) const {
    c.pos = pos;
    c.lpos = lpos;
    c.err = err;
  }
};

struct Gadgets {
  WCtx& c;
  const FpTables& T;
  HD Gadgets(WCtx& ctx) : c(ctx), T(*winv().T) {}

  HD u256 zero() const { return u256_zero(); }
  // Montgomery form v * 2^256 mod r of a small integer (v < 2^24) WITHOUT a table: v c - q r with c = 2^256 mod r and
  // q = floor(v mu / 2^32), mu = floor(c 2^32 / r) — never above floor(v c / r) and at most one below, so the difference is
  // below 2 r and one conditional subtraction finishes.  ~45 vector instructions instead of a 32-byte gather whose index is
  // data dependent: in the gadget chains such a load queues behind every store issued before it (one in-order vmcnt), and
  // the range checks make dozens of them per qmul.
  HD static u256 mont_small(uint32_t v) {
    constexpr uint32_t MU = 0x4a474626u;
    const uint32_t q = (uint32_t)(((uint64_t)v * MU) >> 32);
    u256 t;
    uint64_t a = 0, b = 0;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a += (uint64_t)v * FrParams::R1[i];   // words of v c
      b += (uint64_t)q * FrParams::P[i];    // words of q r
      const uint32_t aw = (uint32_t)a, bw = (uint32_t)b;
      const uint64_t d = (uint64_t)aw - bw - borrow;
      t.w[i] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
      a >>= 32;
      b >>= 32;
    }
    return lazy_canon<Fr>(t);
  }
  HD u256 small(uint32_t v) const { return mont_small(v < 260 ? v : 0); }
  // Montgomery form of a limb (< 2^L)
  HD u256 limb_mont(uint32_t v) const { return mont_small(v); }
  // value of the inverse cell of an is_zero block that starts at the current position (the cell is pos + 2)
  HD u256 inv_cell(const u256& x) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (winv().inv_cnt) {
      if (!c.in_window(c.pos + 2) || !c.in_rank(c.pos + 2)) return mont_one<Fr>();  // this window does not store the cell at all
      uint32_t i = atomicAdd(winv().inv_cnt, 1u);
      if (i < winv().inv_cap) {
        winv().inv_pos[i] = c.pos + 2;
        winv().inv_val[i] = x;
        return mont_one<Fr>();  // placeholder, overwritten by k_inv_fixup
      }
    }
#endif
    return mont_inv<Fr>(x);
  }
  HD u256 inv_or_one(const u256& x) const {
    if (u256_is_zero(x)) return mont_one<Fr>();
    return inv_cell(x);
  }

  // ================================================================ GateChip templates
  HD u256 g_add(const u256& a, const u256& b) {  // [a, b, 1, out]
    u256 o = fr_add(a, b);
    c.push(a, true); c.push(b, false); c.push(mont_one<Fr>(), false, true); c.push(o, false);
    return o;
  }
  HD u256 g_sub(const u256& a, const u256& b) {  // [out, b, 1, a]
    u256 o = fr_sub(a, b);
    c.push(o, true); c.push(b, false); c.push(mont_one<Fr>(), false, true); c.push(a, false);
    return o;
  }
  HD u256 g_neg(const u256& a) {  // [a, out, 1, 0]
    u256 o = fr_neg(a);
    c.push(a, true); c.push(o, false); c.push(mont_one<Fr>(), false, true); c.push(zero(), false, true);
    return o;
  }
  HD u256 g_mul(const u256& a, const u256& b) {  // [0, a, b, out]
    u256 o = fr_mul(a, b);
    c.push(zero(), true, true); c.push(a, false); c.push(b, false); c.push(o, false);
    return o;
  }
  HD u256 g_mul_add(const u256& a, const u256& b, const u256& cc) {  // [c, a, b, out]
    u256 o = fr_add(fr_mul(a, b), cc);
    c.push(cc, true); c.push(a, false); c.push(b, false); c.push(o, false);
    return o;
  }
  HD void g_assert_bit(const u256& x) {  // [0, x, x, x]
    c.push(zero(), true, true); c.push(x, false); c.push(x, false); c.push(x, false);
  }
  HD u256 g_not(const u256& a) { return g_sub(mont_one<Fr>(), a); }
  HD u256 g_and(const u256& a, const u256& b) { return g_mul(a, b); }
  HD u256 g_or(const u256& a, const u256& b) {  // [1-b, 1, b, 1, b, a, 1-b, out]
    u256 nb = fr_sub(mont_one<Fr>(), b);
    u256 o = fr_sub(fr_add(a, b), fr_mul(a, b));
    c.push(nb, true); c.push(mont_one<Fr>(), false, true); c.push(b, false); c.push(mont_one<Fr>(), false, true);
    c.push(b, true); c.push(a, false); c.push(nb, false); c.push(o, false);
    return o;
  }
  HD u256 g_select(const u256& a, const u256& b, const u256& s) {  // [a-b, 1, b, a, b, sel, a-b, out]
    u256 d = fr_sub(a, b);
    u256 o = u256_is_zero(s) ? b : (u256_eq(s, mont_one<Fr>()) ? a : fr_add(fr_mul(d, s), b));
    c.push(d, true); c.push(mont_one<Fr>(), false, true); c.push(b, false); c.push(a, false);
    c.push(b, true); c.push(s, false); c.push(d, false); c.push(o, false);
    return o;
  }
  // is_zero with the inverse cell supplied (WitnessFraction evaluated)
  HD u256 g_is_zero_inv(const u256& a, const u256& inv) {  // [z, a, inv, 1, 0, a, z, 0]
    u256 z = u256_is_zero(a) ? mont_one<Fr>() : zero();
    c.push(z, true); c.push(a, false); c.push(inv, false); c.push(mont_one<Fr>(), false, true);
    c.push(zero(), true, true); c.push(a, false); c.push(z, false); c.push(zero(), false, true);
    return z;
  }
  HD u256 g_is_zero(const u256& a) {
    if (c.skip(8)) {
      c.pos += 8;
      return u256_is_zero(a) ? mont_one<Fr>() : zero();
    }
    return g_is_zero_inv(a, inv_or_one(a));
  }
  HD u256 g_is_equal(const u256& a, const u256& b) {
    u256 d = g_sub(a, b);
    return g_is_zero(d);
  }
  HD u256 load_constant(const u256& v) {
    c.push(v, false, true);
    return v;
  }

  // ================================================================ RangeChip templates
  HD static uint32_t rc_cells(uint32_t bits, uint32_t L) {
    uint32_t k = (bits + L - 1) / L, rem = bits % L;
    return (k == 1 ? 0 : 1 + 3 * (k - 1)) + (rem >= 1 ? 4 : 0);
  }
  HD static uint32_t rc_lookups(uint32_t bits, uint32_t L) {
    uint32_t k = (bits + L - 1) / L, rem = bits % L;
    return k + (rem > 1 ? 1 : 0);
  }
  // range_check(a, bits); `ac` = canonical value of a.  Returns the last cell queued for lookup.
  static HDN EmitOut r_range_check__ool(WCtx cv, u256 a, u256 ac, uint32_t bits) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.r_range_check__body(a, ac, bits);
    o.take(cv);
    return o;
  }
  HD u256 r_range_check(const u256& a, const u256& ac, uint32_t bits) {
    EmitOut o = r_range_check__ool(c, a, ac, bits);
    o.give(c);
    return o.v;
  }
  HD u256 r_range_check__body(const u256& a, const u256& ac, uint32_t bits) {
    const uint32_t L = T.L, k = (bits + L - 1) / L, rem = bits % L;
    u256 last;
    if (k == 1) {
      c.lookup(a);  // (the source is the cell that holds `a`, somewhere earlier: not marked — copymap refuses such circuits)
      last = a;
    } else {
      // inner_product(limbs, [1, 2^L, 2^2L, ...]) starting with the constant 1:
      // [l0, l1, B1, s1, l2, B2, s2, ...] then the k limbs are queued for lookup
      // limbs are peeled off a running right shift (static register indexing only)
      const uint32_t lmask = (1u << L) - 1u;
      u256 rem = ac;
      u256 s = limb_mont(rem.w[0] & lmask);
      const uint64_t p0 = c.pos;  // limb 0 sits here, limb i at p0 + 1 + 3 (i - 1)
      c.push(s, true);
      for (uint32_t i = 1; i < k; i++) {
        rem = u256_shr_small(rem, L);
        u256 lm = limb_mont(rem.w[0] & lmask);
        s = fr_add(s, fr_mul(lm, T.pow2[i * L]));
        c.push(lm, false);
        c.push(T.pow2[i * L], false, true);
        c.push(s, i + 1 < k);
        last = lm;
      }
      rem = ac;
      for (uint32_t i = 0; i < k; i++) {
        c.mark_lookup_source(i ? p0 + 1 + 3 * (uint64_t)(i - 1) : p0);
        c.lookup(limb_mont(rem.w[0] & lmask));
        rem = u256_shr_small(rem, L);
      }
    }
    if (rem == 1) {
      g_assert_bit(last);
    } else if (rem > 1) {
      u256 chk = g_mul(last, T.pow2[L - rem]);
      c.mark_lookup_source(c.pos - 1);
      c.lookup(chk);
      last = chk;
    }
    return last;
  }
  HD u256 r_range_check_skippable(const u256& a, uint32_t bits) {
    uint32_t sz[2] = {rc_cells(bits, T.L), rc_lookups(bits, T.L)};
    if (c.skip2(sz)) {
      c.advance(sz);
      return zero();
    }
    return r_range_check(a, from_mont<Fr>(a), bits);
  }
  HD void r_check_less_than(const u256& a, const u256& b, uint32_t bits, bool b_const = false) {
    // [a + 2^n - b, b, 1, a + 2^n, -2^n, 1, a] gates 0,3 ; then range_check(first, bits)
    uint32_t sz[2] = {7 + rc_cells(bits, T.L), rc_lookups(bits, T.L)};
    if (c.skip2(sz)) {
      c.advance(sz);
      return;
    }
    r_check_less_than_emit(a, b, bits, b_const);
  }
  static HDN EmitOut r_check_less_than_emit__ool(WCtx cv, u256 a, u256 b, uint32_t bits, bool b_const) {
    Gadgets g(cv);
    g.r_check_less_than_emit__body(a, b, bits, b_const);
    EmitOut o;
    o.v = u256_zero();
    o.take(cv);
    return o;
  }
  HD void r_check_less_than_emit(const u256& a, const u256& b, uint32_t bits, bool b_const) {
    EmitOut o = r_check_less_than_emit__ool(c, a, b, bits, b_const);
    o.give(c);
  }
  HD void r_check_less_than_emit__body(const u256& a, const u256& b, uint32_t bits, bool b_const) {
    u256 sa = fr_add(T.pow2[bits], a), chk = fr_sub(sa, b);
    c.push(chk, true); c.push(b, false, b_const); c.push(mont_one<Fr>(), false, true);
    c.push(sa, true); c.push(fr_neg(T.pow2[bits]), false, true); c.push(mont_one<Fr>(), false, true); c.push(a, false);
    r_range_check(chk, from_mont<Fr>(chk), bits);
  }
  HD void r_check_big_less_than_safe(const u256& a, const u256& bound_mont, uint32_t bound_bits) {
    uint32_t rb = (bound_bits + T.L - 1) / T.L * T.L;
    r_range_check_skippable(a, rb);
    r_check_less_than(a, bound_mont, rb, true);  // the bound is a Constant cell
  }
  HD u256 r_is_less_than(const u256& a, const u256& b, uint32_t bits) {
    const uint32_t L = T.L, k = (bits + L - 1) / L, padded = k * L;
    u256 sa = fr_add(T.pow2[padded], a), sh = fr_sub(sa, b);
    c.push(sh, true); c.push(b, false); c.push(mont_one<Fr>(), false, true);
    c.push(sa, true); c.push(fr_neg(T.pow2[padded]), false, true); c.push(mont_one<Fr>(), false, true); c.push(a, false);
    u256 shc = from_mont<Fr>(sh);
    u256 last = r_range_check(sh, shc, padded + L);
    return g_is_zero_inv(last, inv_small_or_full(last, u256_extract(shc, padded, L)));
  }
  // inverse of a value known to equal the small canonical integer `v` (< 260), else Fermat
  HD u256 inv_small_or_full(const u256& x, uint32_t v) const {
    if (v == 0) return mont_one<Fr>();
    if (v < 260) return T.small_inv[v];
    return inv_cell(x);
  }

  // div_mod(a, 2^shift, a_bits): quotient/remainder by a power of two  (range.rs div_mod)
  HD void r_div_mod_pow2(const u256& a, uint32_t shift, uint32_t a_bits, u256& div, u256& rem) {
    u256 ac = from_mont<Fr>(a);
    u256 qc = u256_shr(ac, shift), rc = u256_low_bits(ac, shift);
    div = to_mont<Fr>(qc);
    rem = to_mont<Fr>(rc);
    c.push(rem, true); c.push(T.pow2[shift], false, true); c.push(div, false); c.push(a, false);
    // div < 2^a_bits / 2^shift + 1 ; rem < 2^shift
    u256 bound = fr_add(T.pow2[a_bits - shift], mont_one<Fr>());
    r_check_big_less_than_safe(div, bound, a_bits - shift + 1);
    r_check_big_less_than_safe(rem, T.pow2[shift], shift + 1);
  }
  // 256-bit division of canonical integers (BigUint div_mod_floor), b != 0: long division in base 2^32 (Knuth's algorithm D) in a
  // shape that indexes no register array at run time.  The divisor is shifted left until its top bit is bit 255 (bn), the dividend by
  // the same amount (a 512-bit value hi : lo); hi < bn is the first partial remainder, and each of lo's eight words, from the top,
  // yields one 32-bit quotient digit: the estimate floor(top two words of the remainder / top word of bn) is at most 2 too large for a
  // normalised divisor, which at most two add-backs correct.  Words that cannot produce a digit (remainder below the divisor's top
  // word) cost a shift.  The bit-serial loop this replaces took ~60 instructions for every BIT of the dividend — half of the
  // sequential chain of a distance's tail (qsqrt divides twice).
  HD static void divmod_u256(const u256& a, const u256& b, u256& q, u256& r) {
    q = u256_zero();
    const unsigned na = u256_bits(a), nb = u256_bits(b);
    if (nb == 0 || na < nb) {
      r = nb ? a : u256_zero();
      return;
    }
    const unsigned s = 256u - nb;
    const u256 bn = u256_shl(b, s);
    u256 lo = u256_shl(a, s);
    u256 R = s ? u256_shr(a, nb) : u256_zero();   // hi = a >> (256 - s)
    const uint32_t bt = bn.w[7];
    for (int i = 0; i < 8; i++) {
      const uint32_t R8 = R.w[7];
#pragma unroll
      for (int k = 7; k >= 1; k--) R.w[k] = R.w[k - 1];
      R.w[0] = lo.w[7];
#pragma unroll
      for (int k = 7; k >= 1; k--) lo.w[k] = lo.w[k - 1];
      lo.w[0] = 0;
      uint32_t qh = 0;
      const uint64_t top = ((uint64_t)R8 << 32) | R.w[7];
      if (top >= bt) {
        const uint64_t e = top / bt;
        qh = e > 0xffffffffull ? 0xffffffffu : (uint32_t)e;
        uint64_t carry = 0, borrow = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint64_t p = (uint64_t)qh * bn.w[k] + carry;
          carry = p >> 32;
          const uint64_t d = (uint64_t)R.w[k] - (uint32_t)p - borrow;
          R.w[k] = (uint32_t)d;
          borrow = (d >> 32) & 1;
        }
        int64_t t8 = (int64_t)(uint64_t)R8 - (int64_t)carry - (int64_t)borrow;
#pragma unroll
        for (int fix = 0; fix < 2; fix++) {
          if (t8 < 0) {
            t8 += (int64_t)u256_add(R, R, bn);
            qh--;
          }
        }
      }
#pragma unroll
      for (int k = 7; k >= 1; k--) q.w[k] = q.w[k - 1];
      q.w[0] = qh;
    }
    r = u256_shr(R, s);
  }
  static HDN EmitOut2 r_div_mod_var__ool(WCtx cv, u256 a, u256 b, uint32_t a_bits, uint32_t b_bits) {
    Gadgets g(cv);
    EmitOut2 o;
    g.r_div_mod_var__body(a, b, a_bits, b_bits, o.v, o.w);
    o.take(cv);
    return o;
  }
  HD void r_div_mod_var(const u256& a, const u256& b, uint32_t a_bits, uint32_t b_bits, u256& div, u256& rem) {
    EmitOut2 o = r_div_mod_var__ool(c, a, b, a_bits, b_bits);
    o.give(c);
    div = o.v;
    rem = o.w;
  }
  HD void r_div_mod_var__body(const u256& a, const u256& b, uint32_t a_bits, uint32_t b_bits, u256& div, u256& rem) {
    u256 ac = from_mont<Fr>(a), bc = from_mont<Fr>(b), qc, rc;
    if (u256_is_zero(bc)) {  // BigUint division by zero panics in the reference
      c.err = 1;
      qc = u256_zero();
      rc = u256_zero();
    } else {
      divmod_u256(ac, bc, qc, rc);
    }
    div = to_mont<Fr>(qc);
    rem = to_mont<Fr>(rc);
    c.push(rem, true); c.push(b, false); c.push(div, false); c.push(a, false);
    r_range_check_skippable(div, a_bits);
    r_check_less_than(rem, b, b_bits);
  }

  // ================================================================ FixedPointChip
  // value-only helpers (no cells)
  HD bool v_is_neg(const u256& ac) const { return u256_bits(ac) > 2 * T.P + 1; }  // a >= 2^(2P+1)
  HD u256 v_signed_div_scale(const u256& a) const {  // quotient of fixed_point.rs:974-996
    u256 ac = from_mont<Fr>(a);
    const uint32_t P = T.P;
    if (u256_bits(ac) > 253 || (u256_bits(ac) == 253 && !u256_is_zero(u256_low_bits(ac, 252)))) {
      u256 r = mod_p<Fr>(), aabs;
      u256_sub(aabs, r, ac);
      u256 cq = u256_shr(aabs, P);
      if (!u256_is_zero(u256_low_bits(aabs, P))) {
        u256 t;
        u256_add(t, cq, u256_from_u64(1));
        cq = t;
      }
      return fr_neg(to_mont<Fr>(cq));
    }
    return to_mont<Fr>(u256_shr(ac, P));
  }
  HD u256 v_qmul(const u256& a, const u256& b) const { return v_signed_div_scale(fr_mul(a, b)); }
  HD u256 v_qabs(const u256& a) const { return v_is_neg(from_mont<Fr>(a)) ? fr_neg(a) : a; }
  HD u256 v_qdiv(const u256& a, const u256& b, int& err) const {
    u256 ac = from_mont<Fr>(a), bc = from_mont<Fr>(b);
    bool sa = v_is_neg(ac), sb = v_is_neg(bc);
    u256 aa = sa ? fr_neg(a) : a, ba = sb ? fr_neg(b) : b;
    u256 num = from_mont<Fr>(fr_mul(aa, T.scale)), den = from_mont<Fr>(ba), q, r;
    if (u256_is_zero(den)) {
      err = 1;
      return u256_zero();
    }
    divmod_u256(num, den, q, r);
    u256 qm = to_mont<Fr>(q);
    return (sa != sb) ? fr_neg(qm) : qm;
  }

  // Skipped sub-gadgets never make a call: the window test and the value-only path are inline, only real
  // emission goes through the out-of-line *_emit generators.
  HD u256 fp_is_neg(const u256& a) {  // fixed_point.rs:523-539
    if (c.skip2(T.sz.is_neg)) {
      c.advance(T.sz.is_neg);
      return v_is_neg(from_mont<Fr>(a)) ? mont_one<Fr>() : zero();
    }
    return fp_is_neg_emit(a);
  }
  static HDN EmitOut fp_is_neg_emit__ool(WCtx cv, u256 a) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_is_neg_emit__body(a);
    o.take(cv);
    return o;
  }
  HD u256 fp_is_neg_emit(const u256& a) {
    EmitOut o = fp_is_neg_emit__ool(c, a);
    o.give(c);
    return o.v;
  }
  HD u256 fp_is_neg_emit__body(const u256& a) {
    u256 div, rem;
    r_div_mod_pow2(a, 2 * T.P + 1, 254, div, rem);
    u256 is_pos = g_is_zero(div);
    return g_not(is_pos);
  }
  HD u256 fp_qabs(const u256& a) {  // :511-521
    if (c.skip2(T.sz.qabs)) {
      c.advance(T.sz.qabs);
      return v_qabs(a);
    }
    return fp_qabs_emit(a);
  }
  static HDN EmitOut fp_qabs_emit__ool(WCtx cv, u256 a) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_qabs_emit__body(a);
    o.take(cv);
    return o;
  }
  HD u256 fp_qabs_emit(const u256& a) {
    EmitOut o = fp_qabs_emit__ool(c, a);
    o.give(c);
    return o.v;
  }
  HD u256 fp_qabs_emit__body(const u256& a) {
    u256 rev = g_neg(a);
    u256 n = fp_is_neg(a);
    return g_select(rev, a, n);
  }
  HD u256 fp_cond_neg(const u256& a, const u256& flag) {  // :541-556
    u256 na = g_neg(a);
    return g_select(na, a, flag);
  }
  HD u256 fp_signed_div_scale(const u256& a) {  // :974-1016, returns the quotient
    if (c.skip2(T.sz.sds)) {
      c.advance(T.sz.sds);
      return v_signed_div_scale(a);
    }
    return fp_signed_div_scale_emit(a);
  }
  static HDN EmitOut fp_signed_div_scale_emit__ool(WCtx cv, u256 a) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_signed_div_scale_emit__body(a);
    o.take(cv);
    return o;
  }
  HD u256 fp_signed_div_scale_emit(const u256& a) {
    EmitOut o = fp_signed_div_scale_emit__ool(c, a);
    o.give(c);
    return o.v;
  }
  HD u256 fp_signed_div_scale_emit__body(const u256& a) {
    const uint32_t P = T.P;
    u256 ac = from_mont<Fr>(a), div, rem;
    bool neg = u256_bits(ac) > 253 || (u256_bits(ac) == 253 && !u256_is_zero(u256_low_bits(ac, 252)));  // a > 2^252
    if (neg) {
      u256 r = mod_p<Fr>(), aabs, one = u256_from_u64(1);
      u256_sub(aabs, r, ac);
      u256 cq = u256_shr(aabs, P), low = u256_low_bits(aabs, P), remc = u256_zero();
      if (!u256_is_zero(low)) {
        u256 t;
        u256_add(t, cq, one);
        cq = t;
        u256_sub(remc, u256_shl(one, P), low);  // 2^P * ceil - |a|
      }
      div = fr_neg(to_mont<Fr>(cq));
      rem = to_mont<Fr>(remc);
    } else {
      div = to_mont<Fr>(u256_shr(ac, P));
      rem = to_mont<Fr>(u256_low_bits(ac, P));
    }
    c.push(rem, true); c.push(T.scale, false, true); c.push(div, false); c.push(a, false);
    r_check_big_less_than_safe(rem, T.pow2[P], P + 1);
    u256 dabs = fp_qabs(div);
    r_check_big_less_than_safe(dabs, T.pow2[3 * P], 3 * P + 1);
    return div;
  }
  HD u256 fp_qmul(const u256& a, const u256& b) {  // :588-604
    if (c.skip2(T.sz.qmul)) {
      c.advance(T.sz.qmul);
      return v_qmul(a, b);
    }
    return fp_qmul_emit(a, b);
  }
  static HDN EmitOut fp_qmul_emit__ool(WCtx cv, u256 a, u256 b) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_qmul_emit__body(a, b);
    o.take(cv);
    return o;
  }
  HD u256 fp_qmul_emit(const u256& a, const u256& b) {
    EmitOut o = fp_qmul_emit__ool(c, a, b);
    o.give(c);
    return o.v;
  }
  HD u256 fp_qmul_emit__body(const u256& a, const u256& b) {
    u256 ab = g_mul(a, b);
    return fp_signed_div_scale(ab);
  }
  HD u256 fp_bit_xor(const u256& a, const u256& b) {  // :797-815
    u256 a2 = g_add(zero(), a);
    u256 b2 = g_add(zero(), b);
    g_assert_bit(a2);
    g_assert_bit(b2);
    u256 ab = g_add(a2, b2);
    u256 one = g_add(mont_one<Fr>(), zero());
    u256 d = g_sub(ab, one);
    // d in {-1, 0, 1}
    // d in {-1, 0, 1} for the bits the gadget asserts; anything else (a caller handing in non-bits: the assertions then fail) takes the
    // general inverse
    const u256 m1 = fr_neg(mont_one<Fr>());
    u256 inv = u256_is_zero(d) ? mont_one<Fr>() : (u256_eq(d, mont_one<Fr>()) ? mont_one<Fr>() : (u256_eq(d, m1) ? m1 : inv_or_one(d)));
    return g_is_zero_inv(d, inv);
  }
  HD u256 fp_qdiv(const u256& a, const u256& b) {  // :631-656
    if (c.skip2(T.sz.qdiv)) {
      c.advance(T.sz.qdiv);
      return v_qdiv(a, b, c.err);
    }
    return fp_qdiv_emit(a, b);
  }
  static HDN EmitOut fp_qdiv_emit__ool(WCtx cv, u256 a, u256 b) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_qdiv_emit__body(a, b);
    o.take(cv);
    return o;
  }
  HD u256 fp_qdiv_emit(const u256& a, const u256& b) {
    EmitOut o = fp_qdiv_emit__ool(c, a, b);
    o.give(c);
    return o.v;
  }
  HD u256 fp_qdiv_emit__body(const u256& a, const u256& b) {
    const uint32_t P = T.P;
    u256 sa = fp_is_neg(a);
    u256 sb = fp_is_neg(b);
    u256 aa = fp_qabs(a);
    u256 ba = fp_qabs(b);
    u256 ar = g_mul(aa, T.scale);
    u256 q, r;
    r_div_mod_var(ar, ba, 4 * P, 2 * P, q, r);
    u256 sx = fp_bit_xor(sa, sb);
    return fp_cond_neg(q, sx);
  }
  HD u256 fp_qmin(const u256& a, const u256& b) {  // :936-952
    if (c.skip2(T.sz.qmin)) {
      c.advance(T.sz.qmin);
      return v_is_neg(from_mont<Fr>(fr_sub(a, b))) ? a : b;
    }
    u256 amb = g_sub(a, b);
    u256 s = fp_is_neg(amb);
    return g_select(a, b, s);
  }
  HD u256 fp_qmax(const u256& a, const u256& b) {  // :918-934
    u256 amb = g_sub(a, b);
    u256 s = fp_is_neg(amb);
    return g_select(b, a, s);
  }
  template <int M>
  HD u256 fp_polynomial(const u256& x, const u256 (&coef)[M], const uint32_t (&sz)[2]) {  // :658-686
    if (c.skip2(sz)) {
      c.advance(sz);
      u256 y = zero();
      for (int i = 0; i < M; i++) {
        y = fr_add(y, coef[i]);
        if (i < M - 1) y = v_qmul(x, y);
      }
      return y;
    }
    u256 result = g_add(x, zero());  // dead qadd(x, 0)
    u256 last = zero();
    for (int i = 0; i < M; i++) {
      u256 y_add = fr_add(last, coef[i]);  // qadd(last_y, Constant(coef)): [last_y, coef, 1, out]
      c.push(last, true, i == 0); c.push(coef[i], false, true); c.push(mont_one<Fr>(), false, true); c.push(y_add, false);
      if (i < M - 1) last = fp_qmul(x, y_add);
      else result = y_add;
    }
    return result;
  }
  // GateChip::select_from_idx over `n` cells with idx known to be the small integer `idx_small`
  // (or >= n when out of range): idx_to_indicator (v0.3) + select_by_indicator
  template <class CellFn>
  HD u256 g_select_from_idx(uint32_t n, const u256& idx, uint64_t idx_small, CellFn&& cell, bool cells_const = false) {
    // indicator i: i == 0 -> unrolled is_zero(idx); else is_equal(idx, Constant(i))
    for (uint32_t i = 0; i < n; i++) {
      int64_t diff = (int64_t)idx_small - (int64_t)i;
      if (i == 0) {
        if (c.skip(8)) c.pos += 8;
        else g_is_zero_inv(idx, signed_small_inv(idx, diff));
      } else {
        if (c.skip(12)) {
          c.pos += 12;
        } else {
          u256 ci = mont_small(i);
          u256 d = fr_sub(idx, ci);  // is_equal(idx, Constant(i)) = sub [d, i, 1, idx] + is_zero
          c.push(d, true); c.push(ci, false, true); c.push(mont_one<Fr>(), false, true); c.push(idx, false);
          g_is_zero_inv(d, signed_small_inv(d, diff));
        }
      }
    }
    // select_by_indicator: [0, a0, ind0, s0, a1, ind1, s1, ...]
    u256 s = zero();
    c.push(zero(), n > 0, true);
    for (uint32_t i = 0; i < n; i++) {
      u256 ai = cell(i);
      bool hit = (idx_small == i);
      if (hit) s = ai;
      c.push(ai, false, cells_const);
      c.push(hit ? mont_one<Fr>() : zero(), false);
      c.push(s, i + 1 < n);
    }
    return s;
  }
  HD u256 signed_small_inv(const u256& x, int64_t v) const {
    if (v == 0) return mont_one<Fr>();
    if (v > 0 && v < 260) return T.small_inv[v];
    if (v < 0 && v > -260) return fr_neg(T.small_inv[-v]);
    return inv_cell(x);
  }
  HD void fp_check_power_of_two(const u256& p2, const u256& e, uint64_t e_small) {  // :688-708
    if (c.skip2(T.sz.cpow2)) {
      c.advance(T.sz.cpow2);
      return;
    }
    fp_check_power_of_two_emit(p2, e, e_small);
  }
  static HDN EmitOut fp_check_power_of_two_emit__ool(WCtx cv, u256 p2, u256 e, uint64_t e_small) {
    Gadgets g(cv);
    g.fp_check_power_of_two_emit__body(p2, e, e_small);
    EmitOut o;
    o.v = u256_zero();
    o.take(cv);
    return o;
  }
  HD void fp_check_power_of_two_emit(const u256& p2, const u256& e, uint64_t e_small) {
    EmitOut o = fp_check_power_of_two_emit__ool(c, p2, e, e_small);
    o.give(c);
  }
  HD void fp_check_power_of_two_emit__body(const u256& p2, const u256& e, uint64_t e_small) {
    const uint32_t nb = 2 * T.P;
    u256 pc = from_mont<Fr>(p2);
    // num_to_bits: inner_product(bits, pow2) starting with constant 1, then nb assert_bit
    uint32_t nset = 0;
    {
      u256 rem = pc;  // bits are peeled off a running right shift (static register indexing only)
      u256 s = (rem.w[0] & 1u) ? mont_one<Fr>() : zero();
      nset += rem.w[0] & 1u;
      c.push(s, true);
      for (uint32_t i = 1; i < nb; i++) {
        rem = u256_shr_small(rem, 1);
        uint32_t bit = rem.w[0] & 1u;
        nset += bit;
        if (bit) s = fr_add(s, T.pow2[i]);
        c.push(bit ? mont_one<Fr>() : zero(), false);
        c.push(T.pow2[i], false, true);
        c.push(s, i + 1 < nb);
      }
      rem = pc;
      for (uint32_t i = 0; i < nb; i++) {
        g_assert_bit((rem.w[0] & 1u) ? mont_one<Fr>() : zero());
        rem = u256_shr_small(rem, 1);
      }
    }
    // sum(bits): [b0, b1, 1, s1, b2, 1, s2, ...]
    {
      u256 rem = pc;
      uint32_t run = rem.w[0] & 1u;
      c.push(run ? mont_one<Fr>() : zero(), nb > 1);
      for (uint32_t i = 1; i < nb; i++) {
        rem = u256_shr_small(rem, 1);
        uint32_t bit = rem.w[0] & 1u;
        run += bit;
        c.push(bit ? mont_one<Fr>() : zero(), false);
        c.push(mont_one<Fr>(), false, true);
        c.push(small(run), i + 1 < nb);
      }
    }
    u256 sum = small(nset);
    u256 sm1 = g_sub(sum, mont_one<Fr>());
    g_is_zero_inv(sm1, signed_small_inv(sm1, (int64_t)nset - 1));
    u256 walk = pc;  // the cell callback is invoked for i = 0, 1, 2, ... in order
    u256 bit = g_select_from_idx(nb, e, e_small, [&](uint32_t) {
      u256 v = (walk.w[0] & 1u) ? mont_one<Fr>() : u256_zero();
      walk = u256_shr_small(walk, 1);
      return v;
    });
    u256 bm1 = g_sub(bit, mont_one<Fr>());
    g_is_zero_inv(bm1, u256_is_zero(bm1) ? mont_one<Fr>() : fr_neg(mont_one<Fr>()));
  }
  HD u256 v_qlog2(const u256& a) {
    u256 ac = from_mont<Fr>(a);
    uint32_t nd = u256_is_zero(ac) ? 1 : u256_bits(ac) - 1;
    int64_t shift = (int64_t)T.P + 2 - ((int64_t)nd + 1);
    uint32_t sa = (uint32_t)(shift < 0 ? -shift : shift);
    if (sa >= 254) sa = 0;
    u256 a_norm = shift < 0 ? to_mont<Fr>(u256_shr(ac, sa)) : fr_mul(a, T.pow2[sa]);
    u256 ln = fp_polynomial<15>(a_norm, T.log_poly, T.sz.poly15);  // skipped => value path
    u256 shm = shift < 0 ? fr_neg(small_or_mont((uint64_t)(-shift))) : small_or_mont((uint64_t)shift);
    return fr_add(ln, fr_mul(fr_neg(shm), T.scale));
  }
  HD u256 small_or_mont(uint64_t v) const { return v < (1u << 24) ? mont_small((uint32_t)v) : to_mont<Fr>(u256_from_u64(v)); }
  HD u256 fp_qlog2(const u256& a) {  // :736-795
    if (c.skip2(T.sz.qlog2)) {
      uint64_t p0 = c.pos, l0 = c.lpos;
      u256 v = v_qlog2(a);
      c.pos = p0 + T.sz.qlog2[0];
      c.lpos = l0 + T.sz.qlog2[1];
      return v;
    }
    return fp_qlog2_emit(a);
  }
  static HDN EmitOut fp_qlog2_emit__ool(WCtx cv, u256 a) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_qlog2_emit__body(a);
    o.take(cv);
    return o;
  }
  HD u256 fp_qlog2_emit(const u256& a) {
    EmitOut o = fp_qlog2_emit__ool(c, a);
    o.give(c);
    return o.v;
  }
  HD u256 fp_qlog2_emit__body(const u256& a) {
    const uint32_t P = T.P;
    u256 a_assigned = g_add(a, zero());
    u256 is_neg = fp_is_neg(a);
    u256 is_zero = g_is_zero(a_assigned);
    g_or(is_neg, is_zero);
    u256 ac = from_mont<Fr>(a);
    uint32_t nd = u256_is_zero(ac) ? 1 : u256_bits(ac) - 1;
    u256 pow1w = g_add(T.pow2[nd < 254 ? nd : 0], zero());
    u256 exp1 = g_add(small_or_mont(nd), zero());
    fp_check_power_of_two(pow1w, exp1, nd);
    u256 pow2w = g_mul(pow1w, T.small[2]);
    u256 exp2 = g_add(exp1, mont_one<Fr>());
    fp_check_power_of_two(pow2w, exp2, (uint64_t)nd + 1);
    u256 lt2 = r_is_less_than(a, pow2w, 2 * P);
    u256 gt1 = r_is_less_than(pow1w, a, 2 * P);
    u256 eq1 = g_is_equal(a, pow1w);
    u256 ge1 = g_or(eq1, gt1);
    g_and(lt2, ge1);
    u256 shift = g_sub(small_or_mont((uint64_t)P + 2), exp2);
    int64_t shv = (int64_t)P + 2 - ((int64_t)nd + 1);
    u256 shift_neg = fp_is_neg(shift);
    u256 shift_abs = fp_qabs(shift);
    uint32_t sa = (uint32_t)(shv < 0 ? -shv : shv);
    if (sa >= 254) {  // Vec index out of range panics in the reference
      c.err = 1;
      sa = 0;
    }
    u256 spw = g_add(T.pow2[sa], zero());
    fp_check_power_of_two(spw, shift_abs, sa);
    u256 a_ls = g_mul(a, spw);
    u256 a_rs, rr;
    r_div_mod_var(a, spw, 2 * P, P + 1, a_rs, rr);
    u256 a_norm = g_select(a_rs, a_ls, shift_neg);
    u256 log_norm = fp_polynomial<15>(a_norm, T.log_poly, T.sz.poly15);
    u256 log_shift = g_neg(shift);
    u256 lsq = g_mul(log_shift, T.scale);
    return g_add(log_norm, lsq);
  }
  HD u256 v_qexp2(const u256& a) {
    u256 a_abs = v_qabs(a);
    u256 ac = from_mont<Fr>(a_abs);
    u256 ipc = u256_shr(ac, T.P);
    uint32_t ip = u256_bits(ipc) > 16 ? 0xffffu : ipc.w[0];
    u256 fpart = to_mont<Fr>(u256_low_bits(ac, T.P));
    u256 yf = fp_polynomial<13>(fpart, T.exp2_poly, T.sz.poly13);
    u256 res_pos = ip < 254 ? fr_mul(T.pow2[ip], yf) : zero();
    if (u256_is_zero(res_pos)) c.err = 1;  // the emission path always evaluates qdiv(2^P, res_pos)
    if (v_is_neg(from_mont<Fr>(a))) return v_qdiv(T.scale, res_pos, c.err);
    return res_pos;
  }
  HD u256 fp_qexp2(const u256& a) {  // :710-734
    if (c.skip2(T.sz.qexp2)) {
      uint64_t p0 = c.pos, l0 = c.lpos;
      u256 v = v_qexp2(a);
      c.pos = p0 + T.sz.qexp2[0];
      c.lpos = l0 + T.sz.qexp2[1];
      return v;
    }
    return fp_qexp2_emit(a);
  }
  static HDN EmitOut fp_qexp2_emit__ool(WCtx cv, u256 a) {
    Gadgets g(cv);
    EmitOut o;
    o.v = g.fp_qexp2_emit__body(a);
    o.take(cv);
    return o;
  }
  HD u256 fp_qexp2_emit(const u256& a) {
    EmitOut o = fp_qexp2_emit__ool(c, a);
    o.give(c);
    return o.v;
  }
  HD u256 fp_qexp2_emit__body(const u256& a) {
    const uint32_t P = T.P;
    u256 a_abs = fp_qabs(a);
    u256 ip, fpart;
    r_div_mod_pow2(a_abs, P, 2 * P, ip, fpart);
    u256 ipc = from_mont<Fr>(ip);
    uint64_t ip_small = u256_bits(ipc) > 32 ? 0xffffffffull : ipc.w[0];
    u256 ip2 = g_select_from_idx(254, ip, ip_small, [&](uint32_t i) { return T.pow2[i]; }, true);  // Constant(pow_of_two[i]) cells
    u256 yf = fp_polynomial<13>(fpart, T.exp2_poly, T.sz.poly13);
    u256 res_pos = g_mul(ip2, yf);
    u256 res_neg = fp_qdiv(T.scale, res_pos);
    u256 n = fp_is_neg(a);
    return g_select(res_neg, res_pos, n);
  }
  HD u256 fp_qlog(const u256& a) {  // :954-964
    u256 l2e = load_constant(T.c_log2e);
    u256 l2a = fp_qlog2(a);
    return fp_qdiv(l2a, l2e);
  }
  HD u256 fp_qexp(const u256& a) {  // :876-886
    u256 ln2 = load_constant(T.c_ln2);
    u256 x1 = fp_qdiv(a, ln2);
    return fp_qexp2(x1);
  }
  HD u256 fp_qpow(const u256& x, const u256& e) {  // :441-456
    u256 lx = fp_qlog(x);
    u256 al = fp_qmul(e, lx);
    return fp_qexp(al);
  }
  HD u256 fp_qsqrt(const u256& x) {  // :966-972
    u256 half = load_constant(T.c_half);
    return fp_qpow(x, half);
  }
  // ---- the rest of FixedPointInstructions: not reached from DistanceChip / VectorDBChip (examples/fixed_point.rs calls qsin); plain
  // compositions of the generators above, which bring their own value-only paths
  HD u256 fp_sign(const u256& a) {  // :558-569: the field elements 1 / -1
    u256 neg_one = g_neg(mont_one<Fr>());
    u256 n = fp_is_neg(a);
    return g_select(neg_one, mont_one<Fr>(), n);
  }
  HD u256 fp_clip(const u256& a) {  // :571-586: |a| mod max_value = 2^(2P), the sign restored
    u256 sgn = fp_is_neg(a);
    u256 aa = fp_qabs(a);
    u256 div, rem;
    r_div_mod_pow2(aa, 2 * T.P, 254, div, rem);
    return fp_cond_neg(rem, sgn);
  }
  HD u256 fp_qmod(const u256& a, const u256& b) {  // :606-629; b positive (assert_is_const(b_sign, 0) pushes no cell)
    u256 sa = fp_is_neg(a);
    (void)fp_is_neg(b);
    u256 aa = fp_qabs(a);
    u256 q, r;
    r_div_mod_var(aa, b, 4 * T.P, 2 * T.P, q, r);
    u256 comp = g_sub(b, r);
    return g_select(comp, r, sa);
  }
  HD u256 fp_qsin(const u256& a) {  // :817-841
    u256 aa = fp_qabs(a);
    u256 sa = fp_is_neg(a);
    u256 a_mod = fp_qmod(aa, T.c_two_pi);
    u256 a_mpi = g_sub(a_mod, T.c_pi);
    u256 lower = fp_is_neg(a_mpi);
    u256 s_mod = fp_polynomial<15>(a_mod, T.sin_poly, T.sz.poly15);
    u256 s_rev = fp_polynomial<15>(a_mpi, T.sin_poly, T.sz.poly15);
    u256 s_mpi = g_neg(s_rev);   // -sin(a - pi) for pi <= a < 2 pi
    u256 s_abs = g_select(s_mod, s_mpi, lower);
    return fp_cond_neg(s_abs, sa);
  }
  HD u256 fp_qcos(const u256& a) {  // :843-852
    u256 hp = load_constant(T.c_half_pi);
    return fp_qsin(g_add(a, hp));
  }
  HD u256 fp_qtan(const u256& a) {  // :383-393
    u256 s = fp_qsin(a);
    u256 co = fp_qcos(a);
    return fp_qdiv(s, co);
  }
  HD u256 fp_sinh_cosh(const u256& a, bool cosh) {  // :888-916
    u256 ea = fp_qexp(a);
    u256 na = g_neg(a);
    u256 ena = fp_qexp(na);
    u256 nume = cosh ? g_add(ea, ena) : g_sub(ea, ena);
    u256 two = load_constant(T.c_two);
    return fp_qdiv(nume, two);
  }
  HD u256 fp_qtanh(const u256& a) {  // :407-417
    u256 s = fp_sinh_cosh(a, false);
    u256 co = fp_sinh_cosh(a, true);
    return fp_qdiv(s, co);
  }
};

// Host-side sizing: runs the generators in counting mode and fills T.sz bottom-up.
inline void compute_sizes(FpTables& T) {
  auto measure = [&](auto&& fn, uint32_t out[2]) {
    WCtx c{};
    c.count_only = true;
    c.hi = ~0ull;
    h_winv = WInv{};
    h_winv.rhi = ~0ull;
    h_winv.rlhi = ~0ull;
    h_winv.T = &T;
    Gadgets g(c);
    fn(g);
    out[0] = (uint32_t)c.pos;
    out[1] = (uint32_t)c.lpos;
  };
  u256 x = T.c_one_q, y = T.c_half;
  measure([&](Gadgets& g) { g.fp_is_neg(x); }, T.sz.is_neg);
  measure([&](Gadgets& g) { g.fp_qabs(x); }, T.sz.qabs);
  measure([&](Gadgets& g) { g.fp_signed_div_scale(x); }, T.sz.sds);
  measure([&](Gadgets& g) { g.fp_qmul(x, y); }, T.sz.qmul);
  measure([&](Gadgets& g) { g.fp_qdiv(x, y); }, T.sz.qdiv);
  measure([&](Gadgets& g) { g.fp_qmin(x, y); }, T.sz.qmin);
  measure([&](Gadgets& g) { g.fp_check_power_of_two(T.pow2[3], T.small[3], 3); }, T.sz.cpow2);
  measure([&](Gadgets& g) { g.fp_polynomial<13>(x, T.exp2_poly, T.sz.poly13); }, T.sz.poly13);
  measure([&](Gadgets& g) { g.fp_polynomial<15>(x, T.log_poly, T.sz.poly15); }, T.sz.poly15);
  measure([&](Gadgets& g) { g.fp_qlog2(x); }, T.sz.qlog2);
  measure([&](Gadgets& g) { g.fp_qexp2(x); }, T.sz.qexp2);
  measure([&](Gadgets& g) { g.fp_qlog(x); }, T.sz.qlog);
  measure([&](Gadgets& g) { g.fp_qexp(x); }, T.sz.qexp);
  measure([&](Gadgets& g) { g.fp_qsqrt(x); }, T.sz.qsqrt);
}

// One FixedPointInstructions call by number (vdb_wit_fp_op*; the numbering of include/vdb.h): what the reference's circuits reach
// through the trait, one operation at a time.  b is ignored by the unary ones.
enum FpOp {
  FP_QADD = 0, FP_QSUB, FP_QMUL, FP_QDIV, FP_NEG, FP_QABS, FP_IS_NEG, FP_QMIN, FP_QSQRT, FP_QLOG2, FP_QEXP2, FP_QLOG, FP_QEXP, FP_QPOW, FP_BIT_XOR,
  FP_COND_NEG, FP_SIGNED_DIV_SCALE, FP_QMAX, FP_SIGN, FP_CLIP, FP_QMOD, FP_QSIN, FP_QCOS, FP_QTAN, FP_QSINH, FP_QCOSH, FP_QTANH, FP_OP_COUNT
};
// (every case is a function of its own, never inlined: one kernel body holding all 27 generators took the compiler minutes)
#define VDB_FP_THUNK1(name, expr) \
  __host__ __device__ __noinline__ inline u256 fp_thunk_##name(Gadgets& g, const u256& a) { return expr; }
#define VDB_FP_THUNK2(name, expr) \
  __host__ __device__ __noinline__ inline u256 fp_thunk_##name(Gadgets& g, const u256& a, const u256& b) { return expr; }
VDB_FP_THUNK2(qadd, g.g_add(a, b))   // qadd / qsub are the gate's add / sub (fixed_point.rs:475-509)
VDB_FP_THUNK2(qsub, g.g_sub(a, b))
VDB_FP_THUNK2(qmul, g.fp_qmul(a, b))
VDB_FP_THUNK2(qdiv, g.fp_qdiv(a, b))
VDB_FP_THUNK1(neg, g.g_neg(a))
VDB_FP_THUNK1(qabs, g.fp_qabs(a))
VDB_FP_THUNK1(is_neg, g.fp_is_neg(a))
VDB_FP_THUNK2(qmin, g.fp_qmin(a, b))
VDB_FP_THUNK1(qsqrt, g.fp_qsqrt(a))
VDB_FP_THUNK1(qlog2, g.fp_qlog2(a))
VDB_FP_THUNK1(qexp2, g.fp_qexp2(a))
VDB_FP_THUNK1(qlog, g.fp_qlog(a))
VDB_FP_THUNK1(qexp, g.fp_qexp(a))
VDB_FP_THUNK2(qpow, g.fp_qpow(a, b))
VDB_FP_THUNK2(bit_xor, g.fp_bit_xor(a, b))
VDB_FP_THUNK2(cond_neg, g.fp_cond_neg(a, b))
VDB_FP_THUNK1(sds, g.fp_signed_div_scale(a))
VDB_FP_THUNK2(qmax, g.fp_qmax(a, b))
VDB_FP_THUNK1(sign, g.fp_sign(a))
VDB_FP_THUNK1(clip, g.fp_clip(a))
VDB_FP_THUNK2(qmod, g.fp_qmod(a, b))
VDB_FP_THUNK1(qsin, g.fp_qsin(a))
#undef VDB_FP_THUNK1
#undef VDB_FP_THUNK2
HD inline u256 fp_op_apply(Gadgets& g, int op, const u256& a, const u256& b) {
  switch (op) {
    case FP_QADD: return fp_thunk_qadd(g, a, b);
    case FP_QSUB: return fp_thunk_qsub(g, a, b);
    case FP_QMUL: return fp_thunk_qmul(g, a, b);
    case FP_QDIV: return fp_thunk_qdiv(g, a, b);
    case FP_NEG: return fp_thunk_neg(g, a);
    case FP_QABS: return fp_thunk_qabs(g, a);
    case FP_IS_NEG: return fp_thunk_is_neg(g, a);
    case FP_QMIN: return fp_thunk_qmin(g, a, b);
    case FP_QSQRT: return fp_thunk_qsqrt(g, a);
    case FP_QLOG2: return fp_thunk_qlog2(g, a);
    case FP_QEXP2: return fp_thunk_qexp2(g, a);
    case FP_QLOG: return fp_thunk_qlog(g, a);
    case FP_QEXP: return fp_thunk_qexp(g, a);
    case FP_QPOW: return fp_thunk_qpow(g, a, b);
    case FP_BIT_XOR: return fp_thunk_bit_xor(g, a, b);
    case FP_COND_NEG: return fp_thunk_cond_neg(g, a, b);
    case FP_SIGNED_DIV_SCALE: return fp_thunk_sds(g, a);
    case FP_QMAX: return fp_thunk_qmax(g, a, b);
    case FP_SIGN: return fp_thunk_sign(g, a);
    case FP_CLIP: return fp_thunk_clip(g, a);
    case FP_QMOD: return fp_thunk_qmod(g, a, b);
    case FP_QSIN: return fp_thunk_qsin(g, a);
    case FP_QCOS: {  // :843-852
      u256 hp = g.load_constant(g.T.c_half_pi);
      u256 t = fp_thunk_qadd(g, a, hp);
      return fp_thunk_qsin(g, t);
    }
    case FP_QTAN: {  // :383-393
      u256 s = fp_thunk_qsin(g, a);
      u256 hp = g.load_constant(g.T.c_half_pi);
      u256 t = fp_thunk_qadd(g, a, hp);
      u256 co = fp_thunk_qsin(g, t);
      return fp_thunk_qdiv(g, s, co);
    }
    case FP_QSINH:
    case FP_QCOSH:
    case FP_QTANH: {  // :888-916, :407-417
      u256 num[2];
      for (int pass = (op == FP_QCOSH ? 1 : 0); pass <= (op == FP_QSINH ? 0 : 1); pass++) {
        u256 ea = fp_thunk_qexp(g, a);
        u256 na = fp_thunk_neg(g, a);
        u256 ena = fp_thunk_qexp(g, na);
        u256 nume = pass ? fp_thunk_qadd(g, ea, ena) : fp_thunk_qsub(g, ea, ena);
        u256 two = g.load_constant(g.T.c_two);
        num[pass] = fp_thunk_qdiv(g, nume, two);
      }
      return op == FP_QSINH ? num[0] : (op == FP_QCOSH ? num[1] : fp_thunk_qdiv(g, num[0], num[1]));
    }
  }
  return u256_zero();
}
// cells / lookup cells of one call (data independent: a counting run on the host)
inline void fp_op_size(FpTables& T, int op, uint32_t out[2]) {
  WCtx c{};
  c.count_only = true;
  c.hi = ~0ull;
  h_winv = WInv{};
  h_winv.rhi = ~0ull;
  h_winv.rlhi = ~0ull;
  h_winv.T = &T;
  Gadgets g(c);
  fp_op_apply(g, op, T.c_one_q, op == FP_BIT_XOR || op == FP_COND_NEG ? T.one : T.c_half);
  out[0] = (uint32_t)c.pos;
  out[1] = (uint32_t)c.lpos;
}

}  // namespace vdb
