// Batched radix-2 NTT over BN254 Fr for gfx950.
//
// Replaces halo2 `arithmetic::best_fft` and `EvaluationDomain::{lagrange_to_coeff, coeff_to_extended}`
// (third-party halo2-axiom, reached from /root/reference/src/scaffold/mod.rs:296).
// Semantics: X[i] = sum_j a[j] * omega^(i*j), natural order in and out.
//
// Structure (MI355X-first, not a port of the CPU recursion):
//   n = m_0 * m_1 * ... * m_{L-1}, every m_l = 2^S_l <= 512.  Pass l performs, for every residue of
//   the other digits, a size-m_l DFT along digit l inside LDS (decimation in frequency, 2 butterflies
//   per thread per stage), multiplies by the inter-pass twiddle omega_{N_l}^(q*i) and writes back.
//   A workgroup owns a tile of m_l x G elements (G*32 B contiguous = whole 128-B lines in HBM).
//   The last pass reads contiguous rows and scatters to the digit-reversed natural-order position,
//   G consecutive outputs per row group so stores are again whole lines.
//   LDS holds the tile as two 16-byte planes with a one-element row pad: conflict-free ds_read_b128 /
//   ds_write_b128 both along rows and along columns.
// Roofline: 64 B/element algorithmic HBM traffic per transform; the kernel is integer-ALU bound
// (one 254-bit Montgomery product per butterfly), see DESIGN.md.
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "limb9.hpp"

namespace vdb {

#define NTT_THREADS 256
#define NTT_TILE 1024
#define NTT_MAX_PASSES 4
#define NTT_DEFAULT_COLGROUP 0x7fffffffu  // tile-major over all columns of a launch (measured equal to column-major within noise; 4x less twiddle refetch)

struct NttPass {
  uint32_t log_n, S, log_inner, logG;
  uint32_t nprev;                 // last pass: number of previous passes
  uint32_t prevS[NTT_MAX_PASSES]; // their sizes (S_0 .. S_{L-2})
  uint32_t first;                 // this is pass 0 (input staging rules apply)
  uint32_t coset;                 // 1: multiply input element e by zeta^(e mod 3); 2: by s zeta^(e mod 3) for a scalar s (zeta0 = 32 s);
                                  // 3: by in_tab[slot][e] (32 g^e for the coset shift g of the column's slot: vdb_coeff_to_cosets_dev)
  uint32_t vslots;                // > 0: the launch's columns are virtual — column c transforms input column c / vslots for slot c % vslots
  const u256* in_tab[4];          // coset == 3: the slots' tables of input factors (n entries each)
  uint32_t coset_out;             // last pass: multiply output element e by zeta^-(e mod 3) (extended_to_coeff)
  uint32_t scale;                 // multiply output by n^{-1}
  uint32_t s0;                    // first butterfly stage to run (2 when the top three quarters of every row are zero padding)
  uint64_t in_len;                // elements >= in_len of the input column read as zero
  uint64_t in_stride, out_stride; // column strides (elements)
  uint32_t ren_mask;              // bit i: butterfly step i starts with a carry pass over its operands
  uint32_t ren_out;               // the write-out starts with a carry pass (the last step left limbs a product cannot take)
  uint32_t blk0;                  // the step at stage 2 takes the product-free path for its block 0
  const ColSrc* srcs;             // pass 0 only: columns still lying in a witness stream (null: read `in`)
  uint32_t n_blind;
  uint32_t n_cols;                // columns of this launch
  uint32_t col_group;             // 0: workgroups run column after column; g > 0: inside groups of g columns they run tile
                                  // after tile (the same tile of all g columns back to back), so a tile's 32-B inter-pass
                                  // twiddles and stage twiddles are L2 hits for every column after the first on an XCD
  u256 zeta1, zeta2, fin;         // 32*zeta, 32*zeta^2, 32/n — all mod r, Montgomery (zeta^-1 = zeta^2: the same two serve coset_out)
  u256 zeta0;                     // coset == 2: 32 s (and zeta1, zeta2 carry the factor s too)
  uint32_t ckp[9];                // 14 r as limbs that dominate any normalised operand (see l9_sub)
  // Shoup products for the stage twiddles (shoup_core29, field.hpp): per twiddle index e that is a multiple of 2^sh_res_log the
  // plain residue of omega_m^e and its quotient floor(omega_m^e 2^261 / r), 2 x 9 limbs; null: every product is a Montgomery product
  const uint32_t* sh_tab;
  uint32_t sh_res_log, sh_ns;     // resolution and number of entries (m / 2 >> sh_res_log)
  int32_t sh_max_s;               // radix-4 steps starting at stage s <= sh_max_s find both their stages' twiddles in the table
};

__device__ __forceinline__ uint32_t bitrev_s(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// limbs below 2^32, value below 2^259 -> canonical eight words without a multiplication and without a carry pass of its own:
// subtract q r with q = floor(top / (r_8 + 1)), top = l_8 + (l_7 >> 29) — the value's true top limb or one less (what limbs 0..6
// and the low 29 bits of l_7 carry into bit 232 is at most 1) — so q is never too large and at most 3 too small; the subtraction's
// 64-bit column accumulator propagates every carry on the way, and two conditional subtractions (2 r, then r) finish below 4 r
__device__ __forceinline__ u256 l9_canon_wide(L9 x) {
  constexpr uint32_t MU = 2840127191u;  // floor(2^53 / (0x30644e + 1))
  const uint32_t q = __umulhi((x.l[8] + (x.l[7] >> 29)) << 3, MU) >> 24;
  // x + q (2^261 - r) mod 2^261
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const uint32_t nk = (k == 0 ? 0x20000000u : 0x1fffffffu) - FrParams::P29[k];  // limbs of 2^261 - r
    acc += (uint64_t)x.l[k] + (uint64_t)q * nk;
    x.l[k] = (uint32_t)acc & 0x1fffffffu;
    acc >>= 29;
  }
  u256 v = l9_pack(x), t, p2, pp = mod_p<Fr>();
  u256_add(p2, pp, pp);
  uint32_t keep = u256_sub(t, v, p2);
#pragma unroll
  for (int i = 0; i < 8; i++) v.w[i] = keep ? v.w[i] : t.w[i];
  return lazy_canon<Fr>(v);
}
// products and the final conditional subtraction are over Fr in this file
__device__ __forceinline__ L9 l9_mul(const L9& a, const L9& W) { return l9_mul<Fr>(a, W); }
__device__ __forceinline__ u256 l9_canon(const L9& t) { return l9_canon<Fr>(t); }

struct L9Planes {
  uint4* a;      // limbs 0..3
  uint4* b;      // limbs 4..7
  uint32_t* c;   // limb 8
};
__device__ __forceinline__ L9 lds_get(const L9Planes& P, uint32_t idx) {
  uint4 a = P.a[idx], b = P.b[idx];
  L9 r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = P.c[idx];
  return r;
}
__device__ __forceinline__ void lds_put(const L9Planes& P, uint32_t idx, const L9& v) {
  P.a[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  P.b[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  P.c[idx] = v.l[8];
}

// One pass = a size-2^S DFT along one digit of the index for a tile of 2^S x G elements held in LDS as L9.
// Cooley-Tukey butterflies, natural order in, bit-reversed order out: (u, v) -> (u + w v, u - w v) with one twiddle
// per block, w = omega_m^(bitrev(block)).  Stage 0 has w = 1 everywhere (no product).  Values stay lazy: a product
// is normalised and below 1.2 r, sums and differences just add limbs; every third stage starts with a carry pass so
// multiplier inputs stay below 6 * 2^29 per limb and nothing reaches 2^32.  Values stay below 22 r over ten stages.
// `VS`: the first pass of vdb_coeff_to_cosets_dev (virtual columns, the slot's table of input factors) — an instantiation of its own,
// so that the step's transforms carry none of its branches
// `CS` > 0: the pass size is a compile-time constant, and with it the tile shape (G = 1024 / 2^CS), the first stage (`CS0`), the carry
// schedule (`CREN` = NttPass::ren_mask) and the Shoup table's resolution — the host launches such an instantiation for the sizes of
// the prover's transforms (256- and 512-point passes) when the pass's parameters are exactly these, the generic one (CS = 0)
// otherwise.  Every step then knows its stage: LDS offsets are immediates, the steps' variants that the schedule never takes are not
// instantiated, the loops over a thread's elements have constant trip counts.
template <uint32_t CS, uint32_t s, uint32_t step, class F>
__device__ __forceinline__ void ntt_static_steps(F&& f) {
  if constexpr (s < CS) {
    f(std::integral_constant<uint32_t, s>{}, std::integral_constant<uint32_t, step>{});
    ntt_static_steps<CS, (s + 1 < CS ? s + 2 : s + 1), step + 1>(f);
  }
}
constexpr uint32_t ntt_spec_sh_res_log(uint32_t cs) { return cs == 8 ? 0u : 2u; }   // what ntt_dev's LDS budget gives 256- / 512-point passes
template <bool LAST, bool VS = false, uint32_t CS = 0, uint32_t CS0 = 0, uint32_t CREN = 0>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(const u256* __restrict__ in, u256* __restrict__ out,
                                                         const u256* __restrict__ tw, const u256* __restrict__ tw_inter, NttPass p,
                                                         uint32_t tiles_per_col) {
  extern __shared__ uint4 smem[];
  constexpr bool SPEC = CS > 0;
  const uint32_t S = SPEC ? CS : p.S, P_logG = SPEC ? 10u - CS : p.logG, P_s0 = SPEC ? CS0 : p.s0, P_ren_mask = SPEC ? CREN : p.ren_mask;
  const uint32_t P_sh_res_log = SPEC ? ntt_spec_sh_res_log(CS) : p.sh_res_log, P_sh_ns = SPEC ? ((1u << (CS ? CS - 1 : 0)) >> ntt_spec_sh_res_log(CS)) : p.sh_ns;
  const int32_t P_sh_max_s = SPEC ? (int32_t)CS - 2 - (int32_t)ntt_spec_sh_res_log(CS) : p.sh_max_s;
  const bool P_shoup = SPEC ? true : p.sh_tab != nullptr, P_blk0 = SPEC ? true : p.blk0 != 0;
  const uint32_t m = 1u << S, G = 1u << P_logG, T = m * G;
  const uint32_t row = m + 1, NE = G * row, NW = m / 2 ? m / 2 : 1;
  L9Planes D, W;
  D.a = smem;
  D.b = D.a + NE;
  W.a = D.b + NE;
  W.b = W.a + NW;
  const uint32_t NS = P_shoup ? P_sh_ns : 0;
  L9Planes WS, WQ;
  WS.a = W.b + NW;
  WS.b = WS.a + NS;
  WQ.a = WS.b + NS;
  WQ.b = WQ.a + NS;
  D.c = reinterpret_cast<uint32_t*>(WQ.b + NS);
  W.c = D.c + NE;
  WS.c = W.c + NW;
  WQ.c = WS.c + NS;
  const uint32_t tid = threadIdx.x;
  uint32_t col, tile;
  if (p.col_group == 0) {
    col = blockIdx.x / tiles_per_col;
    tile = blockIdx.x % tiles_per_col;
  } else {
    const uint32_t per_group = p.col_group * tiles_per_col;
    const uint32_t grp = blockIdx.x / per_group, r = blockIdx.x % per_group;
    const uint32_t left = p.n_cols - grp * p.col_group, ncg = left < p.col_group ? left : p.col_group;
    col = grp * p.col_group + r % ncg;
    tile = r / ncg;
  }
  const u256* cin = in + (size_t)(VS ? col / p.vslots : col) * p.in_stride;
  const u256* ctab = nullptr;
  if constexpr (VS) {
    const uint32_t slot = col % p.vslots;
    ctab = slot == 0 ? p.in_tab[0] : (slot == 1 ? p.in_tab[1] : (slot == 2 ? p.in_tab[2] : p.in_tab[3]));
  }
  ColSrc csrc;
  const bool from_src = p.first && p.srcs != nullptr;
  if (from_src) csrc = p.srcs[col];
  u256* cout = out + (size_t)col * p.out_stride;

  // tile geometry
  uint64_t base;      // element index of (j=0, g=0)
  uint64_t jstride;   // index step along j
  uint64_t gstride;   // index step along g
  uint32_t i0 = 0, q0_base = 0, rdig = 0, rest = 1;
  if (!LAST) {
    uint32_t tiles_inner = (1u << p.log_inner) >> P_logG;
    uint32_t o = tile / tiles_inner;
    i0 = (tile % tiles_inner) << P_logG;
    base = ((uint64_t)o << (S + p.log_inner)) + i0;
    jstride = 1ull << p.log_inner;
    gstride = 1;
  } else {
    if (p.nprev == 0) {
      base = 0;
      jstride = 1;
      gstride = 0;
    } else {
      rest = 1u << (p.log_n - S - p.prevS[0]);
      q0_base = (tile / rest) << P_logG;
      rdig = tile % rest;
      base = ((uint64_t)q0_base * rest + rdig) << S;
      jstride = 1;
      gstride = (uint64_t)rest << S;
    }
  }

  // stage twiddles 32 * omega_m^e = tw[e * n/m], as limbs
  for (uint32_t e = tid; e < m / 2; e += NTT_THREADS) lds_put(W, e, l9_split(ld256(tw + ((size_t)e << (p.log_n - S)))));
  for (uint32_t e = tid; e < NS; e += NTT_THREADS) {
    L9 ws, wq;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      ws.l[k] = p.sh_tab[18 * e + k];
      wq.l[k] = p.sh_tab[18 * e + 9 + k];
    }
    lds_put(WS, e, ws);
    lds_put(WQ, e, wq);
  }
  // load tile.  With s0 == 2 only the first quarter of every row is data; the two skipped stages would just copy
  // it into the other three quarters.
  const uint32_t mload = m >> P_s0, Tload = mload * G;
  L9 Z0, Z1, Z2;
  if (!VS && p.coset) {
    Z1 = l9_split(p.zeta1);
    Z2 = l9_split(p.zeta2);
    if (p.coset == 2) Z0 = l9_split(p.zeta0);
  }
  for (uint32_t e = tid; e < Tload; e += NTT_THREADS) {
    uint32_t j, g;
    if (!LAST) {
      g = e & (G - 1);
      j = e >> P_logG;
    } else {
      j = e & (mload - 1);
      g = e / mload;
    }
    uint64_t idx = base + j * jstride + g * gstride;
    L9 v;
    if (!p.first || idx < p.in_len) {
      v = l9_split(from_src ? colsrc_fetch(csrc, idx, 1ull << p.log_n, p.n_blind) : ld256(cin + idx));
      if constexpr (VS) {
        v = l9_mul(v, l9_split(ld256(ctab + idx)));
      } else if (p.coset) {
        uint32_t r3 = (uint32_t)(idx % 3);
        if (r3 == 1) v = l9_mul(v, Z1);
        else if (r3 == 2) v = l9_mul(v, Z2);
        else if (p.coset == 2) v = l9_mul(v, Z0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) v.l[k] = 0;
    }
    for (uint32_t rep = 0; rep < (1u << P_s0); rep++) lds_put(D, g * row + j + rep * mload, v);
  }
  __syncthreads();
  // butterfly steps: two stages at a time on four elements held in registers (one LDS round trip and one barrier
  // per two stages), a single radix-2 stage at the end when the number of stages is odd
  auto do_step = [&](auto s_c, auto step_c) {
    const uint32_t s = s_c, step = step_c;
    const uint32_t logh = S - 1 - s, h = 1u << logh;
    const bool ren = (P_ren_mask >> step) & 1;
    if (s + 1 < S) {
      const uint32_t h2 = h >> 1, lh1 = logh ? logh - 1 : 0;   // (logh >= 1 here; the guard keeps the dead instantiations of a compile-time schedule well formed)
      // (the body is instantiated per (first step, carry pass) combination: a run-time `if (ren)` inside one body makes the
      // compiler merge the two register sets with ~30 moves per step on the path that does not renormalise)
      // `s2_c`: the step that starts at stage 2.  Its block 0 — a quarter of its butterflies — has the twiddles of the very first step
      // (1, 1 and omega_4): three of its four products are products by one.  The threads are numbered block-major there (block,
      // row, position), so that with 1024-element tiles a wavefront lies inside one block and the branch costs nothing.
      auto radix4 = [&](auto first_c, auto ren_c, auto shoup_c, auto s2_c) {
        constexpr bool FIRST = decltype(first_c)::value, REN = decltype(ren_c)::value, SHOUP = decltype(shoup_c)::value, S2 = decltype(s2_c)::value;
        for (uint32_t t = tid; t < T / 4; t += NTT_THREADS) {
          uint32_t g, r, blk;
          if (S2) {
            const uint32_t lb = P_logG + lh1;   // log2(threads per block) = log2(G h2)
            // (rotated by the workgroup's number: a CU's resident workgroups then put their light wavefront on different SIMDs)
            blk = ((t >> lb) + blockIdx.x + (blockIdx.x >> 8)) & 3u;
            g = (t >> lh1) & (G - 1);
            r = t & (h2 - 1);
          } else {
            const uint32_t gi = t & ((m >> 2) - 1);
            g = t >> (S - 2);
            r = gi & (h2 - 1);
            blk = gi >> lh1;
          }
          const uint32_t j0 = g * row + (blk << (logh + 1)) + r;
          L9 x0 = lds_get(D, j0), x1 = lds_get(D, j0 + h2), x2 = lds_get(D, j0 + h), x3 = lds_get(D, j0 + h + h2);
          if (REN) {
            l9_renorm(x0);
            l9_renorm(x1);
            l9_renorm(x2);
            l9_renorm(x3);
          }
          if (FIRST) {
            // stage 0: every twiddle is 1.  stage 1: block 0 has twiddle 1, block 1 has omega_4.
            L9 a0 = l9_add(x0, x2), a2 = l9_sub(x0, x2, p.ckp);
            L9 a1 = l9_add(x1, x3), a3 = l9_sub(x1, x3, p.ckp);
            l9_carry(a1);  // a1 is subtracted below: its limbs must be below 2^29 again
            const L9 t3 = l9_mul(a3, lds_get(W, m >> 2));
            x0 = l9_add(a0, a1);
            x1 = l9_sub(a0, a1, p.ckp);
            x2 = l9_add(a2, t3);
            x3 = l9_sub(a2, t3, p.ckp);
          } else if (S2 && blk == 0) {
            // twiddles 1, 1, omega_4.  x2, x3 and a1 are subtracted as they are: their limbs are brought below 2^29 first (their values,
            // at most 4.8 r and 9.6 r — block 0 holds the sums of sums of the first step, or freshly loaded values — stay below the
            // 13 r the offset covers); every limb bound stays below the general path's
            l9_carry(x2);
            l9_carry(x3);
            L9 a0 = l9_add(x0, x2), a2 = l9_sub(x0, x2, p.ckp);
            L9 a1 = l9_add(x1, x3), a3 = l9_sub(x1, x3, p.ckp);
            l9_carry(a1);
            L9 t3;
            if (SHOUP) {
              const uint32_t iq = (m >> 2) >> P_sh_res_log;
              t3 = l9_mul_shoup<Fr>(a3, lds_get(WS, iq), lds_get(WQ, iq));
            } else {
              t3 = l9_mul(a3, lds_get(W, m >> 2));
            }
            x0 = l9_add(a0, a1);
            x1 = l9_sub(a0, a1, p.ckp);
            x2 = l9_add(a2, t3);
            x3 = l9_sub(a2, t3, p.ckp);
          } else if (SHOUP) {
            // both stages' twiddles lie in the quarter- (or full-) resolution table of plain residues and quotients: the products
            // come out below 3 r with normalised limbs, which is all the sums and differences below ask of them
            const uint32_t e = bitrev_s(blk, s) << logh, rl = P_sh_res_log;
            const uint32_t i0 = e >> rl, i1 = (e >> 1) >> rl, i2 = ((e >> 1) + (m >> 2)) >> rl;
            const L9 ws = lds_get(WS, i0), wq = lds_get(WQ, i0);
            const L9 t2 = l9_mul_shoup<Fr>(x2, ws, wq), t3 = l9_mul_shoup<Fr>(x3, ws, wq);
            const L9 a0 = l9_add(x0, t2), a2 = l9_sub(x0, t2, p.ckp);
            const L9 a1 = l9_add(x1, t3), a3 = l9_sub(x1, t3, p.ckp);
            const L9 u1 = l9_mul_shoup<Fr>(a1, lds_get(WS, i1), lds_get(WQ, i1));
            const L9 u3 = l9_mul_shoup<Fr>(a3, lds_get(WS, i2), lds_get(WQ, i2));
            x0 = l9_add(a0, u1);
            x1 = l9_sub(a0, u1, p.ckp);
            x2 = l9_add(a2, u3);
            x3 = l9_sub(a2, u3, p.ckp);
          } else {
            const uint32_t e = bitrev_s(blk, s) << logh;
            const L9 w = lds_get(W, e);
            const L9 t2 = l9_mul(x2, w), t3 = l9_mul(x3, w);
            const L9 a0 = l9_add(x0, t2), a2 = l9_sub(x0, t2, p.ckp);
            const L9 a1 = l9_add(x1, t3), a3 = l9_sub(x1, t3, p.ckp);
            const L9 u1 = l9_mul(a1, lds_get(W, e >> 1));
            const L9 u3 = l9_mul(a3, lds_get(W, (e >> 1) + (m >> 2)));
            x0 = l9_add(a0, u1);
            x1 = l9_sub(a0, u1, p.ckp);
            x2 = l9_add(a2, u3);
            x3 = l9_sub(a2, u3, p.ckp);
          }
          lds_put(D, j0, x0);
          lds_put(D, j0 + h2, x1);
          lds_put(D, j0 + h, x2);
          lds_put(D, j0 + h + h2, x3);
        }
      };
      using T_ = std::true_type;
      using F_ = std::false_type;
      const bool s2 = s == 2 && !ren && P_blk0;   // (the schedule never carries at the start of this step; if it ever did: the general body)
      if (s == 0) {
        if (ren) radix4(T_{}, T_{}, F_{}, F_{});
        else radix4(T_{}, F_{}, F_{}, F_{});
      } else if (P_shoup && (int32_t)s <= P_sh_max_s) {
        if (ren) radix4(F_{}, T_{}, T_{}, F_{});
        else if (s2) radix4(F_{}, F_{}, T_{}, T_{});
        else radix4(F_{}, F_{}, T_{}, F_{});
      } else {
        if (ren) radix4(F_{}, T_{}, F_{}, F_{});
        else if (s2) radix4(F_{}, F_{}, F_{}, T_{});
        else radix4(F_{}, F_{}, F_{}, F_{});
      }
    } else {
      for (uint32_t b = tid; b < T / 2; b += NTT_THREADS) {
        const uint32_t g = b >> (S - 1), pj = b & ((m >> 1) - 1);
        const uint32_t r = pj & (h - 1), blk = pj >> logh, j0 = (blk << (logh + 1)) + r;
        const uint32_t a0 = g * row + j0, a1 = a0 + h;
        L9 u = lds_get(D, a0), v = lds_get(D, a1);
        if (ren) {
          l9_renorm(u);
          l9_renorm(v);
        }
        if (s) v = l9_mul(v, lds_get(W, bitrev_s(blk, s) << logh));
        lds_put(D, a0, l9_add(u, v));
        lds_put(D, a1, l9_sub(u, v, p.ckp));
      }
    }
    __syncthreads();
  };
  if constexpr (SPEC) {
    ntt_static_steps<CS, CS0, 0>(do_step);
  } else {
    uint32_t step = 0;
    for (uint32_t s = P_s0; s < S; step++) {
      do_step(s, step);
      s += (s + 1 < S) ? 2 : 1;
    }
  }
  // write out: the inter-pass twiddle product (or the 1/n product of an inverse transform) brings the value below 2r
  // on its own; a forward transform's last pass reduces without a product (l9_canon_wide)
  L9 FIN, ZO1, ZO2;
  if (LAST && p.scale) FIN = l9_split(p.fin);
  if (LAST && p.coset_out) {
    ZO1 = l9_split(p.zeta2);  // zeta^-1
    ZO2 = l9_split(p.zeta1);  // zeta^-2
  }
  // inter-pass twiddles are fetched for all of the thread's elements up front: on gfx9 a load that follows stores
  // waits for their acknowledgement too (one in-order vmcnt), which would put a store round trip between elements
  constexpr uint32_t EPT = NTT_TILE / NTT_THREADS;
  u256 twv[EPT];
  if (!LAST) {
#pragma unroll
    for (uint32_t it = 0; it < EPT; it++) {
      const uint32_t e = tid + it * NTT_THREADS;
      const uint32_t g = e & (G - 1), q = e >> P_logG;
      const uint64_t ex = ((uint64_t)q * ((uint64_t)i0 + g)) << (p.log_n - S - p.log_inner);
      if (e < T) twv[it] = ld256(tw_inter + ex);
    }
  }
#pragma unroll
  for (uint32_t it = 0; it < EPT; it++) {
    const uint32_t e = tid + it * NTT_THREADS;
    if (e >= T) break;
    uint32_t g = e & (G - 1), q = e >> P_logG;
    L9 v = lds_get(D, g * row + bitrev_s(q, S));
    if (p.ren_out) l9_renorm(v);
    if (!LAST) {
      v = l9_mul(v, l9_split(twv[it]));
      // below 1.2 r and exactly normalised: stored without the final conditional subtraction (the next pass only needs
      // normalised limbs below 1.9 r)
      st256(cout + base + (uint64_t)q * jstride + g, l9_pack(v));
    } else {
      uint64_t pos;
      if (p.nprev == 0) {
        pos = q;
      } else {
        pos = (uint64_t)q0_base + g;
        uint32_t r = rdig;
        uint32_t shift = 0;
        for (uint32_t l = 0; l < p.nprev; l++) shift += p.prevS[l];
        // peel digits q_{L-2} .. q_1 (least significant first)
        for (int l = (int)p.nprev - 1; l >= 1; l--) {
          shift -= p.prevS[l];
          uint32_t ql = r & ((1u << p.prevS[l]) - 1);
          r >>= p.prevS[l];
          pos += (uint64_t)ql << shift;
        }
        pos += (uint64_t)q << (p.log_n - S);
      }
      if (p.scale) {
        v = l9_mul(v, FIN);  // single-pass inverse transform: 1/n here
        const uint32_t r3 = p.coset_out ? (uint32_t)(pos % 3) : 0u;
        if (r3) v = l9_mul(v, r3 == 1 ? ZO1 : ZO2);
        st256(cout + pos, l9_canon(v));
      } else {
        const uint32_t r3 = p.coset_out ? (uint32_t)(pos % 3) : 0u;
        st256(cout + pos, r3 ? l9_canon(l9_mul(v, r3 == 1 ? ZO1 : ZO2)) : l9_canon_wide(v));
      }
    }
  }
}

// tw[e] = 32 * omega^e for e < n (the factor 32 = 2^261 / 2^256 makes a 9 x 29-bit product with the table entry an
// ordinary Montgomery product, see mont_core29); `m32` is 32 in Montgomery form
__global__ __launch_bounds__(256) void k_twiddles(u256* __restrict__ tw, u256 omega, u256 m32, uint64_t n, uint32_t chunk) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t lo = t * chunk;
  if (lo >= n) return;
  u256 w = mont_pow<Fr>(omega, u256_from_u64(lo));
  for (uint32_t i = 0; i < chunk && lo + i < n; i++) {
    st256(tw + lo + i, fr_mul(w, m32));
    w = fr_mul(w, omega);
  }
}

// `factor`: what every entry is multiplied by — 32 (plain table) or 32 / n (`scaled`: the inter-pass table of an inverse
// transform, which folds the final 1/n into a product that is made anyway)
static const u256* get_twiddles(uint32_t log_n, const u256& omega, const u256& factor, bool scaled, int* err) {
  Context& c = ctx();
  Context::TwKey key{log_n | (scaled ? 0x80000000u : 0u), omega};
  auto it = c.twiddles.find(key);
  if (it != c.twiddles.end()) return it->second;
  uint64_t n = 1ull << log_n;
  u256* tw = nullptr;
  hipError_t e = hipMalloc(&tw, n * sizeof(u256));
  if (e != hipSuccess) {
    *err = hip_fail(e, "hipMalloc(twiddles)", __FILE__, __LINE__);
    return nullptr;
  }
  uint32_t chunk = 16;
  uint64_t threads = (n + chunk - 1) / chunk;
  {
    VDB_PROF("k_twiddles");
    hipLaunchKernelGGL(k_twiddles, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, tw, omega, factor, n, chunk);
  }
  e = hipGetLastError();
  if (e != hipSuccess) {
    *err = hip_fail(e, "k_twiddles", __FILE__, __LINE__);
    (void)hipFree(tw);
    return nullptr;
  }
  c.twiddles[key] = tw;
  return tw;
}

// Shoup table of one pass size (see NttPass::sh_tab): entry j holds, for e = j << res_log, the plain residue of
// omega_m^e = omega^(e n / m) and floor(that * 2^261 / r), nine 29-bit limbs each.  Built on the host (a few hundred entries, a
// 261-step long division each), cached per device beside the twiddle tables.
static void limbs29(const u256& a, uint32_t extra_top /* bits 256.. */, uint32_t out[9]) {
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, wd = bit >> 5, sh = bit & 31;
    uint64_t v = wd < 8 ? (uint64_t)a.w[wd] >> sh : 0;
    if (sh && wd + 1 < 8) v |= (uint64_t)a.w[wd + 1] << (32 - sh);
    if (wd + 1 == 8 && sh) v |= (uint64_t)extra_top << (32 - sh);
    if (wd == 8) v = extra_top >> sh;
    out[k] = (uint32_t)v & 0x1fffffffu;
  }
}
static const uint32_t* get_shoup_table(uint32_t log_n, const u256& omega, uint32_t S, uint32_t res_log, int* err) {
  Context& c = ctx();
  Context::TwKey key{log_n | (S << 8) | (res_log << 16) | 0x40000000u, omega};
  auto it = c.twiddles.find(key);
  if (it != c.twiddles.end()) return reinterpret_cast<const uint32_t*>(it->second);
  const uint32_t m = 1u << S, ns = (m / 2) >> res_log;
  std::vector<uint32_t> tab((size_t)ns * 18);
  const u256 p = mod_p<Fr>();
  const u256 step = mont_pow<Fr>(omega, u256_from_u64(((uint64_t)1 << (log_n - S)) << res_log));  // omega_m^(2^res_log)
  u256 cur = mont_one<Fr>();
  for (uint32_t j = 0; j < ns; j++) {
    const u256 w = from_mont<Fr>(cur);
    limbs29(w, 0, &tab[18 * (size_t)j]);
    // q = floor(w 2^261 / p): r starts at w (< p) and takes 261 doubling steps; r stays below 2 p < 2^255 before each reduction
    u256 r = w;
    uint32_t q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 260; b >= 0; b--) {
      u256 d;
      u256_add(d, r, r);
      u256 t;
      const uint32_t borrow = u256_sub(t, d, p);
      r = borrow ? d : t;
      if (!borrow) q[b / 29] |= 1u << (b % 29);
    }
    for (int k = 0; k < 9; k++) tab[18 * (size_t)j + 9 + k] = q[k];
    cur = fr_mul(cur, step);
  }
  uint32_t* d = nullptr;
  hipError_t e = hipMalloc(&d, tab.size() * sizeof(uint32_t));
  if (e != hipSuccess) {
    *err = hip_fail(e, "hipMalloc(shoup table)", __FILE__, __LINE__);
    return nullptr;
  }
  e = hipMemcpy(d, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    *err = hip_fail(e, "hipMemcpy(shoup table)", __FILE__, __LINE__);
    (void)hipFree(d);
    return nullptr;
  }
  c.twiddles[key] = reinterpret_cast<u256*>(d);
  return d;
}

// Plans and runs the passes.  data: n_cols columns (stride in_len when in_len != 0, else n);
// result goes to out (stride n) or back into data when out == nullptr.
int ntt_dev(u256* data, u256* out_or_null, size_t n_cols, uint32_t log_n, const u256& omega, bool scale_ninv,
            bool coset_in, size_t in_len, const ColSrc* srcs, uint32_t n_blind, bool coset_out, const u256* in_scale, const u256* const* in_tabs,
            uint32_t vslots) {
  // `in_tabs` / `vslots`: n_cols counts virtual columns — column c is input column c / vslots multiplied element by element by
  // in_tabs[c % vslots] (the cosets of one polynomial, one after the other in the output)
  Context& c = ctx();
  if (n_cols == 0) return VDB_OK;
  if (log_n > 26) {
    set_error("ntt: log_n > 26 unsupported");
    return VDB_ERR_ARG;
  }
  const uint64_t n = 1ull << log_n;
  const uint64_t in_stride = in_len ? in_len : n;
  if (!in_len) in_len = n;
  u256* dst = out_or_null ? out_or_null : data;
  if (!out_or_null && in_stride != n) {
    set_error("ntt: in-place transform needs in_len == n");
    return VDB_ERR_ARG;
  }
  if (vslots && (!out_or_null || !in_tabs || vslots > 4 || srcs || coset_in || n_cols % vslots)) {
    set_error("ntt: virtual columns need an output buffer and their input tables");
    return VDB_ERR_ARG;
  }
  if (srcs && (!out_or_null || in_len != n || log_n <= 10)) {
    set_error("ntt: column sources need an output buffer and a multi-pass size");
    return VDB_ERR_ARG;
  }
  int err = VDB_OK;
  const u256 m32 = host_fr_from_u64(32);
  const u256* tw = get_twiddles(log_n, omega, m32, false, &err);
  if (!tw) return err;

  // pass sizes
  uint32_t L, S[NTT_MAX_PASSES];
  if (log_n <= 10) {
    L = 1;
    S[0] = log_n;
  } else {
    static const uint32_t max_s = getenv("VDB_NTT_MAX_S") ? (uint32_t)atoi(getenv("VDB_NTT_MAX_S")) : 9u;
    uint32_t ms = max_s >= 4 && max_s <= 9 ? max_s : 9u;
    while ((log_n + ms - 1) / ms > NTT_MAX_PASSES) ms++;  // at most NTT_MAX_PASSES digits
    L = (log_n + ms - 1) / ms;  // passes of up to 512 points: 2^16 = 256 x 256, 2^18 = 512 x 512
    for (uint32_t l = 0; l < L; l++) S[l] = log_n / L + (l < log_n % L ? 1 : 0);
  }
  const u256 fin = scale_ninv ? fr_mul(mont_inv<Fr>(host_fr_from_u64(n)), m32) : m32;
  // multi-pass inverse transforms take 1/n with the inter-pass twiddles of pass 0; their last pass then ends like a
  // forward transform's, without a product
  const u256* tw_scaled = tw;
  if (scale_ninv && L > 1) {
    tw_scaled = get_twiddles(log_n, omega, fin, true, &err);
    if (!tw_scaled) return err;
  }
  u256 z1 = host_zeta(), z2 = fr_mul(z1, z1);
  z1 = fr_mul(z1, m32);
  z2 = fr_mul(z2, m32);
  u256 z0 = m32;
  if (in_scale) {  // the polynomial s p(X) on the coset: the scalar rides on the coset factors (one more product on a third of the inputs)
    z0 = fr_mul(z0, *in_scale);
    z1 = fr_mul(z1, *in_scale);
    z2 = fr_mul(z2, *in_scale);
  }
  static const bool shoup_on = !(getenv("VDB_NTT_SHOUP") && getenv("VDB_NTT_SHOUP")[0] == '0');
  // 14 r with limbs that dominate a normalised subtrahend (l9_sub)
  uint32_t ckp[9];
  const double cmax = l9_offset_limbs<FrParams>(14, ckp);

  // column chunking bounds the scratch buffer (<= ~2 GiB)
  size_t chunk_cols = n_cols;
  u256* scratch = nullptr;
  if (L > 1) {
    size_t max_cols = ((size_t)2 << 30) / (n * sizeof(u256));
    if (max_cols < 1) max_cols = 1;
    if (chunk_cols > max_cols) chunk_cols = max_cols;
    if (vslots) chunk_cols = chunk_cols / vslots * vslots ? chunk_cols / vslots * vslots : vslots;
    scratch = (u256*)scratch_get(3, chunk_cols * n * sizeof(u256));
    if (!scratch) return VDB_ERR_OOM;
  }
  // launch order of the non-last passes (VDB_NTT_COLGROUP: 0 = column after column)
  static const uint32_t col_group = getenv("VDB_NTT_COLGROUP") ? (uint32_t)strtoul(getenv("VDB_NTT_COLGROUP"), nullptr, 10) : NTT_DEFAULT_COLGROUP;
  for (size_t c0 = 0; c0 < n_cols; c0 += chunk_cols) {
    size_t nc = n_cols - c0 < chunk_cols ? n_cols - c0 : chunk_cols;
    uint32_t done_bits = 0;
    for (uint32_t l = 0; l < L; l++) {
      NttPass p;
      memset(&p, 0, sizeof(p));
      p.log_n = log_n;
      p.S = S[l];
      p.log_inner = log_n - done_bits - S[l];
      p.first = (l == 0);
      const bool last = (l == L - 1);
      p.coset = (l == 0 && coset_in) ? (in_scale ? 2u : 1u) : ((l == 0 && vslots) ? 3u : 0u);
      p.vslots = vslots;
      for (uint32_t t = 0; t < vslots; t++) p.in_tab[t] = in_tabs[t];
      p.zeta0 = z0;
      p.in_len = in_len;
      p.srcs = (l == 0 && srcs) ? srcs + c0 : nullptr;
      p.n_blind = n_blind;
      p.n_cols = (uint32_t)nc;
      p.col_group = last ? 0u : (col_group < nc ? col_group : (uint32_t)nc);  // the last pass reads no inter-pass twiddles
      p.zeta1 = z1;
      p.zeta2 = z2;
      p.fin = fin;
      memcpy(p.ckp, ckp, sizeof(ckp));
      p.scale = last && scale_ninv && L == 1;
      p.coset_out = last && coset_out;
      // zero-padded input (coeff_to_extended): when rows of pass 0 run along the top digit and only their first
      // quarter is data, stages 0 and 1 are pure replication
      p.s0 = (l == 0 && !last && S[0] >= 3 && in_len * 4 <= n) ? 2 : 0;
      // carry-pass schedule: limb bounds in units of 2^29 (a sum adds the operands' bounds, a difference adds cmax);
      // products need operands below 6.1, nothing may reach 8
      {
        double b = 1.0001;
        uint32_t step = 0;
        for (uint32_t st = p.s0; st < S[l]; step++) {
          if (st + 1 < S[l]) {
            if (st == 0) {
              b = 1.0 + 2.0 * cmax;
            } else {
              if (b + cmax >= 6.1 || b + 2.0 * cmax >= 7.95) {
                p.ren_mask |= 1u << step;
                b = 1.0001;
              }
              b += 2.0 * cmax;
            }
            st += 2;
          } else {
            if (st == 0) {
              b = 1.0 + cmax;
            } else {
              if (b >= 6.1 || b + cmax >= 7.95) {
                p.ren_mask |= 1u << step;
                b = 1.0001;
              }
              b += cmax;
            }
            st += 1;
          }
        }
        // the write-out multiplies (inter-pass twiddle, 1/n, zeta) unless it is a forward transform's last pass, whose reduction
        // carries on its own (l9_canon_wide): a carry pass only when the last step left limbs above what a product takes
        const bool out_mul = !last || p.scale || p.coset_out;
        static const bool blk0_on = !(getenv("VDB_NTT_BLK0") && getenv("VDB_NTT_BLK0")[0] == '0');
        p.blk0 = blk0_on ? 1u : 0u;
        static const bool always = getenv("VDB_NTT_REN_OUT") && getenv("VDB_NTT_REN_OUT")[0] == '1';   // A/B: the unconditional pass of rounds 1-2
        p.ren_out = (always || (out_mul && b >= 6.1)) ? 1u : 0u;
      }
      const u256* src;
      u256* out;
      if (l == 0) {
        src = srcs ? scratch : data + (vslots ? c0 / vslots : c0) * in_stride;  // with column sources `in` is not read
        p.in_stride = in_stride;
      } else {
        src = scratch;
        p.in_stride = n;
      }
      if (last) {
        out = dst + c0 * n;
        p.out_stride = n;
      } else {
        out = scratch;
        p.out_stride = n;
      }
      uint32_t m = 1u << S[l];
      if (L == 1) {
        p.logG = 0;
        p.nprev = 0;
      } else {
        p.logG = 10 - S[l];  // 1024-element tiles: G = 4 for 256-point, G = 2 for 512-point DFTs
        // a tile cannot be wider than the digit it is cut from: the inner index of a non-last pass, the first digit of the last
        const uint32_t room = last ? S[0] : p.log_inner;
        if (p.logG > room) p.logG = room;
        p.nprev = l;  // only read when last
        for (uint32_t q = 0; q < l && q < NTT_MAX_PASSES; q++) p.prevS[q] = S[q];
      }
      uint32_t G = 1u << p.logG;
      uint32_t tiles = (uint32_t)(n / ((uint64_t)m * G));
      size_t lds = (size_t)(G * (m + 1) + (m / 2 ? m / 2 : 1)) * (2 * sizeof(uint4) + sizeof(uint32_t));
      // Shoup table in what is left of a third of the CU's LDS (three workgroups per CU stay resident): full resolution for the
      // 256-point passes, a quarter for the 512-point ones; passes too small to have a general radix-4 step do without
      p.sh_tab = nullptr;
      p.sh_res_log = p.sh_ns = 0;
      p.sh_max_s = -1;
      if (shoup_on && S[l] >= 4) {
        const size_t room = (160 * 1024) / 3;
        uint32_t rl = 0;
        while (rl + 2 < S[l] && lds + (size_t)((m / 2) >> rl) * 72 > room) rl++;
        const int32_t max_s = (int32_t)S[l] - 2 - (int32_t)rl;
        if (lds + (size_t)((m / 2) >> rl) * 72 <= room && max_s >= 1) {
          p.sh_tab = get_shoup_table(log_n, omega, S[l], rl, &err);
          if (!p.sh_tab) return err;
          p.sh_res_log = rl;
          p.sh_ns = (m / 2) >> rl;
          p.sh_max_s = max_s;
          lds += (size_t)p.sh_ns * 72;
        }
      }
      dim3 grid((unsigned)(nc * tiles));
      if (lds > 64 * 1024) {
        bool& raised = c.ntt_lds_raised;  // per device: the attribute belongs to the device's code object
        if (!raised) {
          VDB_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          VDB_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          VDB_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          VDB_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          raised = true;
        }
      }
      // the instantiation with this pass's size, first stage, carry schedule and Shoup resolution as compile-time constants when there is
      // one (the 256- and 512-point passes of the prover's transforms), the generic kernel otherwise (VDB_NTT_SPEC=0: always)
      static const bool spec_on = !(getenv("VDB_NTT_SPEC") && getenv("VDB_NTT_SPEC")[0] == '0');
      const u256* twi = last ? tw : (l == 0 ? tw_scaled : tw);
      const bool vs = (l == 0 && vslots);
      auto spec = [&](uint32_t cs, uint32_t cs0, uint32_t cren) {
        const uint32_t rl = ntt_spec_sh_res_log(cs);
        return spec_on && L > 1 && p.S == cs && p.s0 == cs0 && p.ren_mask == cren && p.logG == 10 - cs && p.sh_tab && p.blk0 && p.sh_res_log == rl &&
               p.sh_ns == ((1u << (cs - 1)) >> rl) && p.sh_max_s == (int32_t)cs - 2 - (int32_t)rl;
      };
#define NTT_LAUNCH(...) hipLaunchKernelGGL((k_ntt_pass<__VA_ARGS__>), grid, dim3(NTT_THREADS), lds, c.stream, src, out, tw, twi, p, tiles)
      {
        VDB_PROF("k_ntt_pass");
        if (last) {
          if (vs) NTT_LAUNCH(true, true);
          else if (spec(8, 0, 4)) NTT_LAUNCH(true, false, 8, 0, 4);
          else if (spec(9, 0, 20)) NTT_LAUNCH(true, false, 9, 0, 20);
          else NTT_LAUNCH(true, false);
        } else {
          if (vs && spec(8, 0, 4)) NTT_LAUNCH(false, true, 8, 0, 4);
          else if (vs) NTT_LAUNCH(false, true);
          else if (spec(8, 0, 4)) NTT_LAUNCH(false, false, 8, 0, 4);
          else if (spec(9, 2, 4)) NTT_LAUNCH(false, false, 9, 2, 4);
          else if (spec(9, 0, 20)) NTT_LAUNCH(false, false, 9, 0, 20);
          else NTT_LAUNCH(false, false);
        }
      }
#undef NTT_LAUNCH
      VDB_LAUNCH_CHECK();
      done_bits += S[l];
    }
  }
  return VDB_OK;
}

// out[j][i] = sum_t C[j][t] * tab_t[i] * P[t][i]: the pieces of a polynomial of degree below n_slots n from its residues modulo
// X^n - u_t (what an inverse transform of its values on the coset g_t H gives, once coefficient i is divided by g_t^i)
struct CosetMix {
  const u256* tab[4];  // g_t^-i, Montgomery
  u256 c[4][4];
};
__global__ __launch_bounds__(256) void k_cosets_combine(const u256* __restrict__ P, uint64_t n, uint32_t n_slots, CosetMix mx, u256* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 v[4];
#pragma unroll
  for (uint32_t t = 0; t < 4; t++)
    if (t < n_slots) v[t] = fr_mul(ld256(P + t * n + i), ld256(mx.tab[t] + i));
#pragma unroll
  for (uint32_t j = 0; j < 4; j++) {
    if (j >= n_slots) break;
    u256 acc = fr_mul(v[0], mx.c[j][0]);
#pragma unroll
    for (uint32_t t = 1; t < 4; t++)
      if (t < n_slots) acc = fr_add(acc, fr_mul(v[t], mx.c[j][t]));
    st256(out + j * n + i, acc);
  }
}

u256 host_coset_shift(uint32_t k, uint32_t t) {
  static const uint32_t rev[4] = {0, 2, 1, 3};
  return fr_mul(host_zeta(), mont_pow<Fr>(host_root_of_unity(k + 2), u256_from_u64(rev[t & 3])));
}

static int host_cols_roundtrip(vdb_fr* const* cols, size_t n_cols, size_t n, u256** dbuf, bool upload) {
  Context& c = ctx();
  size_t bytes = n * sizeof(u256);
  if (upload) {
    *dbuf = (u256*)scratch_get(0, n_cols * bytes);
    if (!*dbuf) return VDB_ERR_OOM;
    for (size_t i = 0; i < n_cols; i++) VDB_HIP(hipMemcpyAsync(*dbuf + i * n, cols[i], bytes, hipMemcpyHostToDevice, c.stream));
  } else {
    for (size_t i = 0; i < n_cols; i++) VDB_HIP(hipMemcpyAsync(cols[i], *dbuf + i * n, bytes, hipMemcpyDeviceToHost, c.stream));
    VDB_HIP(hipStreamSynchronize(c.stream));
  }
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_ntt_batch_dev(vdb_fr* cols_dev, size_t n_cols, uint32_t log_n, const vdb_fr* omega, int flags) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols_dev && omega, "null pointer");
  u256 w;
  memcpy(&w, omega, 32);
  return ntt_dev(as_u256(cols_dev), nullptr, n_cols, log_n, w, (flags & VDB_NTT_INVERSE_SCALE) != 0, false, 0, nullptr, 0, false, nullptr);
}
int vdb_ntt_batch(vdb_fr* const* cols, size_t n_cols, uint32_t log_n, const vdb_fr* omega, int flags) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols && omega, "null pointer");
  if (n_cols == 0) return VDB_OK;
  u256* d = nullptr;
  int rc = host_cols_roundtrip(cols, n_cols, (size_t)1 << log_n, &d, true);
  if (rc) return rc;
  rc = vdb_ntt_batch_dev(reinterpret_cast<vdb_fr*>(d), n_cols, log_n, omega, flags);
  if (rc) return rc;
  return host_cols_roundtrip(cols, n_cols, (size_t)1 << log_n, &d, false);
}
int vdb_lagrange_to_coeff_dev(vdb_fr* cols_dev, size_t n_cols, uint32_t k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols_dev && k <= 26, "bad argument");
  u256 w = mont_inv<Fr>(host_root_of_unity(k));
  return ntt_dev(as_u256(cols_dev), nullptr, n_cols, k, w, true, false, 0, nullptr, 0, false, nullptr);
}
int vdb_lagrange_to_coeff_src_dev(const vdb_colsrc* src_dev, vdb_fr* coeff_dev, size_t n_cols, uint32_t k, uint32_t n_blind) {
  VDB_REQUIRE_INIT();
  VDB_ARG(src_dev && coeff_dev && k <= 26 && k > 10, "bad argument (column sources are supported for k > 10)");
  u256 w = mont_inv<Fr>(host_root_of_unity(k));
  return ntt_dev(nullptr, as_u256(coeff_dev), n_cols, k, w, true, false, 0, reinterpret_cast<const ColSrc*>(src_dev), n_blind, false, nullptr);
}
int vdb_lagrange_to_coeff(vdb_fr* const* cols, size_t n_cols, uint32_t k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols, "null pointer");
  if (n_cols == 0) return VDB_OK;
  u256* d = nullptr;
  int rc = host_cols_roundtrip(cols, n_cols, (size_t)1 << k, &d, true);
  if (rc) return rc;
  rc = vdb_lagrange_to_coeff_dev(reinterpret_cast<vdb_fr*>(d), n_cols, k);
  if (rc) return rc;
  return host_cols_roundtrip(cols, n_cols, (size_t)1 << k, &d, false);
}
int vdb_coeff_to_extended_dev(const vdb_fr* coeff_dev, vdb_fr* ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && ext_dev && k + ext_k <= 26, "bad argument");
  u256 w = host_root_of_unity(k + ext_k);
  return ntt_dev(const_cast<u256*>(as_u256(coeff_dev)), as_u256(ext_dev), n_cols, k + ext_k, w, false, true, (size_t)1 << k, nullptr, 0, false, nullptr);
}
int vdb_coeff_to_extended_scaled_dev(const vdb_fr* coeff_dev, vdb_fr* ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* scale) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && ext_dev && scale && k + ext_k <= 26, "bad argument");
  u256 w = host_root_of_unity(k + ext_k), sv;
  memcpy(&sv, scale, 32);
  return ntt_dev(const_cast<u256*>(as_u256(coeff_dev)), as_u256(ext_dev), n_cols, k + ext_k, w, false, true, (size_t)1 << k, nullptr, 0, false, &sv);
}
int vdb_extended_to_coeff_dev(vdb_fr* ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(ext_dev && k + ext_k <= 26, "bad argument");
  u256 w = mont_inv<Fr>(host_root_of_unity(k + ext_k));
  return ntt_dev(as_u256(ext_dev), nullptr, n_cols, k + ext_k, w, true, false, 0, nullptr, 0, true, nullptr);
}
// The cosets of the extended domain one by one ("slots"): slot t of a column is the polynomial on g_t H, H the 2^k-th roots of unity,
// g_t = zeta w_{4n}^(bitrev2(t)) — g_0 = zeta, g_1 = zeta w_{2n}, g_2 = zeta w_{4n}, g_3 = zeta w_{4n}^3: together the 4 n points of
// coeff_to_extended (slot t, row r = point 4 r + bitrev2(t) of the natural order), the first two the coset of 2 n points.  A quotient of
// degree below 3 n is determined on three of them, one of degree below 2 n on two: a quarter / a half of the transform is not made at all.
int vdb_coeff_to_cosets_dev(const vdb_fr* coeff_dev, vdb_fr* cosets_dev, size_t n_cols, uint32_t k, uint32_t n_slots, const vdb_fr* scale_or_null) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && cosets_dev && k + 2 <= 26 && n_slots >= 1 && n_slots <= 4, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& c = ctx();
  const uint64_t n = 1ull << k;
  const u256 m32 = host_fr_from_u64(32);
  const u256* tabs[4] = {nullptr, nullptr, nullptr, nullptr};
  int err = VDB_OK;
  if (scale_or_null) {
    // s p(X) on the cosets: the scalar rides on the input factors — tables of this call alone
    u256 sv;
    memcpy(&sv, scale_or_null, 32);
    u256* buf = (u256*)scratch_get(6, (size_t)n_slots * n * sizeof(u256));
    if (!buf) return VDB_ERR_OOM;
    const uint32_t chunk = 16;
    const uint64_t threads = (n + chunk - 1) / chunk;
    for (uint32_t t = 0; t < n_slots; t++) {
      hipLaunchKernelGGL(k_twiddles, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, buf + t * n, host_coset_shift(k, t), fr_mul(m32, sv), n, chunk);
      tabs[t] = buf + t * n;
    }
    VDB_LAUNCH_CHECK();
  } else {
    for (uint32_t t = 0; t < n_slots; t++) {
      tabs[t] = get_twiddles(k, host_coset_shift(k, t), m32, false, &err);
      if (!tabs[t]) return err;
    }
  }
  return ntt_dev(const_cast<u256*>(as_u256(coeff_dev)), as_u256(cosets_dev), n_cols * n_slots, k, host_root_of_unity(k), false, false, 0, nullptr, 0, false, nullptr,
                 tabs, n_slots);
}
// The way back for ONE polynomial, the quotient: `cosets_dev` holds the numerator's values on the first n_slots cosets ([slot][row];
// overwritten); they are divided by X^n - 1 (a constant per coset), brought back to the residues modulo X^n - g_t^n, and the n_slots
// pieces h_0 .. of n coefficients each of h = sum_j X^(n j) h_j — a polynomial of degree below n_slots n, which is what the caller
// asserts — are solved from the Vandermonde system in u_t = g_t^n (h mod (X^n - u_t) = sum_j u_t^j h_j).
int vdb_cosets_to_coeff_dev(vdb_fr* cosets_dev, vdb_fr* coeff_dev, uint32_t k, uint32_t n_slots) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cosets_dev && coeff_dev && k + 2 <= 26 && n_slots >= 1 && n_slots <= 4, "bad argument");
  Context& c = ctx();
  const uint64_t n = 1ull << k;
  int rc = ntt_dev(as_u256(cosets_dev), nullptr, n_slots, k, mont_inv<Fr>(host_root_of_unity(k)), true, false, 0, nullptr, 0, false, nullptr);
  if (rc) return rc;
  CosetMix mx;
  memset(&mx, 0, sizeof(mx));
  u256 u[4];
  const u256 one = mont_one<Fr>();
  int err = VDB_OK;
  for (uint32_t t = 0; t < n_slots; t++) {
    const u256 g = host_coset_shift(k, t);
    // (tables of plain powers: cached under the base g^-1, which no transform uses as its root)
    mx.tab[t] = get_twiddles(k, mont_inv<Fr>(g), one, false, &err);
    if (!mx.tab[t]) return err;
    u[t] = g;
    for (uint32_t i = 0; i < k; i++) u[t] = fr_mul(u[t], u[t]);
  }
  // column t of the inverse Vandermonde matrix = the coefficients of the Lagrange polynomial L_t over the nodes u; times 1 / (u_t - 1)
  for (uint32_t t = 0; t < n_slots; t++) {
    u256 poly[5] = {one, u256_zero(), u256_zero(), u256_zero(), u256_zero()};  // prod_{s != t} (X - u_s)
    uint32_t deg = 0;
    u256 den = fr_sub(u[t], one);                                               // (u_t - 1) prod_{s != t} (u_t - u_s)
    for (uint32_t s2 = 0; s2 < n_slots; s2++) {
      if (s2 == t) continue;
      for (int d = (int)deg + 1; d >= 0; d--) {
        u256 lower = d > 0 ? poly[d - 1] : u256_zero();
        poly[d] = fr_sub(lower, d <= (int)deg ? fr_mul(poly[d], u[s2]) : u256_zero());
      }
      deg++;
      den = fr_mul(den, fr_sub(u[t], u[s2]));
    }
    const u256 inv = mont_inv<Fr>(den);
    for (uint32_t j = 0; j < n_slots; j++) mx.c[j][t] = fr_mul(poly[j], inv);
  }
  {
    VDB_PROF("k_cosets_combine");
    hipLaunchKernelGGL(k_cosets_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, as_u256(cosets_dev), n, n_slots, mx, as_u256(coeff_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_coeff_to_extended(const vdb_fr* const* coeff_cols, vdb_fr* const* ext_cols, size_t n_cols, uint32_t k, uint32_t ext_k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_cols && ext_cols, "null pointer");
  if (n_cols == 0) return VDB_OK;
  Context& c = ctx();
  size_t n = (size_t)1 << k, ne = (size_t)1 << (k + ext_k);
  u256* din = (u256*)scratch_get(0, n_cols * n * sizeof(u256));
  u256* dout = (u256*)scratch_get(1, n_cols * ne * sizeof(u256));
  if (!din || !dout) return VDB_ERR_OOM;
  for (size_t i = 0; i < n_cols; i++) VDB_HIP(hipMemcpyAsync(din + i * n, coeff_cols[i], n * sizeof(u256), hipMemcpyHostToDevice, c.stream));
  int rc = vdb_coeff_to_extended_dev(reinterpret_cast<vdb_fr*>(din), reinterpret_cast<vdb_fr*>(dout), n_cols, k, ext_k);
  if (rc) return rc;
  for (size_t i = 0; i < n_cols; i++) VDB_HIP(hipMemcpyAsync(ext_cols[i], dout + i * ne, ne * sizeof(u256), hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

}  // extern "C"
