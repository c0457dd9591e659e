// Prover-round primitives that follow commit + NTT (first brick): the batched grand product
//     z[0] = 1,   z[i + 1] = z[i] * num[i] / den[i]
// of the permutation and lookup arguments (halo2 plonk/permutation/prover.rs, plonk/lookup/prover.rs: the running products
// over rows of prod(value + beta * sigma + gamma) ratios, [UPSTREAM-RECALL]).  A zero denominator inverts to zero, as
// halo2's batch_invert leaves it, so the product is zero from that row on.
//
// One 1024-thread workgroup per column; thread t owns the contiguous rows [t E, (t + 1) E).  With
//     N_i = prod_{j <= i} num_j,   S_i = prod_{j > i} den_j,   D = prod_j den_j   (j over the n - 1 rows that enter z)
// z[i + 1] = N_i * S_i / D: ONE field inversion per column, five products per row, no per-row inversion.
//   pass 1: per-thread totals of num and den (streaming), block-wide exclusive prefix (num) and suffix (den) products;
//   pass 2a: rows backwards — S_i / D into z[i + 1];  pass 2b: rows forwards — times N_i.
// Roofline: 224 B of HBM traffic and 5 products per row — about balanced between HBM and the integer ALU.
#include "common.hpp"
#include "limb9.hpp"

namespace vdb {

#define GP_THREADS 1024
#define GP_PF 4

// inclusive scan (products) of one u256 per thread over the block, in LDS; `rev` scans from the last thread down
__device__ __forceinline__ u256 block_scan_mul(u256 v, u256* sh, bool rev) {
  const uint32_t t = rev ? GP_THREADS - 1 - threadIdx.x : threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (uint32_t o = 1; o < GP_THREADS; o <<= 1) {
    u256 other = t >= o ? sh[t - o] : mont_one<Fr>();
    __syncthreads();
    if (t >= o) {
      v = fr_mul(v, other);
      sh[t] = v;
    }
    __syncthreads();
  }
  return v;  // product of the values of threads 0 .. t (in scan order)
}

__global__ __launch_bounds__(GP_THREADS) void k_grand_product(const u256* __restrict__ num, const u256* __restrict__ den, u256* __restrict__ z, uint64_t n,
                                                              uint64_t stride) {
  __shared__ u256 sh[GP_THREADS];
  __shared__ u256 s_dinv;
  const uint64_t col = blockIdx.x;
  const u256* nu = num + col * stride;
  const u256* de = den + col * stride;
  u256* zo = z + col * stride;
  const uint64_t rows = n - 1;  // rows that enter the product
  const uint64_t E = (rows + GP_THREADS - 1) / GP_THREADS;
  const uint64_t lo = (uint64_t)threadIdx.x * E < rows ? (uint64_t)threadIdx.x * E : rows;
  const uint64_t hi = lo + E < rows ? lo + E : rows;
  const u256 one = mont_one<Fr>();
  // pass 1: thread totals
  // (operands are fetched GP_PF rows ahead of the dependent product chains: one load per step on the chain is pure latency)
  u256 nt = one, dt = one;
  for (uint64_t i0 = lo; i0 < hi; i0 += GP_PF) {
    u256 dv[GP_PF], nv[GP_PF];
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) {
      const uint64_t i = i0 + q < hi ? i0 + q : hi - 1;
      dv[q] = ld256(de + i);
      nv[q] = ld256(nu + i);
    }
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) {
      if (i0 + q >= hi) break;
      if (u256_is_zero(dv[q])) {
        nt = u256_zero();  // this row's ratio is zero
      } else {
        dt = fr_mul(dt, dv[q]);
        nt = fr_mul(nt, nv[q]);
      }
    }
  }
  const u256 n_inc = block_scan_mul(nt, sh, false);  // prod of totals of threads <= t
  __syncthreads();
  const u256 n_pre = threadIdx.x ? sh[threadIdx.x - 1] : one;  // exclusive
  __syncthreads();
  const u256 d_inc = block_scan_mul(dt, sh, true);   // prod of totals of threads >= t
  __syncthreads();
  const u256 d_suf = threadIdx.x + 1 < GP_THREADS ? sh[GP_THREADS - 1 - (threadIdx.x + 1)] : one;  // threads > t
  if (threadIdx.x == 0) s_dinv = mont_inv<Fr>(d_inc);  // d_inc of thread 0 = D (never zero: zero rows were replaced by one)
  (void)n_inc;
  __syncthreads();
  // pass 2a: backwards, z[i + 1] = S_i / D
  u256 s = fr_mul(d_suf, s_dinv);
  for (uint64_t i0 = hi; i0 > lo; i0 -= (i0 - lo < GP_PF ? i0 - lo : GP_PF)) {
    u256 dv[GP_PF];
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) dv[q] = ld256(de + (i0 - 1 - q >= lo && i0 - 1 >= q ? i0 - 1 - q : lo));
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) {
      if (i0 < lo + q + 1) break;
      const uint64_t i = i0 - q;
      st256(zo + i, s);  // row i - 1 -> z[i]
      if (!u256_is_zero(dv[q])) s = fr_mul(s, dv[q]);
    }
  }
  // pass 2b: forwards, times N_i
  u256 acc = n_pre;
  for (uint64_t i0 = lo; i0 < hi; i0 += GP_PF) {
    u256 dv[GP_PF], nv[GP_PF], zv[GP_PF];
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) {
      const uint64_t i = i0 + q < hi ? i0 + q : hi - 1;
      dv[q] = ld256(de + i);
      nv[q] = ld256(nu + i);
      zv[q] = ld256(zo + i + 1);
    }
#pragma unroll
    for (uint32_t q = 0; q < GP_PF; q++) {
      if (i0 + q >= hi) break;
      acc = u256_is_zero(dv[q]) ? u256_zero() : fr_mul(acc, nv[q]);
      st256(zo + i0 + q + 1, fr_mul(zv[q], acc));
    }
  }
  if (threadIdx.x == 0) st256(zo, one);
}

// ---- evaluation of coefficient-form polynomials at one point (halo2 arithmetic::eval_polynomial, used for every advice
// polynomial when the prover opens at x, [UPSTREAM-RECALL]) -----------------------------------------------------------
// out[c] = sum_i a[c][i] x^i.  One 256-thread workgroup per polynomial; thread t owns the coefficients i = t (mod 256)
// — consecutive lanes read consecutive coefficients, whole lines — and runs Horner in y = x^256:
//     P_t = sum_k a[t + 256 k] y^k,      out = sum_t x^t P_t   (x^t by square-and-multiply, LDS tree for the sum).
// Roofline: 32 B per coefficient, one product and one addition: HBM bound in principle (17 GB for C4's columns in
// 2.7 ms at 6.3 TB/s), about 4.5 ms of integer ALU at the measured product rate — the two are close.
#define EV_THREADS 256
// The Horner accumulator lives in nine-limb form (limb9.hpp): acc * y is one 9 x 29-bit product with the pre-scaled
// constant 32 y, the coefficient is added without carries (limbs stay below 2 * 2^29, values below 2.2 r: no carry pass
// is ever needed), so a step costs ~250 instructions instead of ~350.
__global__ __launch_bounds__(EV_THREADS) void k_eval_polys(const u256* __restrict__ coeff, uint64_t n, u256 x, u256 y32 /* 32 x^EV_THREADS */,
                                                           u256* __restrict__ out) {
  __shared__ u256 sh[EV_THREADS];
  const uint64_t col = blockIdx.x;
  const u256* a = coeff + col * n;
  const uint32_t t = threadIdx.x;
  u256 acc = u256_zero();
  if (t < n) {
    const L9 Y = l9_split(y32);
    L9 A;
#pragma unroll
    for (int k9 = 0; k9 < 9; k9++) A.l[k9] = 0;
    // highest k with t + 256 k < n
    uint64_t k = (n - 1 - t) / EV_THREADS;
    for (;;) {
      // (coefficients are fetched four steps ahead of the dependent Horner chain)
      u256 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) v[q] = ld256(a + t + EV_THREADS * (k >= q ? k - q : 0));
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        if (k < q) break;
        A = l9_add(l9_mul<Fr>(A, Y), l9_split(v[q]));
      }
      if (k < 4) break;
      k -= 4;
    }
    const u256 xt32 = fr_mul(mont_pow<Fr>(x, u256_from_u64(t)), to_mont<Fr>(u256_from_u64(32)));
    acc = l9_canon<Fr>(l9_mul<Fr>(A, l9_split(xt32)));
  }
  sh[t] = acc;
  __syncthreads();
  for (uint32_t o = EV_THREADS / 2; o >= 1; o >>= 1) {
    if (t < o) sh[t] = fr_add(sh[t], sh[t + o]);
    __syncthreads();
  }
  if (t == 0) st256(out + col, sh[0]);
}

// ---- lookup argument: permuted input / table columns (halo2 plonk/lookup/prover.rs permute_expression_pair,
// [UPSTREAM-RECALL]) ------------------------------------------------------------------------------------------------
// Over the usable rows: A' = the input values sorted ascending (canonical integer order); S' holds, at the first row of
// every run of equal A' values, that value (taken out of the table's multiset), and at the other ("repeated") rows the
// table values that are left over, in ascending order handed to the repeated rows from the LAST one backwards
// (upstream pops them off a stack).  Then A'[i] == S'[i] or A'[i] == A'[i - 1] on every row, which is what the lookup
// argument's grand product checks.
// halo2-base only looks values up in a range table (cells below 2^lookup_bits), so the sort is a counting
// sort over 2^max_bits bins: no comparison sort of 254-bit keys.  One 1024-thread workgroup per input column.
#define LP_THREADS 1024
__device__ __forceinline__ bool small_canonical(const u256& mont, uint32_t max_bits, uint32_t* v) {
  const u256 c = from_mont<Fr>(mont);
  uint32_t hi = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) hi |= c.w[i];
  *v = c.w[0];
  return hi == 0 && (max_bits >= 32 || (c.w[0] >> max_bits) == 0);
}
__global__ __launch_bounds__(256) void k_lp_hist(const u256* __restrict__ col, uint64_t stride, uint64_t usable, uint32_t max_bits, uint32_t bins,
                                                 uint32_t* __restrict__ hist, int* __restrict__ err) {
  const u256* c = col + (uint64_t)blockIdx.y * stride;
  uint32_t* h = hist + (uint64_t)blockIdx.y * bins;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < usable; r += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v;
    if (small_canonical(ld256(c + r), max_bits, &v)) atomicAdd(&h[v], 1u);
    else atomicOr(err, 1);
  }
}
// block-wide exclusive scan of one u32 per thread
__device__ __forceinline__ uint32_t lp_block_scan(uint32_t v, uint32_t* sh, uint32_t* total) {
  const uint32_t t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (uint32_t o = 1; o < LP_THREADS; o <<= 1) {
    uint32_t x = t >= o ? sh[t - o] : 0u;
    __syncthreads();
    sh[t] += x;
    __syncthreads();
  }
  const uint32_t inc = sh[t];
  *total = sh[LP_THREADS - 1];
  __syncthreads();
  return inc - v;
}
// per column: off[v] = rows before the run of v, dist[v] = distinct input values <= v, lpre[v] = left-over table values < v
__global__ __launch_bounds__(LP_THREADS) void k_lp_scan(const uint32_t* __restrict__ in_hist, const uint32_t* __restrict__ tab_hist, uint32_t bins,
                                                        uint32_t* __restrict__ off, uint32_t* __restrict__ dist, uint32_t* __restrict__ lpre,
                                                        uint32_t* __restrict__ totals /* per column: repeated rows, left-over */, int* __restrict__ err) {
  __shared__ uint32_t sh[LP_THREADS];
  const uint64_t col = blockIdx.x;
  const uint32_t* ih = in_hist + col * bins;
  const uint32_t per = (bins + LP_THREADS - 1) / LP_THREADS, lo = threadIdx.x * per, hi = lo + per < bins ? lo + per : bins;
  uint32_t c_sum = 0, d_sum = 0, l_sum = 0;
  for (uint32_t v = lo; v < hi; v++) {
    const uint32_t c = ih[v], t = tab_hist[v];
    c_sum += c;
    d_sum += c ? 1u : 0u;
    if (c && !t) atomicOr(err, 2);  // an input value that is not in the table: the reference panics
    l_sum += t - (c && t ? 1u : 0u);
  }
  uint32_t tot_c, tot_d, tot_l;
  uint32_t c_pre = lp_block_scan(c_sum, sh, &tot_c);
  uint32_t d_pre = lp_block_scan(d_sum, sh, &tot_d);
  uint32_t l_pre = lp_block_scan(l_sum, sh, &tot_l);
  for (uint32_t v = lo; v < hi; v++) {
    const uint32_t c = ih[v], t = tab_hist[v];
    off[col * bins + v] = c_pre;
    lpre[col * bins + v] = l_pre;
    c_pre += c;
    d_pre += c ? 1u : 0u;
    dist[col * bins + v] = d_pre;
    l_pre += t - (c && t ? 1u : 0u);
  }
  if (threadIdx.x == 0) {
    totals[2 * col] = tot_c - tot_d;  // repeated rows
    totals[2 * col + 1] = tot_l;      // left-over table values: must be the same number
    if (tot_c - tot_d != tot_l) atomicOr(err, 4);
  }
}
// largest index v in [0, bins) with a[v] <= x  (a non-decreasing, a[0] <= x)
__device__ __forceinline__ uint32_t lp_upper(const uint32_t* __restrict__ a, uint32_t bins, uint32_t x) {
  uint32_t lo = 0, hi = bins;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid;
    else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void k_lp_rows(const uint32_t* __restrict__ in_hist, const uint32_t* __restrict__ tab_hist, const uint32_t* __restrict__ off,
                                                 const uint32_t* __restrict__ dist, const uint32_t* __restrict__ lpre, const uint32_t* __restrict__ totals,
                                                 uint32_t bins, uint64_t usable, uint64_t n, u256* __restrict__ out_in, u256* __restrict__ out_tab) {
  const uint64_t col = blockIdx.y;
  const uint32_t *o = off + col * bins, *d = dist + col * bins, *lp = lpre + col * bins, *ih = in_hist + col * bins;
  const uint32_t R = totals[2 * col];
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
    u256 a = u256_zero(), t = u256_zero();
    if (r < usable) {
      // run of r: the last value v with off[v] <= r that actually occurs
      uint32_t v = lp_upper(o, bins, (uint32_t)r);
      while (ih[v] == 0) v--;  // empty bins share their successor's offset: step back to the occupied one
      a = to_mont<Fr>(u256_from_u64(v));
      if ((uint32_t)r == o[v]) {
        t = a;  // first row of the run takes the value itself out of the table
      } else {
        const uint32_t rank = (uint32_t)r - d[v];        // repeated rows before this one
        const uint32_t k = R - 1 - rank;                 // left-over values are handed out from the last repeated row backwards
        uint32_t u = lp_upper(lp, bins, k);
        // bins without left-over share their successor's prefix: the owner of item k is the last bin with lpre <= k that has any
        while (tab_hist[u] - (ih[u] && tab_hist[u] ? 1u : 0u) == 0) u--;
        t = to_mont<Fr>(u256_from_u64(u));
      }
    }
    st256(out_in + col * n + r, a);
    st256(out_tab + col * n + r, t);
  }
}

// ---- gate part of the quotient numerator on the extended coset ---------------------------------------------------------
// halo2-base's one custom gate (FlexGateConfig, "vertical" strategy): for every advice column i with selector q_i,
//     q_i(X) * (a_i(X) + a_i(wX) * a_i(w^2 X) - a_i(w^3 X)) = 0 on the rows,
// i.e. cells [a, b, c, d] in four consecutive rows satisfy a + b c = d where q = 1.  The prover accumulates the gates
// with powers of the challenge y (halo2 evaluates them column after column by Horner's rule: acc = acc * y + gate,
// [UPSTREAM-RECALL] for the order only).  On the extended coset of size 2^(k+e) a rotation by one row is a step of 2^e.
// One thread per extended row, columns in a loop: consecutive lanes read consecutive rows of the same column.
// The accumulator and the gate value live in nine-limb form (limb9.hpp): b c is one product with the second factor
// re-limbed as 32 c, a + b c - d is added without carries (limbs below 4.1 * 2^29, value below 5 r), and h y + q g is ONE
// reduction over two products (l9_mul2: limbs of h and g together below 6 * 2^29; the result is normalised and below
// (2 r * r + 5 r * 32 r) / 2^261 + r < 2 r).  2.4 reductions' worth per cell instead of 3 full field products.
struct GateK {
  uint32_t c2[9];  // 2 r written with dominating limbs (l9_offset_limbs): subtracting a canonical value
};
// `sub` > 0: the advice cosets live on 2^(log_ne + sub) points and the gate is evaluated on every 2^sub-th of them — the gate
// has degree 3, so its share of the quotient is determined on the coset of 2 n points inside the 4 n the permutation needs.
// Coset by coset ("slots", vdb_coeff_to_cosets_dev): blockIdx.y is the slot, every array is offset by slot * ne, and the column strides
// (adv_cs, sel_cs: slots per column * ne) are no longer the number of points; a caller of the natural-order layout passes adv_cs = ne_a,
// sel_cs = ne and one slot.
__global__ __launch_bounds__(256) void k_gate_eval(const u256* __restrict__ adv, const u256* __restrict__ sel, uint64_t n_cols, uint32_t log_ne, uint32_t e,
                                                   uint32_t sub, uint64_t adv_cs, uint64_t sel_cs, u256 y32 /* 32 y */, GateK gk, u256* __restrict__ acc) {
  const uint64_t ne = 1ull << log_ne, j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ne) return;
  const uint64_t ne_a = ne << sub, mask = ne_a - 1, r = 1ull << (e + sub), ja = j << sub;
  const uint64_t so = (uint64_t)blockIdx.y * ne;
  adv += so;
  sel += so;
  acc += so;
  const L9 Y = l9_split(y32);
  L9 h = l9_split(ld256(acc + j));
  for (uint64_t c = 0; c < n_cols; c++) {
    const u256* a = adv + c * adv_cs;
    const L9 q32 = l9_split32(ld256(sel + c * sel_cs + j));
    const L9 bc = l9_mul<Fr>(l9_split(ld256(a + ((ja + r) & mask))), l9_split32(ld256(a + ((ja + 2 * r) & mask))));
    const L9 g = l9_sub(l9_add(l9_split(ld256(a + ja)), bc), l9_split(ld256(a + ((ja + 3 * r) & mask))), gk.c2);
    h = l9_mul2<Fr>(h, Y, g, q32);
  }
  st256(acc + j, l9_canon<Fr>(h));
}
// h[j] *= t[j mod 2^e],  t[m] = 1 / (zeta^n * w_{2^e}^m - 1): division by the vanishing polynomial X^n - 1 on the coset
__global__ __launch_bounds__(256) void k_mul_periodic(u256* __restrict__ h, uint64_t ne, const u256* __restrict__ t, uint32_t period_mask) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ne) return;
  st256(h + j, fr_mul(ld256(h + j), ld256(t + (j & period_mask))));
}

// ---- product terms of the permutation and lookup arguments (halo2 plonk/permutation/prover.rs and
// plonk/lookup/prover.rs commit_product, [UPSTREAM-RECALL] for the formulas) -------------------------------------------
// pw[i] = beta * omega^i: one square-and-multiply per row, once per call
__global__ __launch_bounds__(256) void k_beta_omega_powers(u256 omega, u256 beta, uint64_t n, u256* __restrict__ pw) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u256 acc = beta, b = omega;
  for (uint64_t e = i; e; e >>= 1) {
    if (e & 1) acc = fr_mul(acc, b);
    b = fr_mul(b, b);
  }
  st256(pw + i, acc);
}
// sigma[c][row] = delta^c' * omega^row' for the cell (c', row') the permutation sends (c, row) to; map = c' << 32 | row'
__global__ __launch_bounds__(256) void k_perm_sigma(const uint64_t* __restrict__ map, uint64_t n_cols, uint64_t n, const u256* __restrict__ wpow /* omega^i */,
                                                    const u256* __restrict__ dpow /* delta^c */, u256* __restrict__ sigma, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cols * n) return;
  const uint64_t m = map[i], c = m >> 32, r = m & 0xffffffffull;
  if (c >= n_cols || r >= n) {
    *err = 1;
    return;
  }
  st256(sigma + i, fr_mul(ld256(dpow + c), ld256(wpow + r)));
}
// The same from the packed mapping (c' << k | row' in 32 bits) for a block of columns: what the prover's product round reads per proof
// — one product per cell instead of a transform per sigma column.
__global__ __launch_bounds__(256) void k_perm_map_pack(const uint64_t* __restrict__ map, uint64_t cells, uint32_t k, uint32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const uint64_t m = map[i];
  out[i] = (uint32_t)((m >> 32) << k) | (uint32_t)(m & 0xffffffffull);
}
__global__ __launch_bounds__(256) void k_perm_sigma_packed(const uint32_t* __restrict__ map, uint64_t cells, uint64_t n_cols_total, uint32_t k,
                                                           const u256* __restrict__ wpow, const u256* __restrict__ dpow, u256* __restrict__ sigma,
                                                           int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const uint32_t m = map[i], c = m >> k, r = m & ((1u << k) - 1u);
  if (c >= n_cols_total) {
    *err = 1;
    return;
  }
  st256(sigma + i, fr_mul(ld256(dpow + c), ld256(wpow + r)));
}
// One thread per (row, chunk of columns):  num = prod_c (v_c + delta^c * beta * omega^row + gamma),
//                                          den = prod_c (v_c + beta * sigma_c[row] + gamma).
__global__ __launch_bounds__(256) void k_perm_terms(const u256* __restrict__ cols, const u256* __restrict__ sigma, uint64_t n_cols, uint64_t n, uint64_t rows,
                                                    uint32_t chunk_len, const u256* __restrict__ bw /* beta omega^row */, u256 beta, u256 gamma, u256 delta,
                                                    const u256* __restrict__ dstart /* delta^(chunk * chunk_len) */, u256* __restrict__ num,
                                                    u256* __restrict__ den) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, chunk = blockIdx.y;
  if (row >= rows) return;
  const uint64_t c0 = chunk * chunk_len, c1 = c0 + chunk_len < n_cols ? c0 + chunk_len : n_cols;
  u256 cur = fr_mul(ld256(bw + row), ld256(dstart + chunk));
  u256 nu = mont_one<Fr>(), de = nu;
  for (uint64_t c = c0; c < c1; c++) {
    const u256 v = fr_add(ld256(cols + c * n + row), gamma);
    de = fr_mul(de, fr_add(v, fr_mul(beta, ld256(sigma + c * n + row))));
    nu = fr_mul(nu, fr_add(v, cur));
    cur = fr_mul(cur, delta);
  }
  st256(num + chunk * n + row, nu);
  st256(den + chunk * n + row, de);
}
// num = (A + beta)(S + gamma),  den = (A' + beta)(S' + gamma); the table column is shared by all input columns
__global__ __launch_bounds__(256) void k_lookup_terms(const u256* __restrict__ a, const u256* __restrict__ tab, const u256* __restrict__ pa, const u256* __restrict__ pt,
                                                      uint64_t n, uint64_t rows, u256 beta, u256 gamma, u256* __restrict__ num, u256* __restrict__ den) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
  if (row >= rows) return;
  const uint64_t i = col * n + row;
  st256(num + i, fr_mul(fr_add(ld256(a + i), beta), fr_add(ld256(tab + row), gamma)));
  st256(den + i, fr_mul(fr_add(ld256(pa + i), beta), fr_add(ld256(pt + i), gamma)));
}
// The permutation product runs on from one chunk of columns to the next: z_c starts where z_{c-1} ended.  The chunks'
// products are computed independently from one; f[c] = prod_{c' < c} z_c'[last] puts them on one chain.
__global__ __launch_bounds__(GP_THREADS) void k_chain_factors(const u256* __restrict__ z, uint64_t n_chunks, uint64_t stride, uint64_t last, u256* __restrict__ f) {
  __shared__ u256 sh[GP_THREADS];
  const uint64_t E = (n_chunks + GP_THREADS - 1) / GP_THREADS;
  const uint64_t lo = (uint64_t)threadIdx.x * E < n_chunks ? (uint64_t)threadIdx.x * E : n_chunks;
  const uint64_t hi = lo + E < n_chunks ? lo + E : n_chunks;
  u256 t = mont_one<Fr>();
  for (uint64_t c = lo; c < hi; c++) t = fr_mul(t, ld256(z + c * stride + last));
  block_scan_mul(t, sh, false);
  __syncthreads();
  u256 acc = threadIdx.x ? sh[threadIdx.x - 1] : mont_one<Fr>();
  for (uint64_t c = lo; c < hi; c++) {
    st256(f + c, acc);
    acc = fr_mul(acc, ld256(z + c * stride + last));
  }
}
__global__ __launch_bounds__(256) void k_scale_columns(u256* __restrict__ z, uint64_t stride, uint64_t rows, const u256* __restrict__ f) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y + 1;  // column 0 has factor one
  if (row >= rows) return;
  st256(z + col * stride + row, fr_mul(ld256(z + col * stride + row), ld256(f + col)));
}
__global__ __launch_bounds__(256) void k_zero_tail(u256* __restrict__ z, uint64_t stride, uint64_t from, uint64_t n) {
  const uint64_t row = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
  if (row >= n) return;
  st256(z + col * stride + row, u256_zero());
}

// ---- permutation and lookup parts of the quotient numerator on the extended coset (halo2 plonk/evaluation.rs evaluate_h,
// [UPSTREAM-RECALL] for the order of the terms; each term is folded in as acc = acc * y + term) --------------------------
// Rotation by one row = a step of 2^e on the coset of 2^(k+e) points.  l0, l_last, l_active = 1 - (l_last + l_blind) are the
// Lagrange selectors on the coset; bx[j] = beta * X_j.  One thread per extended row, sets / columns in a loop.
// All arithmetic in nine-limb form (limb9.hpp) on representatives scaled by 2^261 instead of 2^256: a value loaded from
// memory (x 2^256, below r) enters as 32 times itself — the same bits re-limbed (l9_split32), no arithmetic —, constants are
// scaled on the host, a nine-limb product of two such values is again such a value, sums and differences add limbs without
// carries, and `h y + l t` is one reduction over two products (l9_mul2).  One product by 2^256 per thread at the very end
// returns to the memory form.  Bounds (value in multiples of r, before the + r of a product): a loaded value is below 32,
// a product of two loaded values below 32 * 32 * r / 2^261 = 6.1, any product with a reduced factor far below that; the
// offsets added by subtractions (c34 = 34 r when a loaded value is subtracted, c9 = 9 r for products) keep limbs below
// 3 * 2^29 and values below 70 r, all far below 2^261 = 169 r.
struct QuotArgs {
  const u256 *l0, *l_last, *l_active;
  u256 beta, gamma, delta, y;  // scaled: 32 * value
  uint32_t log_ne, e;
  uint64_t cs;        // column stride of the coset arrays: ne in the natural-order layout, slots per column * ne coset by coset (blockIdx.y = slot)
  uint64_t last_rot;  // rows between the last usable row and the end: n - usable_rows
  uint32_t c9[9], c34[9];
};
// Sets [set_lo, set_hi) of the product terms; the terms that involve only the product columns come with set_lo == 0.
// `sigma` points at the first column of set_lo (the sigma cosets may be produced block by block); bx = beta X, dstart =
// 32 delta^(set_lo chunk_len).
// Streaming form: `adv` holds the cosets of columns adv_col0 .., `z` those of product sets z_set0 .., `z_first` / `z_last` the
// cosets of the first and the last product set (one column each); `head`: fold in the two terms that read only those;
// [chain_lo, chain_hi): the sets i whose chaining term l0 (z_i - z_{i-1}(..)) is folded in (needs sets chain_lo - 1 .. chain_hi - 1
// in `z`); [set_lo, set_hi): the sets whose product term is folded in.  A resident caller passes whole arrays and zero offsets.
struct PermParts {
  uint64_t adv_col0, z_set0, chain_lo, chain_hi;
  const u256 *z_first, *z_last;
  int head;          // bit 0: fold in the two terms of the first / last product alone; bit 1: the sigma cosets are those of beta sigma
};
__global__ __launch_bounds__(256) void k_perm_eval(const u256* __restrict__ adv, const u256* __restrict__ sigma, const u256* __restrict__ z, uint64_t n_cols,
                                                   uint32_t chunk_len, uint64_t set_lo, uint64_t set_hi, const u256* __restrict__ bx, u256 dstart, QuotArgs q,
                                                   PermParts pp, u256* __restrict__ acc) {
  const uint64_t ne = 1ull << q.log_ne, j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ne) return;
  const uint64_t mask = ne - 1, r = 1ull << q.e, cs = q.cs, so = (uint64_t)blockIdx.y * ne;
  const uint64_t n_sets = (n_cols + chunk_len - 1) / chunk_len;
  const L9 Y = l9_split(q.y), B = l9_split(q.beta), G = l9_split(q.gamma), D = l9_split(q.delta);
  const L9 L0 = l9_split32(ld256(q.l0 + so + j)), LL = l9_split32(ld256(q.l_last + so + j)), LA = l9_split32(ld256(q.l_active + so + j));
  acc += so;
  bx += so;
  sigma += so;
  pp.z_first += so;
  pp.z_last += so;
  L9 h = l9_split32(ld256(acc + j));
  (void)n_sets;
  z += so;
  adv += so;
  z -= pp.z_set0 * cs;          // index by global set number from here on (only sets the caller provides are touched)
  adv -= pp.adv_col0 * cs;
  if (pp.head & 1) {
    // l0 (1 - z_0)
    h = l9_mul2<Fr>(h, Y, l9_sub(l9_split32(mont_one<Fr>()), l9_split32(ld256(pp.z_first + j)), q.c34), L0);
    // l_last (z_last^2 - z_last)
    {
      const L9 zl = l9_split32(ld256(pp.z_last + j));
      h = l9_mul2<Fr>(h, Y, l9_sub(l9_mul<Fr>(zl, zl), zl, q.c34), LL);
    }
  }
  if (pp.chain_lo < pp.chain_hi) {
    // l0 (z_i - z_{i-1}(w^-(blinding+1) X)): every set starts where the one before ended
    const uint64_t jb = (j + ne - ((q.last_rot << q.e) & mask)) & mask;
    for (uint64_t i = pp.chain_lo; i < pp.chain_hi; i++)
      h = l9_mul2<Fr>(h, Y, l9_sub(l9_split32(ld256(z + i * cs + j)), l9_split32(ld256(z + (i - 1) * cs + jb)), q.c34), L0);
  }
  // l_active (z_i(w X) prod (v + beta sigma + gamma) - z_i(X) prod (v + delta^c beta X + gamma))
  L9 cur = l9_mul<Fr>(l9_split32(ld256(bx + j)), l9_split(dstart));
  const uint64_t cb = set_lo * chunk_len;  // first column of the sigma block
  for (uint64_t i = set_lo; i < set_hi; i++) {
    const uint64_t c0 = i * chunk_len, c1 = c0 + chunk_len < n_cols ? c0 + chunk_len : n_cols;
    L9 left = l9_split32(ld256(z + i * cs + ((j + r) & mask))), right = l9_split32(ld256(z + i * cs + j));
    for (uint64_t c = c0; c < c1; c++) {
      const L9 v = l9_add(l9_split32(ld256(adv + c * cs + j)), G);
      // (beta sigma: one product per column and extended row, unless the caller extended beta sigma(X) in the first place —
      //  vdb_coeff_to_extended_scaled_dev: the scalar then costs a third of a product per BASE row)
      const L9 sg = l9_split32(ld256(sigma + (c - cb) * cs + j));
      left = l9_mul<Fr>(l9_add(v, (pp.head & 2) ? sg : l9_mul<Fr>(sg, B)), left);
      right = l9_mul<Fr>(l9_add(v, cur), right);
      cur = l9_mul<Fr>(cur, D);
    }
    h = l9_mul2<Fr>(h, Y, l9_sub(left, right, q.c9), LA);
  }
  st256(acc + j, l9_canon<Fr>(l9_mul<Fr>(h, l9_split(mont_one<Fr>()))));
}
__global__ __launch_bounds__(256) void k_lookup_eval(const u256* __restrict__ a, const u256* __restrict__ tab, const u256* __restrict__ pa, const u256* __restrict__ pt,
                                                     const u256* __restrict__ z, uint64_t n_cols, QuotArgs q, u256 beta_m, u256 gamma_m /* memory form */,
                                                     u256* __restrict__ acc) {
  const uint64_t ne = 1ull << q.log_ne, j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ne) return;
  const uint64_t mask = ne - 1, r = 1ull << q.e, so = (uint64_t)blockIdx.y * ne;
  const L9 Y = l9_split(q.y), B = l9_split(q.beta);
  const L9 L0 = l9_split32(ld256(q.l0 + so + j)), LL = l9_split32(ld256(q.l_last + so + j)), LA = l9_split32(ld256(q.l_active + so + j));
  const L9 ONE = l9_split32(mont_one<Fr>());
  a += so;
  tab += so;
  pa += so;
  pt += so;
  z += so;
  acc += so;
  // second factors of a product must be normalised: sums that serve as one are made in memory form first
  const L9 sg = l9_split32(fr_add(ld256(tab + j), gamma_m));
  L9 h = l9_split32(ld256(acc + j));
  for (uint64_t c = 0; c < n_cols; c++) {
    const uint64_t o = c * q.cs;
    const L9 zc = l9_split32(ld256(z + o + j)), zn = l9_split32(ld256(z + o + ((j + r) & mask)));
    const u256 pav_m = ld256(pa + o + j), ptv_m = ld256(pt + o + j);
    const L9 av = l9_split32(ld256(a + o + j)), pav = l9_split32(pav_m), ptv = l9_split32(ptv_m);
    h = l9_mul2<Fr>(h, Y, l9_sub(ONE, zc, q.c34), L0);                                     // l0 (1 - z)
    h = l9_mul2<Fr>(h, Y, l9_sub(l9_mul<Fr>(zc, zc), zc, q.c34), LL);                      // l_last (z^2 - z)
    const L9 left = l9_mul<Fr>(l9_mul<Fr>(l9_add(pav, B), l9_split32(fr_add(ptv_m, gamma_m))), zn);
    const L9 right = l9_mul<Fr>(l9_mul<Fr>(l9_add(av, B), sg), zc);
    h = l9_mul2<Fr>(h, Y, l9_sub(left, right, q.c9), LA);                                  // l_active (z(wX)(a'+b)(s'+g) - z (a+b)(s+g))
    const L9 d = l9_sub(pav, ptv, q.c34);
    h = l9_mul2<Fr>(h, Y, d, L0);                                                          // l0 (a' - s')
    const L9 e = l9_split32(fr_sub(pav_m, ld256(pa + o + ((j + ne - r) & mask))));
    h = l9_mul2<Fr>(h, Y, l9_mul<Fr>(d, e), LA);                                           // l_active (a' - s')(a' - a'(w^-1 X))
  }
  (void)beta_m;
  st256(acc + j, l9_canon<Fr>(l9_mul<Fr>(h, l9_split(mont_one<Fr>()))));
}

// ---- opening proofs: division by a linear factor (halo2 arithmetic::kate_division, [UPSTREAM-RECALL]) and the linear
// combination of polynomials with powers of a challenge that precedes it -------------------------------------------------
// q(X) = (p(X) - p(x)) / (X - x):  S(i) = sum_{j >= i} a_j x^(j-i) satisfies S(i) = a_i + x S(i+1); q_{i-1} = S(i), p(x) = S(0).
// One 256-thread workgroup per polynomial walks it from the top in chunks of 256 x KD_E coefficients; thread t owns KD_E
// consecutive coefficients (whole cache lines), evaluates them (Horner), the block turns the 256 values into suffix sums with
// a log-step scan whose multiplier is squared each step, the carry of the chunk above comes in with one more product, and a
// second Horner walk writes the quotient.  (2 KD_E + 9) / KD_E = 3.1 products per coefficient.
#define KD_THREADS 256
#define KD_E 8
__global__ __launch_bounds__(KD_THREADS) void k_kate_div(const u256* __restrict__ coeff, uint64_t n, u256 x, u256 xe /* x^KD_E */, u256* __restrict__ quot,
                                                         u256* __restrict__ rem) {
  __shared__ u256 sh[KD_THREADS];
  __shared__ u256 s_carry;
  const uint64_t col = blockIdx.x;
  const u256* a = coeff + col * n;
  u256* q = quot + col * n;
  const uint32_t t = threadIdx.x;
  const u256 pw = mont_pow<Fr>(xe, u256_from_u64(KD_THREADS - 1 - t));  // x^(E (255 - t))
  const uint64_t chunk = (uint64_t)KD_THREADS * KD_E;
  if (t == 0) {
    s_carry = u256_zero();
    st256(q + n - 1, u256_zero());
  }
  __syncthreads();
  for (uint64_t c = (n + chunk - 1) / chunk; c-- > 0;) {
    const uint64_t lo = c * chunk + (uint64_t)t * KD_E;
    u256 v[KD_E];
#pragma unroll
    for (int j = 0; j < KD_E; j++) v[j] = lo + j < n ? ld256(a + lo + j) : u256_zero();
    u256 T = v[KD_E - 1];
#pragma unroll
    for (int j = KD_E - 2; j >= 0; j--) T = fr_add(fr_mul(T, x), v[j]);
    // inclusive suffix scan R_t = sum_{t' >= t} T_t' X^(t' - t), X = x^E
    sh[t] = T;
    __syncthreads();
    u256 m = xe;
    for (uint32_t o = 1; o < KD_THREADS; o <<= 1) {
      const bool on = t + o < KD_THREADS;
      u256 other = on ? sh[t + o] : u256_zero();
      __syncthreads();
      if (on) {
        T = fr_add(T, fr_mul(m, other));
        sh[t] = T;
      }
      m = fr_mul(m, m);
      __syncthreads();
    }
    const u256 carry = s_carry;
    u256 s = fr_mul(pw, carry);                                   // S at the first index above this thread's coefficients
    if (t + 1 < KD_THREADS) s = fr_add(s, sh[t + 1]);
    __syncthreads();
    if (t == 0) s_carry = fr_add(T, fr_mul(fr_mul(pw, xe), carry));  // S(c * chunk) = R_0 + X^256 carry
#pragma unroll
    for (int j = KD_E - 1; j >= 0; j--) {
      const uint64_t i = lo + j;
      s = fr_add(fr_mul(s, x), v[j]);
      if (i >= 1 && i < n) st256(q + i - 1, s);
    }
    __syncthreads();
  }
  if (t == 0 && rem) st256(rem + col, s_carry);
}
// out = sum_c v^(n_cols-1-c) p_c  (Horner over the polynomials: out = out * v + p_c), one thread per coefficient
// (nine-limb Horner on 2^261-scaled representatives, as in the quotient kernels: acc * v is one product, the coefficient
// enters as 32 times itself without arithmetic and is added without carries — acc stays below 34 r, limbs below 2 * 2^29)
__global__ __launch_bounds__(256) void k_poly_lincomb(const u256* __restrict__ polys, uint64_t n_cols, uint64_t n, u256 v32 /* 32 v */, u256* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const L9 V = l9_split(v32);
  L9 acc = l9_split32(ld256(out + i));
  for (uint64_t c = 0; c < n_cols; c++) acc = l9_add(l9_mul<Fr>(acc, V), l9_split32(ld256(polys + c * n + i)));
  st256(out + i, l9_canon<Fr>(l9_mul<Fr>(acc, l9_split(mont_one<Fr>()))));
}

// rows [from, n) of every column <- src (n_cols x (n - from)): the blinding rows the prover appends to the columns it derives
__global__ __launch_bounds__(256) void k_fill_rows(u256* __restrict__ cols, uint64_t n, uint64_t from, const u256* __restrict__ src, uint64_t total) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t cnt = n - from, c = i / cnt, r = i % cnt;
  st256(cols + c * n + from + r, ld256(src + i));
}
// acc += a * x, coefficient by coefficient (the scaled sums of SHPLONK's linearisation polynomial)
// out[i] = src[idx[i]]: the cells a circuit makes public, read out of the witness stream in instance order
__global__ __launch_bounds__(256) void k_gather_fr(const u256* __restrict__ src, const int64_t* __restrict__ idx, uint64_t n, u256* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st256(out + i, ld256(src + idx[i]));
}
__global__ __launch_bounds__(256) void k_poly_axpy(u256* __restrict__ acc, u256 a, const u256* __restrict__ x, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st256(acc + i, fr_add(ld256(acc + i), fr_mul(a, ld256(x + i))));
}
__global__ __launch_bounds__(256) void k_poly_scale(u256* __restrict__ x, u256 a, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st256(x + i, fr_mul(a, ld256(x + i)));
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_grand_product_dev(const vdb_fr* num_dev, const vdb_fr* den_dev, size_t n_cols, size_t n, vdb_fr* z_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(num_dev && den_dev && z_dev && n >= 1, "bad argument");
  if (n_cols == 0) return VDB_OK;
  {
    VDB_PROF("k_grand_product");
    hipLaunchKernelGGL(k_grand_product, dim3((unsigned)n_cols), dim3(GP_THREADS), 0, ctx().stream, as_u256(num_dev), as_u256(den_dev), as_u256(z_dev),
                     (uint64_t)n, (uint64_t)n);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

// queues the evaluation kernel: out_dev receives n_cols values
static int eval_polys_launch(const vdb_fr* coeff_dev, size_t n_cols, size_t n, const vdb_fr* x, u256* dout) {
  Context& cx = ctx();
  u256 xv;
  memcpy(&xv, x, 32);
  u256 y = xv;
  for (int i = 0; i < 8; i++) y = fr_mul(y, y);  // x^256
  y = fr_mul(y, host_fr_from_u64(32));             // pre-scaled for the nine-limb product
  {
    VDB_PROF("k_eval_polys");
    hipLaunchKernelGGL(k_eval_polys, dim3((unsigned)n_cols), dim3(EV_THREADS), 0, cx.stream, as_u256(coeff_dev), (uint64_t)n, xv, y, dout);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_eval_polys_dev(const vdb_fr* coeff_dev, size_t n_cols, size_t n, const vdb_fr* x, vdb_fr* out_host) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && x && out_host && n >= 1, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  u256* dout = (u256*)scratch_get(5, n_cols * sizeof(u256));
  if (!dout) return VDB_ERR_OOM;
  int rc = eval_polys_launch(coeff_dev, n_cols, n, x, dout);
  if (rc) return rc;
  VDB_HIP(hipMemcpyAsync(out_host, dout, n_cols * sizeof(u256), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}
int vdb_eval_polys_dev_out(const vdb_fr* coeff_dev, size_t n_cols, size_t n, const vdb_fr* x, vdb_fr* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && x && out_dev && n >= 1, "bad argument");
  if (n_cols == 0) return VDB_OK;
  return eval_polys_launch(coeff_dev, n_cols, n, x, as_u256(out_dev));
}

int vdb_gate_eval_sub_dev(const vdb_fr* adv_ext_dev, uint32_t adv_ext_k, const vdb_fr* sel_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* y,
                          vdb_fr* acc_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(adv_ext_dev && sel_ext_dev && y && acc_dev && adv_ext_k >= ext_k && k + adv_ext_k <= 28, "bad argument");
  if (n_cols == 0) return VDB_OK;
  u256 yv;
  memcpy(&yv, y, 32);
  const uint64_t ne = 1ull << (k + ext_k);
  {
    VDB_PROF("k_gate_eval");
    GateK gk;
    l9_offset_limbs<FrParams>(2, gk.c2);
    hipLaunchKernelGGL(k_gate_eval, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(adv_ext_dev), as_u256(sel_ext_dev), (uint64_t)n_cols,
                     k + ext_k, ext_k, adv_ext_k - ext_k, ne << (adv_ext_k - ext_k), ne, fr_mul(yv, host_fr_from_u64(32)), gk, as_u256(acc_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
// the gates coset by coset: advice cosets [column][adv_slots][row], selector cosets and accumulator [..][n_slots][row]; slots 0 .. n_slots - 1
int vdb_gate_eval_cosets_dev(const vdb_fr* adv_cosets_dev, uint32_t adv_slots, const vdb_fr* sel_cosets_dev, size_t n_cols, uint32_t k, uint32_t n_slots,
                             const vdb_fr* y, vdb_fr* acc_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(adv_cosets_dev && sel_cosets_dev && y && acc_dev && n_slots >= 1 && n_slots <= adv_slots && adv_slots <= 4 && k <= 26, "bad argument");
  if (n_cols == 0) return VDB_OK;
  u256 yv;
  memcpy(&yv, y, 32);
  const uint64_t n = 1ull << k;
  {
    VDB_PROF("k_gate_eval");
    GateK gk;
    l9_offset_limbs<FrParams>(2, gk.c2);
    hipLaunchKernelGGL(k_gate_eval, dim3((unsigned)((n + 255) / 256), n_slots), dim3(256), 0, ctx().stream, as_u256(adv_cosets_dev), as_u256(sel_cosets_dev),
                     (uint64_t)n_cols, k, 0u, 0u, (uint64_t)adv_slots * n, (uint64_t)n_slots * n, fr_mul(yv, host_fr_from_u64(32)), gk, as_u256(acc_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_gate_eval_dev(const vdb_fr* adv_ext_dev, const vdb_fr* sel_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* y, vdb_fr* acc_dev) {
  return vdb_gate_eval_sub_dev(adv_ext_dev, ext_k, sel_ext_dev, n_cols, k, ext_k, y, acc_dev);
}
int vdb_divide_by_vanishing_dev(vdb_fr* h_ext_dev, uint32_t k, uint32_t ext_k) {
  VDB_REQUIRE_INIT();
  VDB_ARG(h_ext_dev && k + ext_k <= 28 && ext_k >= 1 && ext_k <= 8, "bad argument");
  Context& cx = ctx();
  const uint32_t period = 1u << ext_k;
  // X^n on the coset point zeta * w_ext^j is zeta^n * w_{2^e}^j (zeta^3 = 1, n = 2^k: zeta^n = zeta or zeta^2)
  u256 zn = host_zeta();
  for (uint32_t i = 0; i < k; i++) zn = fr_mul(zn, zn);
  const u256 wp = host_root_of_unity(ext_k);
  std::vector<u256> t(period);
  u256 cur = zn;
  for (uint32_t m = 0; m < period; m++) {
    t[m] = mont_inv<Fr>(fr_sub(cur, mont_one<Fr>()));  // never zero: the coset misses the 2^k-th roots of unity
    cur = fr_mul(cur, wp);
  }
  u256* dt = (u256*)scratch_get(5, period * sizeof(u256));
  if (!dt) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(dt, t.data(), period * sizeof(u256), hipMemcpyHostToDevice, cx.stream));
  const uint64_t ne = 1ull << (k + ext_k);
  {
    VDB_PROF("k_mul_periodic");
    hipLaunchKernelGGL(k_mul_periodic, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, cx.stream, as_u256(h_ext_dev), ne, dt, period - 1);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipStreamSynchronize(cx.stream));  // t is a host vector
  return VDB_OK;
}
// z columns of `n` entries from the terms in scratch: z[0 .. usable] is the running product, the rest zero
static int product_columns(const u256* num, const u256* den, size_t n_z, size_t n, size_t usable_rows, u256* z) {
  Context& cx = ctx();
  {
    VDB_PROF("k_grand_product");
    hipLaunchKernelGGL(k_grand_product, dim3((unsigned)n_z), dim3(GP_THREADS), 0, cx.stream, num, den, z, (uint64_t)usable_rows + 1, (uint64_t)n);
  }
  VDB_LAUNCH_CHECK();
  if (usable_rows + 1 < n) {
    hipLaunchKernelGGL(k_zero_tail, dim3((unsigned)((n - usable_rows - 1 + 255) / 256), (unsigned)n_z), dim3(256), 0, cx.stream, z, (uint64_t)n,
                       (uint64_t)usable_rows + 1, (uint64_t)n);
    VDB_LAUNCH_CHECK();
  }
  return VDB_OK;
}

int vdb_fr_delta(vdb_fr* out) {
  VDB_ARG(out, "bad argument");
  u256 e = u256_zero();
  e.w[0] = 1u << 28;  // GENERATOR^(2^S): generates the odd-order part of the multiplicative group
  const u256 d = mont_pow<Fr>(host_fr_from_u64(7), e);
  memcpy(out, &d, 32);
  return VDB_OK;
}

int vdb_permutation_sigma_dev(const uint64_t* mapping_dev, size_t n_cols, uint32_t k, const vdb_fr* delta, vdb_fr* sigma_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(mapping_dev && delta && sigma_dev && k <= 28, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  const uint64_t n = 1ull << k;
  u256 dv;
  memcpy(&dv, delta, 32);
  u256* buf = (u256*)scratch_get(5, (n + n_cols + 1) * sizeof(u256));
  if (!buf) return VDB_ERR_OOM;
  u256 *wpow = buf, *dpow = buf + n;
  int* derr = (int*)(buf + n + n_cols);
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), cx.stream));
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, host_root_of_unity(k), mont_one<Fr>(), n, wpow);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, cx.stream, dv, mont_one<Fr>(), (uint64_t)n_cols, dpow);
  {
    VDB_PROF("k_perm_sigma");
    hipLaunchKernelGGL(k_perm_sigma, dim3((unsigned)((n_cols * n + 255) / 256)), dim3(256), 0, cx.stream, mapping_dev, (uint64_t)n_cols, n, wpow, dpow,
                       as_u256(sigma_dev), derr);
  }
  VDB_LAUNCH_CHECK();
  int herr = 0;
  VDB_HIP(hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  if (herr) {
    set_error("permutation mapping points outside the columns");
    return VDB_ERR_ARG;
  }
  return VDB_OK;
}

int vdb_permutation_mapping_pack_dev(const uint64_t* mapping_dev, size_t n_cols, uint32_t k, uint32_t* packed_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(mapping_dev && packed_dev && k <= 28, "bad argument");
  uint32_t col_bits = 0;
  while (((uint64_t)1 << col_bits) < n_cols) col_bits++;
  VDB_ARG(col_bits + k <= 32, "column and row of a cell do not fit 32 bits together");
  if (n_cols == 0) return VDB_OK;
  const uint64_t cells = (uint64_t)n_cols << k;
  hipLaunchKernelGGL(k_perm_map_pack, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx().stream, mapping_dev, cells, k, packed_dev);
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_permutation_sigma_packed_dev(const uint32_t* packed_block_dev, size_t n_block_cols, size_t n_cols_total, uint32_t k, const vdb_fr* delta,
                                     vdb_fr* sigma_block_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(packed_block_dev && delta && sigma_block_dev && k <= 28 && n_block_cols <= n_cols_total, "bad argument");
  if (n_block_cols == 0) return VDB_OK;
  Context& cx = ctx();
  const uint64_t n = 1ull << k;
  u256 dv;
  memcpy(&dv, delta, 32);
  u256* buf = (u256*)scratch_get(5, (n + n_cols_total + 1) * sizeof(u256));
  if (!buf) return VDB_ERR_OOM;
  u256 *wpow = buf, *dpow = buf + n;
  int* derr = (int*)(buf + n + n_cols_total);
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), cx.stream));
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, host_root_of_unity(k), mont_one<Fr>(), n, wpow);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n_cols_total + 255) / 256)), dim3(256), 0, cx.stream, dv, mont_one<Fr>(), (uint64_t)n_cols_total, dpow);
  const uint64_t cells = (uint64_t)n_block_cols << k;
  {
    VDB_PROF("k_perm_sigma");
    hipLaunchKernelGGL(k_perm_sigma_packed, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, cx.stream, packed_block_dev, cells, (uint64_t)n_cols_total, k, wpow,
                       dpow, as_u256(sigma_block_dev), derr);
  }
  VDB_LAUNCH_CHECK();
  int herr = 0;
  VDB_HIP(hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  if (herr) {
    set_error("packed permutation mapping points outside the columns");
    return VDB_ERR_ARG;
  }
  return VDB_OK;
}

int vdb_permutation_chain_dev(vdb_fr* z_dev, size_t n_chunks, uint32_t k, size_t usable_rows) {
  VDB_REQUIRE_INIT();
  const uint64_t n = 1ull << (k <= 28 ? k : 0);
  VDB_ARG(z_dev && k <= 28 && usable_rows < n, "bad argument");
  if (n_chunks <= 1) return VDB_OK;
  Context& cx = ctx();
  u256* fac = (u256*)scratch_get(5, n_chunks * sizeof(u256));
  if (!fac) return VDB_ERR_OOM;
  {
    VDB_PROF("k_chain");
    hipLaunchKernelGGL(k_chain_factors, dim3(1), dim3(GP_THREADS), 0, cx.stream, as_u256(z_dev), (uint64_t)n_chunks, n, (uint64_t)usable_rows, fac);
    hipLaunchKernelGGL(k_scale_columns, dim3((unsigned)((usable_rows + 1 + 255) / 256), (unsigned)(n_chunks - 1)), dim3(256), 0, cx.stream, as_u256(z_dev), n,
                       (uint64_t)usable_rows + 1, fac);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_permutation_product_range_dev(const vdb_fr* cols_block_dev, const vdb_fr* sigma_block_dev, size_t n_block_cols, size_t col0, uint32_t k, size_t usable_rows,
                                      size_t chunk_len, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, vdb_fr* z_block_dev) {
  VDB_REQUIRE_INIT();
  const uint64_t n = 1ull << (k <= 28 ? k : 0);
  VDB_ARG(cols_block_dev && sigma_block_dev && beta && gamma && delta && z_block_dev && k <= 28 && chunk_len >= 1 && usable_rows < n, "bad argument");
  VDB_ARG(col0 % chunk_len == 0, "a block of columns starts at a chunk boundary");
  if (n_block_cols == 0) return VDB_OK;
  Context& cx = ctx();
  const size_t n_chunks = (n_block_cols + chunk_len - 1) / chunk_len;
  u256 bv, gv, dv;
  memcpy(&bv, beta, 32);
  memcpy(&gv, gamma, 32);
  memcpy(&dv, delta, 32);
  u256* buf = (u256*)scratch_get(5, (n + n_chunks + 2 * n_chunks * n) * sizeof(u256));
  if (!buf) return VDB_ERR_OOM;
  u256 *bw = buf, *dstart = bw + n, *num = dstart + n_chunks, *den = num + n_chunks * n;
  u256 dchunk = mont_one<Fr>();
  for (size_t i = 0; i < chunk_len; i++) dchunk = fr_mul(dchunk, dv);
  // delta^(first column of chunk c of the block) = delta^col0 * (delta^chunk_len)^c
  u256 e = u256_zero();
  e.w[0] = (uint32_t)col0;
  e.w[1] = (uint32_t)((uint64_t)col0 >> 32);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, host_root_of_unity(k), bv, n, bw);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, cx.stream, dchunk, mont_pow<Fr>(dv, e), (uint64_t)n_chunks, dstart);
  if (usable_rows) {
    VDB_PROF("k_perm_terms");
    hipLaunchKernelGGL(k_perm_terms, dim3((unsigned)((usable_rows + 255) / 256), (unsigned)n_chunks), dim3(256), 0, cx.stream, as_u256(cols_block_dev),
                       as_u256(sigma_block_dev), (uint64_t)n_block_cols, n, (uint64_t)usable_rows, (uint32_t)chunk_len, bw, bv, gv, dv, dstart, num, den);
  }
  VDB_LAUNCH_CHECK();
  return product_columns(num, den, n_chunks, n, usable_rows, as_u256(z_block_dev));
}
int vdb_permutation_product_dev(const vdb_fr* cols_dev, const vdb_fr* sigma_dev, size_t n_cols, uint32_t k, size_t usable_rows, size_t chunk_len,
                                const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, vdb_fr* z_dev) {
  VDB_REQUIRE_INIT();
  const uint64_t n = 1ull << (k <= 28 ? k : 0);
  VDB_ARG(cols_dev && sigma_dev && beta && gamma && delta && z_dev && k <= 28 && chunk_len >= 1 && usable_rows < n, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  const size_t n_chunks = (n_cols + chunk_len - 1) / chunk_len;
  u256 bv, gv, dv;
  memcpy(&bv, beta, 32);
  memcpy(&gv, gamma, 32);
  memcpy(&dv, delta, 32);
  // scratch: beta omega^row (n), delta^(chunk start) and chain factors (n_chunks each), the terms (2 x n_chunks x n)
  u256* buf = (u256*)scratch_get(5, (n + 2 * n_chunks + 2 * n_chunks * n) * sizeof(u256));
  if (!buf) return VDB_ERR_OOM;
  u256 *bw = buf, *dstart = bw + n, *fac = dstart + n_chunks, *num = fac + n_chunks, *den = num + n_chunks * n;
  u256 dchunk = mont_one<Fr>();
  for (size_t i = 0; i < chunk_len; i++) dchunk = fr_mul(dchunk, dv);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, host_root_of_unity(k), bv, n, bw);
  hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, cx.stream, dchunk, mont_one<Fr>(), (uint64_t)n_chunks, dstart);
  if (usable_rows) {
    VDB_PROF("k_perm_terms");
    hipLaunchKernelGGL(k_perm_terms, dim3((unsigned)((usable_rows + 255) / 256), (unsigned)n_chunks), dim3(256), 0, cx.stream, as_u256(cols_dev), as_u256(sigma_dev),
                       (uint64_t)n_cols, n, (uint64_t)usable_rows, (uint32_t)chunk_len, bw, bv, gv, dv, dstart, num, den);
  }
  VDB_LAUNCH_CHECK();
  int rc = product_columns(num, den, n_chunks, n, usable_rows, as_u256(z_dev));
  if (rc) return rc;
  if (n_chunks > 1) {
    VDB_PROF("k_chain");
    hipLaunchKernelGGL(k_chain_factors, dim3(1), dim3(GP_THREADS), 0, cx.stream, as_u256(z_dev), (uint64_t)n_chunks, n, (uint64_t)usable_rows, fac);
    hipLaunchKernelGGL(k_scale_columns, dim3((unsigned)((usable_rows + 1 + 255) / 256), (unsigned)(n_chunks - 1)), dim3(256), 0, cx.stream, as_u256(z_dev), n,
                       (uint64_t)usable_rows + 1, fac);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_lookup_product_dev(const vdb_fr* input_dev, const vdb_fr* table_dev, const vdb_fr* perm_input_dev, const vdb_fr* perm_table_dev, size_t n_cols, size_t n,
                           size_t usable_rows, const vdb_fr* beta, const vdb_fr* gamma, vdb_fr* z_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(input_dev && table_dev && perm_input_dev && perm_table_dev && beta && gamma && z_dev && n >= 1 && usable_rows < n, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  u256 bv, gv;
  memcpy(&bv, beta, 32);
  memcpy(&gv, gamma, 32);
  u256* buf = (u256*)scratch_get(5, 2 * n_cols * n * sizeof(u256));
  if (!buf) return VDB_ERR_OOM;
  u256 *num = buf, *den = buf + n_cols * n;
  if (usable_rows) {
    VDB_PROF("k_lookup_terms");
    hipLaunchKernelGGL(k_lookup_terms, dim3((unsigned)((usable_rows + 255) / 256), (unsigned)n_cols), dim3(256), 0, cx.stream, as_u256(input_dev), as_u256(table_dev),
                       as_u256(perm_input_dev), as_u256(perm_table_dev), (uint64_t)n, (uint64_t)usable_rows, bv, gv, num, den);
  }
  VDB_LAUNCH_CHECK();
  return product_columns(num, den, n_cols, n, usable_rows, as_u256(z_dev));
}

// `n_slots` > 0: coset by coset — every array is [column][n_slots][row] and the kernels run once per slot (blockIdx.y) on 2^k points
// with rotations by one; 0: the natural order of the 2^(k + ext_k) points, rotations by 2^ext_k
static int quot_args(QuotArgs& q, const vdb_fr* l0, const vdb_fr* l_last, const vdb_fr* l_active, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta,
                     const vdb_fr* y, uint32_t k, uint32_t ext_k, size_t usable_rows, uint32_t n_slots = 0) {
  q.l0 = as_u256(l0);
  q.l_last = as_u256(l_last);
  q.l_active = as_u256(l_active);
  const u256 m32 = host_fr_from_u64(32);
  u256 t;
  memcpy(&t, beta, 32);
  q.beta = fr_mul(t, m32);
  memcpy(&t, gamma, 32);
  q.gamma = fr_mul(t, m32);
  if (delta) memcpy(&t, delta, 32);
  else t = mont_one<Fr>();
  q.delta = fr_mul(t, m32);
  memcpy(&t, y, 32);
  q.y = fr_mul(t, m32);
  l9_offset_limbs<FrParams>(9, q.c9);
  l9_offset_limbs<FrParams>(34, q.c34);
  q.log_ne = n_slots ? k : k + ext_k;
  q.e = n_slots ? 0 : ext_k;
  q.cs = n_slots ? (uint64_t)n_slots << k : 1ull << (k + ext_k);
  q.last_rot = (1ull << k) - usable_rows;
  return VDB_OK;
}

static int permutation_eval_parts(const vdb_fr* adv_ext_dev, const vdb_fr* sigma_ext_block_dev, const vdb_fr* z_ext_dev, size_t n_cols, size_t chunk_len,
                                  uint32_t k, uint32_t ext_k, size_t usable_rows, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                                  const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev,
                                  size_t set_lo, size_t set_hi, PermParts pp, uint32_t n_slots = 0);
int vdb_permutation_eval_range_dev(const vdb_fr* adv_ext_dev, const vdb_fr* sigma_ext_block_dev, const vdb_fr* z_ext_dev, size_t n_cols, size_t chunk_len,
                                   uint32_t k, uint32_t ext_k, size_t usable_rows, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                                   const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev,
                                   size_t set_lo, size_t set_hi) {
  // everything resident: whole arrays; the terms of the product columns alone come with the first range
  VDB_ARG(z_ext_dev && chunk_len >= 1, "bad argument");
  const size_t n_sets = (n_cols + chunk_len - 1) / chunk_len;
  const uint64_t ne = 1ull << (k + ext_k);
  PermParts pp{0, 0, set_lo == 0 ? 1u : 0u, set_lo == 0 ? n_sets : 0u, as_u256(z_ext_dev), as_u256(z_ext_dev) + (n_sets ? n_sets - 1 : 0) * ne, set_lo == 0 ? 1 : 0};
  return permutation_eval_parts(adv_ext_dev, sigma_ext_block_dev, z_ext_dev, n_cols, chunk_len, k, ext_k, usable_rows, l0_ext_dev, l_last_ext_dev, l_active_ext_dev,
                                beta, gamma, delta, y, acc_dev, set_lo, set_hi, pp);
}
int vdb_permutation_eval_parts_dev(const vdb_fr* adv_ext_block_dev, size_t adv_col0, const vdb_fr* sigma_ext_block_dev, const vdb_fr* z_ext_block_dev, size_t z_set0,
                                   const vdb_fr* z_first_ext_dev, const vdb_fr* z_last_ext_dev, size_t n_cols, size_t chunk_len, uint32_t k, uint32_t ext_k,
                                   size_t usable_rows, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev, const vdb_fr* l_active_ext_dev, const vdb_fr* beta,
                                   const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev, int head, size_t chain_lo, size_t chain_hi,
                                   size_t set_lo, size_t set_hi) {
  VDB_ARG(!(head & 1) || (z_first_ext_dev && z_last_ext_dev), "the head terms read the first and the last product coset");
  VDB_ARG(chain_lo >= chain_hi || (chain_lo >= 1 && chain_lo - 1 >= z_set0), "the chaining term of set i reads set i - 1");
  VDB_ARG(set_lo >= set_hi || (set_lo >= z_set0 && set_lo * chunk_len >= adv_col0), "the block buffers start after the first set asked for");
  PermParts pp{adv_col0, z_set0, chain_lo, chain_hi, as_u256(z_first_ext_dev), as_u256(z_last_ext_dev), head};
  // (a call without product terms still needs non-null column pointers for the argument check: any device pointer does)
  return permutation_eval_parts(adv_ext_block_dev ? adv_ext_block_dev : acc_dev, sigma_ext_block_dev ? sigma_ext_block_dev : acc_dev,
                                z_ext_block_dev ? z_ext_block_dev : acc_dev, n_cols, chunk_len, k, ext_k, usable_rows, l0_ext_dev, l_last_ext_dev,
                                l_active_ext_dev, beta, gamma, delta, y, acc_dev, set_lo, set_hi, pp);
}
// the same coset by coset (vdb_coeff_to_cosets_dev): every array [column][n_slots][row], the accumulator and the Lagrange selectors [n_slots][row]
int vdb_permutation_eval_parts_cosets_dev(const vdb_fr* adv_block_dev, size_t adv_col0, const vdb_fr* sigma_block_dev, const vdb_fr* z_block_dev, size_t z_set0,
                                          const vdb_fr* z_first_dev, const vdb_fr* z_last_dev, size_t n_cols, size_t chunk_len, uint32_t k, uint32_t n_slots,
                                          size_t usable_rows, const vdb_fr* l0_dev, const vdb_fr* l_last_dev, const vdb_fr* l_active_dev, const vdb_fr* beta,
                                          const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev, int head, size_t chain_lo, size_t chain_hi,
                                          size_t set_lo, size_t set_hi) {
  VDB_ARG(n_slots >= 1 && n_slots <= 4, "bad argument");
  VDB_ARG(!(head & 1) || (z_first_dev && z_last_dev), "the head terms read the first and the last product coset");
  VDB_ARG(chain_lo >= chain_hi || (chain_lo >= 1 && chain_lo - 1 >= z_set0), "the chaining term of set i reads set i - 1");
  VDB_ARG(set_lo >= set_hi || (set_lo >= z_set0 && set_lo * chunk_len >= adv_col0), "the block buffers start after the first set asked for");
  PermParts pp{adv_col0, z_set0, chain_lo, chain_hi, as_u256(z_first_dev), as_u256(z_last_dev), head};
  return permutation_eval_parts(adv_block_dev ? adv_block_dev : acc_dev, sigma_block_dev ? sigma_block_dev : acc_dev, z_block_dev ? z_block_dev : acc_dev, n_cols,
                                chunk_len, k, 2, usable_rows, l0_dev, l_last_dev, l_active_dev, beta, gamma, delta, y, acc_dev, set_lo, set_hi, pp, n_slots);
}
static int permutation_eval_parts(const vdb_fr* adv_ext_dev, const vdb_fr* sigma_ext_block_dev, const vdb_fr* z_ext_dev, size_t n_cols, size_t chunk_len,
                                  uint32_t k, uint32_t ext_k, size_t usable_rows, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                                  const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev,
                                  size_t set_lo, size_t set_hi, PermParts pp, uint32_t n_slots) {
  VDB_REQUIRE_INIT();
  VDB_ARG(adv_ext_dev && sigma_ext_block_dev && z_ext_dev && l0_ext_dev && l_last_ext_dev && l_active_ext_dev && beta && gamma && delta && y && acc_dev,
          "null pointer");
  VDB_ARG(k + ext_k <= 28 && chunk_len >= 1 && usable_rows < (1ull << k), "bad argument");
  if (n_cols == 0) return VDB_OK;
  VDB_ARG(set_lo <= set_hi && set_hi <= (n_cols + chunk_len - 1) / chunk_len, "bad set range");
  Context& cx = ctx();
  QuotArgs q;
  VDB_ARG(n_slots <= 4 && (n_slots == 0 || ext_k == 2), "bad argument");
  quot_args(q, l0_ext_dev, l_last_ext_dev, l_active_ext_dev, beta, gamma, delta, y, k, ext_k, usable_rows, n_slots);
  const uint64_t ne = 1ull << q.log_ne;     // points per launch row: the whole domain, or one coset of it
  const uint32_t ny = n_slots ? n_slots : 1;
  u256* bx = (u256*)scratch_get(5, ny * ne * sizeof(u256));
  if (!bx) return VDB_ERR_OOM;
  u256 beta_m, delta_m;
  memcpy(&beta_m, beta, 32);
  memcpy(&delta_m, delta, 32);
  // bx = beta X on the points: beta zeta w^j in the natural order; beta g_t w_n^r on slot t
  for (uint32_t t = 0; t < ny; t++)
    hipLaunchKernelGGL(k_beta_omega_powers, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, cx.stream, host_root_of_unity(q.log_ne),
                       fr_mul(beta_m, n_slots ? host_coset_shift(k, t) : host_zeta()), ne, bx + t * ne);
  {
    VDB_PROF("k_perm_eval");
    u256 e = u256_zero();
    const uint64_t pw = (uint64_t)set_lo * chunk_len;
    e.w[0] = (uint32_t)pw;
    e.w[1] = (uint32_t)(pw >> 32);
    hipLaunchKernelGGL(k_perm_eval, dim3((unsigned)((ne + 255) / 256), ny), dim3(256), 0, cx.stream, as_u256(adv_ext_dev), as_u256(sigma_ext_block_dev),
                       as_u256(z_ext_dev), (uint64_t)n_cols, (uint32_t)chunk_len, (uint64_t)set_lo, (uint64_t)set_hi, bx,
                       fr_mul(mont_pow<Fr>(delta_m, e), host_fr_from_u64(32)), q, pp, as_u256(acc_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_permutation_eval_dev(const vdb_fr* adv_ext_dev, const vdb_fr* sigma_ext_dev, const vdb_fr* z_ext_dev, size_t n_cols, size_t chunk_len, uint32_t k,
                             uint32_t ext_k, size_t usable_rows, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev, const vdb_fr* l_active_ext_dev,
                             const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* delta, const vdb_fr* y, vdb_fr* acc_dev) {
  return vdb_permutation_eval_range_dev(adv_ext_dev, sigma_ext_dev, z_ext_dev, n_cols, chunk_len, k, ext_k, usable_rows, l0_ext_dev, l_last_ext_dev, l_active_ext_dev,
                                        beta, gamma, delta, y, acc_dev, 0, chunk_len ? (n_cols + chunk_len - 1) / chunk_len : 0);
}

static int lookup_eval(const vdb_fr* input_ext_dev, const vdb_fr* table_ext_dev, const vdb_fr* perm_input_ext_dev, const vdb_fr* perm_table_ext_dev,
                       const vdb_fr* z_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                       const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* y, vdb_fr* acc_dev, uint32_t n_slots);
int vdb_lookup_eval_dev(const vdb_fr* input_ext_dev, const vdb_fr* table_ext_dev, const vdb_fr* perm_input_ext_dev, const vdb_fr* perm_table_ext_dev,
                        const vdb_fr* z_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                        const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* y, vdb_fr* acc_dev) {
  return lookup_eval(input_ext_dev, table_ext_dev, perm_input_ext_dev, perm_table_ext_dev, z_ext_dev, n_cols, k, ext_k, l0_ext_dev, l_last_ext_dev, l_active_ext_dev,
                     beta, gamma, y, acc_dev, 0);
}
// coset by coset: every array [column][n_slots][row]
int vdb_lookup_eval_cosets_dev(const vdb_fr* input_dev, const vdb_fr* table_dev, const vdb_fr* perm_input_dev, const vdb_fr* perm_table_dev, const vdb_fr* z_dev,
                               size_t n_cols, uint32_t k, uint32_t n_slots, const vdb_fr* l0_dev, const vdb_fr* l_last_dev, const vdb_fr* l_active_dev,
                               const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* y, vdb_fr* acc_dev) {
  VDB_ARG(n_slots >= 1 && n_slots <= 4, "bad argument");
  return lookup_eval(input_dev, table_dev, perm_input_dev, perm_table_dev, z_dev, n_cols, k, 2, l0_dev, l_last_dev, l_active_dev, beta, gamma, y, acc_dev, n_slots);
}
static int lookup_eval(const vdb_fr* input_ext_dev, const vdb_fr* table_ext_dev, const vdb_fr* perm_input_ext_dev, const vdb_fr* perm_table_ext_dev,
                       const vdb_fr* z_ext_dev, size_t n_cols, uint32_t k, uint32_t ext_k, const vdb_fr* l0_ext_dev, const vdb_fr* l_last_ext_dev,
                       const vdb_fr* l_active_ext_dev, const vdb_fr* beta, const vdb_fr* gamma, const vdb_fr* y, vdb_fr* acc_dev, uint32_t n_slots) {
  VDB_REQUIRE_INIT();
  VDB_ARG(input_ext_dev && table_ext_dev && perm_input_ext_dev && perm_table_ext_dev && z_ext_dev && l0_ext_dev && l_last_ext_dev && l_active_ext_dev && beta &&
              gamma && y && acc_dev,
          "null pointer");
  VDB_ARG(k + ext_k <= 28, "bad argument");
  if (n_cols == 0) return VDB_OK;
  QuotArgs q;
  quot_args(q, l0_ext_dev, l_last_ext_dev, l_active_ext_dev, beta, gamma, nullptr, y, k, ext_k, 0, n_slots);
  u256 beta_m, gamma_m;
  memcpy(&beta_m, beta, 32);
  memcpy(&gamma_m, gamma, 32);
  const uint64_t ne = 1ull << q.log_ne;
  {
    VDB_PROF("k_lookup_eval");
    hipLaunchKernelGGL(k_lookup_eval, dim3((unsigned)((ne + 255) / 256), n_slots ? n_slots : 1), dim3(256), 0, ctx().stream, as_u256(input_ext_dev), as_u256(table_ext_dev),
                       as_u256(perm_input_ext_dev), as_u256(perm_table_ext_dev), as_u256(z_ext_dev), (uint64_t)n_cols, q, beta_m, gamma_m, as_u256(acc_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_fill_rows_dev(vdb_fr* cols_dev, size_t n_cols, size_t n, size_t from_row, const vdb_fr* src_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols_dev && src_dev && from_row <= n, "bad argument");
  const uint64_t total = (uint64_t)n_cols * (n - from_row);
  if (total == 0) return VDB_OK;
  hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(cols_dev), (uint64_t)n, (uint64_t)from_row,
                     as_u256(src_dev), total);
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_gather_fr_dev(const vdb_fr* src_dev, const int64_t* idx_dev, size_t n, vdb_fr* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(n == 0 || (src_dev && idx_dev && out_dev), "bad argument");
  if (n == 0) return VDB_OK;
  {
    VDB_PROF("k_gather_fr");
    hipLaunchKernelGGL(k_gather_fr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(src_dev), idx_dev, (uint64_t)n, as_u256(out_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_poly_axpy_dev(vdb_fr* acc_dev, const vdb_fr* a, const vdb_fr* x_dev, size_t n) {
  VDB_REQUIRE_INIT();
  VDB_ARG(acc_dev && a && x_dev, "null pointer");
  if (n == 0) return VDB_OK;
  u256 av;
  memcpy(&av, a, 32);
  hipLaunchKernelGGL(k_poly_axpy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(acc_dev), av, as_u256(x_dev), (uint64_t)n);
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_poly_scale_dev(vdb_fr* x_dev, const vdb_fr* a, size_t n) {
  VDB_REQUIRE_INIT();
  VDB_ARG(x_dev && a, "null pointer");
  if (n == 0) return VDB_OK;
  u256 av;
  memcpy(&av, a, 32);
  hipLaunchKernelGGL(k_poly_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(x_dev), av, (uint64_t)n);
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_kate_div_dev(const vdb_fr* coeff_dev, size_t n_cols, size_t n, const vdb_fr* x, vdb_fr* quot_dev, vdb_fr* rem_host) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && x && quot_dev && n >= 1, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  u256 xv;
  memcpy(&xv, x, 32);
  u256 xe = xv;
  for (int i = 1; i < KD_E; i <<= 1) xe = fr_mul(xe, xe);  // KD_E is a power of two
  u256* drem = nullptr;
  if (rem_host) {
    drem = (u256*)scratch_get(5, n_cols * sizeof(u256));
    if (!drem) return VDB_ERR_OOM;
  }
  {
    VDB_PROF("k_kate_div");
    hipLaunchKernelGGL(k_kate_div, dim3((unsigned)n_cols), dim3(KD_THREADS), 0, cx.stream, as_u256(coeff_dev), (uint64_t)n, xv, xe, as_u256(quot_dev), drem);
  }
  VDB_LAUNCH_CHECK();
  if (rem_host) {
    VDB_HIP(hipMemcpyAsync(rem_host, drem, n_cols * sizeof(u256), hipMemcpyDeviceToHost, cx.stream));
    VDB_HIP(hipStreamSynchronize(cx.stream));
  }
  return VDB_OK;
}

int vdb_poly_lincomb_dev(const vdb_fr* polys_dev, size_t n_cols, size_t n, const vdb_fr* v, vdb_fr* acc_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(polys_dev && v && acc_dev, "bad argument");
  if (n_cols == 0 || n == 0) return VDB_OK;
  u256 vv;
  memcpy(&vv, v, 32);
  {
    VDB_PROF("k_poly_lincomb");
    hipLaunchKernelGGL(k_poly_lincomb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(polys_dev), (uint64_t)n_cols, (uint64_t)n,
                       fr_mul(vv, host_fr_from_u64(32)), as_u256(acc_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_lookup_permute_dev(const vdb_fr* input_dev, const vdb_fr* table_dev, size_t n_cols, size_t n, size_t usable_rows, uint32_t max_bits,
                           vdb_fr* permuted_input_dev, vdb_fr* permuted_table_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(input_dev && table_dev && permuted_input_dev && permuted_table_dev && usable_rows <= n && max_bits >= 1 && max_bits <= 20, "bad argument");
  if (n_cols == 0 || n == 0) return VDB_OK;
  Context& cx = ctx();
  const uint32_t bins = 1u << max_bits;
  // scratch: table histogram, then per column: input histogram, off, dist, lpre (bins each) and two totals; one error word
  const size_t words = (size_t)bins + n_cols * ((size_t)4 * bins + 2) + 16;
  uint32_t* buf = (uint32_t*)scratch_get(5, words * sizeof(uint32_t));
  if (!buf) return VDB_ERR_OOM;
  int* derr = (int*)buf;
  uint32_t* tab_hist = buf + 16;
  uint32_t* in_hist = tab_hist + bins;
  uint32_t* off = in_hist + n_cols * bins;
  uint32_t* dist = off + n_cols * bins;
  uint32_t* lpre = dist + n_cols * bins;
  uint32_t* totals = lpre + n_cols * bins;
  VDB_HIP(hipMemsetAsync(buf, 0, (16 + (size_t)bins * (1 + n_cols)) * sizeof(uint32_t), cx.stream));
  const unsigned gx = (unsigned)((usable_rows + 255) / 256 > 0 ? ((usable_rows + 255) / 256 < 256 ? (usable_rows + 255) / 256 : 256) : 1);
  {
    VDB_PROF("k_lp_hist");
    hipLaunchKernelGGL(k_lp_hist, dim3(gx, 1), dim3(256), 0, cx.stream, as_u256(table_dev), (uint64_t)0, (uint64_t)usable_rows, max_bits, bins, tab_hist, derr);
    hipLaunchKernelGGL(k_lp_hist, dim3(gx, (unsigned)n_cols), dim3(256), 0, cx.stream, as_u256(input_dev), (uint64_t)n, (uint64_t)usable_rows, max_bits, bins,
                     in_hist, derr);
  }
  VDB_LAUNCH_CHECK();
  {
    VDB_PROF("k_lp_scan");
    hipLaunchKernelGGL(k_lp_scan, dim3((unsigned)n_cols), dim3(LP_THREADS), 0, cx.stream, in_hist, tab_hist, bins, off, dist, lpre, totals, derr);
  }
  VDB_LAUNCH_CHECK();
  int h = 0;
  VDB_HIP(hipMemcpyAsync(&h, derr, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  if (h) {
    set_error(h & 1 ? "lookup_permute: a value is not below 2^max_bits" : "lookup_permute: an input value does not occur in the table (the reference panics)");
    return VDB_ERR_DOMAIN;
  }
  const unsigned gr = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  {
    VDB_PROF("k_lp_rows");
    hipLaunchKernelGGL(k_lp_rows, dim3(gr, (unsigned)n_cols), dim3(256), 0, cx.stream, in_hist, tab_hist, off, dist, lpre, totals, bins, (uint64_t)usable_rows,
                     (uint64_t)n, as_u256(permuted_input_dev), as_u256(permuted_table_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

}  // extern "C"
