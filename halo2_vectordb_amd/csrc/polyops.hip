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

__global__ __launch_bounds__(GP_THREADS) void k_grand_product(const u256* __restrict__ num, const u256* __restrict__ den, u256* __restrict__ z, uint64_t n) {
  __shared__ u256 sh[GP_THREADS];
  __shared__ u256 s_dinv;
  const uint64_t col = blockIdx.x;
  const u256* nu = num + col * n;
  const u256* de = den + col * n;
  u256* zo = z + col * n;
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
                     (uint64_t)n);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_eval_polys_dev(const vdb_fr* coeff_dev, size_t n_cols, size_t n, const vdb_fr* x, vdb_fr* out_host) {
  VDB_REQUIRE_INIT();
  VDB_ARG(coeff_dev && x && out_host && n >= 1, "bad argument");
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  u256 xv;
  memcpy(&xv, x, 32);
  u256 y = xv;
  for (int i = 0; i < 8; i++) y = fr_mul(y, y);  // x^256
  y = fr_mul(y, host_fr_from_u64(32));             // pre-scaled for the nine-limb product
  u256* dout = (u256*)scratch_get(5, n_cols * sizeof(u256));
  if (!dout) return VDB_ERR_OOM;
  {
    VDB_PROF("k_eval_polys");
    hipLaunchKernelGGL(k_eval_polys, dim3((unsigned)n_cols), dim3(EV_THREADS), 0, cx.stream, as_u256(coeff_dev), (uint64_t)n, xv, y, dout);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(out_host, dout, n_cols * sizeof(u256), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}

}  // extern "C"
