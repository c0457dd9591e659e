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

}  // extern "C"
