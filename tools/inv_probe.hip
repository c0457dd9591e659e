// Probe for the batched-affine question: what one field inversion costs a wavefront in units of field products,
// measured — a Fermat inversion (a^(q-2) over Fq, as the bucket accumulator would need) against the nine-limb product, both as
// dependent chains in every lane with enough wavefronts in flight to fill the SIMDs.  The inversion's cost does not shrink when the
// lanes of a wavefront share one inverse: the chain occupies all 64 lanes either way.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o inv_probe inv_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../halo2_vectordb_amd/csrc/field.hpp"
#include "../halo2_vectordb_amd/csrc/limb9.hpp"
using namespace vdb;

__global__ __launch_bounds__(256) void k_inv(u256* __restrict__ io, int iters) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u256 x = io[t];
  for (int i = 0; i < iters; i++) x = mont_inv<Fq>(x);
  io[t] = x;
}
__global__ __launch_bounds__(256) void k_mul(u256* __restrict__ io, int iters) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  L9 x = l9_split(io[t]);
  const L9 w = l9_split(io[t ^ 1]);
  for (int i = 0; i < iters; i++) x = l9_mul<Fq>(x, w);
  io[t] = l9_canon<Fq>(x);
}
int main() {
  const size_t n = (size_t)256 * 256 * 16;   // 16 wavefronts per SIMD's worth of threads
  u256* d;
  if (hipMalloc(&d, n * sizeof(u256)) != hipSuccess) return 1;
  u256* h = new u256[n];
  for (size_t i = 0; i < n; i++)
    for (int w = 0; w < 8; w++) h[i].w[w] = (uint32_t)(0x9E3779B9u * (uint32_t)(i * 8 + w + 1)) >> (w == 7 ? 4 : 0);
  (void)hipMemcpy(d, h, n * sizeof(u256), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms_inv = 1e9f, ms_mul = 1e9f;
  const int it_inv = 4, it_mul = 4096;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_inv, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, it_inv);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < ms_inv) ms_inv = ms;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mul, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, it_mul);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < ms_mul) ms_mul = ms;
  }
  const double inv_per_s = (double)n * it_inv / (ms_inv * 1e-3), mul_per_s = (double)n * it_mul / (ms_mul * 1e-3);
  printf("{\"inversions_per_s\": %.4g, \"nine_limb_products_per_s\": %.4g, \"products_per_inversion\": %.1f}\n", inv_per_s, mul_per_s, mul_per_s / inv_per_s);
  return 0;
}
