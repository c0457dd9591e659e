// How the bare nine-limb Montgomery product (mont_core29) scales with occupancy: the same kernel launched with a dynamic LDS
// request that allows 1, 2, 3, 4, 6, 8 workgroups of 256 threads per CU (= waves per SIMD), with 1, 2 or 4 independent product
// chains per thread.  The NTT runs at 3 workgroups per CU (LDS bound): this says what a fourth would be worth.
// Build: hipcc -O3 --offload-arch=gfx950 -I../halo2_vectordb_amd/csrc occ_probe.hip -o occ_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "field.hpp"
using namespace vdb;

template <int CH>
__global__ __launch_bounds__(256) void probe(uint32_t* out, const uint32_t* w, int iters) {
  extern __shared__ uint32_t lds[];
  uint32_t a[CH][9], W[9];
  for (int j = 0; j < 9; j++) {
    W[j] = (w[j] + threadIdx.x) & 0x1fffffffu;
    for (int c = 0; c < CH; c++) a[c][j] = (w[j] * (3 + 2 * c) + blockIdx.x) & 0x1fffffffu;
  }
  if (iters < 0) lds[threadIdx.x] = W[0];  // keep the allocation alive
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      uint32_t r[9];
      mont_core29<Fr>(r, a[c], W);
#pragma unroll
      for (int j = 0; j < 9; j++) a[c][j] = r[j];
    }
  }
  uint32_t s = 0;
  for (int c = 0; c < CH; c++)
    for (int j = 0; j < 9; j++) s += a[c][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static void run(const uint32_t* dw, int wg_per_cu) {
  const int blocks = 256 * 24, iters = 4096 / CH;
  size_t lds = wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) - 512;
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute((const void*)probe<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(256), lds, 0, out, dw, 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<CH>, dim3(blocks), dim3(256), lds, 0, out, dw, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * CH;
  printf("chains %d  workgroups/CU %d (waves/SIMD %d): %7.2f G mul/s\n", CH, wg_per_cu, wg_per_cu, ops / (ms * 1e-3) / 1e9);
  hipFree(out);
}
int main() {
  uint32_t hw[18];
  for (int i = 0; i < 18; i++) hw[i] = 0x12345678u * (i + 1) + 0x9abcdefu;
  uint32_t* dw;
  hipMalloc(&dw, sizeof(hw));
  hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
  for (int wg : {1, 2, 3, 4, 6, 8}) run<1>(dw, wg);
  for (int wg : {1, 2, 3, 4, 6, 8}) run<2>(dw, wg);
  for (int wg : {1, 2, 3, 4, 6, 8}) run<4>(dw, wg);
  return 0;
}
