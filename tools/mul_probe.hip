// Throughput probe: Montgomery product (mont_core29) against a Shoup-style product by a constant with a precomputed
// quotient (a * w - floor(a * w' / 2^261) * p, low 261 bits), both on nine 29-bit limbs.  Timing only.
// Build: hipcc -O3 --offload-arch=gfx950 -I../halo2_vectordb_amd/csrc mul_probe.hip -o mul_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "field.hpp"
using namespace vdb;

template <class M>
__device__ __forceinline__ void shoup_core29(uint32_t out[9], const uint32_t A[9], const uint32_t W[9], const uint32_t WQ[9]) {
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t q[9], PN[9];
  // 2^261 - p, limb by limb (p's limbs are below 2^29; the constant folding does the borrow chain)
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      uint32_t v = (0u - M::P29[j] - borrow);
      PN[j] = v & MASK;
      borrow = (M::P29[j] + borrow) ? 1 : 0;
    }
  }
  uint64_t acc = 0;
#pragma unroll
  for (int k = 7; k < 18; k++) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) mad64(acc, A[i], WQ[k - i]);
    if (k >= 9) q[k - 9] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
  q[8] = (uint32_t)acc;
  acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) {
      mad64(acc, A[i], W[k - i]);
      mad64(acc, q[i], PN[k - i]);
    }
    out[k] = (uint32_t)acc & MASK;
    acc >>= 29;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint32_t* out, const uint32_t* w, int iters) {
  uint32_t a[2][9], W[9], WQ[9];
  for (int j = 0; j < 9; j++) {
    W[j] = (w[j] + threadIdx.x) & 0x1fffffffu;
    WQ[j] = (w[9 + j] + threadIdx.x) & 0x1fffffffu;
    a[0][j] = (w[j] * 3 + blockIdx.x) & 0x1fffffffu;
    a[1][j] = (w[j] * 5 + blockIdx.x) & 0x1fffffffu;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      uint32_t r[9];
      if (KIND == 0) mont_core29<Fr>(r, a[c], W);
      else shoup_core29<Fr>(r, a[c], W, WQ);
#pragma unroll
      for (int j = 0; j < 9; j++) a[c][j] = r[j];
    }
  }
  uint32_t s = 0;
  for (int j = 0; j < 9; j++) s += a[0][j] ^ a[1][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, const uint32_t* dw) {
  const int blocks = 256 * 16, iters = 2048;
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, dw, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, dw, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * 2;
  printf("%-18s %8.2f G mul/s   (%.3f ms)\n", name, ops / (ms * 1e-3) / 1e9, ms);
  hipFree(out);
}
int main() {
  uint32_t hw[18];
  for (int i = 0; i < 18; i++) hw[i] = 0x12345678u * (i + 1) + 0x9abcdefu;
  uint32_t* dw;
  hipMalloc(&dw, sizeof(hw));
  hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
  run<0>("mont_core29", dw);
  run<1>("shoup_core29", dw);
  run<0>("mont_core29", dw);
  run<1>("shoup_core29", dw);
  return 0;
}
