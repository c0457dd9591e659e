// Probe: the nine-limb Montgomery core as the compiler schedules its C++ form (two chains joined by a 64-bit add per column, moves)
// against the library's device form, ONE inline-asm statement with a single accumulator chain (field.hpp, core29_mul.inc:
// 162 v_mad_u64_u32 + 9 v_mul_lo_u32 + 18 v_and + 17 shifts = 206 instructions).  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../halo2_vectordb_amd/csrc/field.hpp"
using namespace vdb;

// the C++ form of the core (what field.hpp keeps for the host), compiled for the device: the compiler's own schedule
__device__ __forceinline__ void core_cxx(uint32_t out[9], const uint32_t A[9], const uint32_t B[9]) {
  constexpr uint32_t MASK = 0x1fffffffu;
  uint32_t mq[9];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (uint64_t)A[i] * B[k - i];
#pragma unroll
    for (int j = 1; j <= k; j++) acc += (uint64_t)mq[k - j] * FrParams::P29[j];
    mq[k] = ((uint32_t)acc * FrParams::INV29) & MASK;
    acc += (uint64_t)mq[k] * FrParams::P29[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 18; k++) {
#pragma unroll
    for (int i = k - 8; i <= 8; i++) acc += (uint64_t)A[i] * B[k - i];
#pragma unroll
    for (int j = k - 8; j <= 8; j++) acc += (uint64_t)mq[k - j] * FrParams::P29[j];
    out[k - 9] = k < 17 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
    acc >>= 29;
  }
}
template <int KIND>
__global__ __launch_bounds__(256) void k_probe(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t X[9], B[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { X[i] = in[t * 18 + i]; B[i] = in[t * 18 + 9 + i]; }
  for (int it = 0; it < iters; it++) {
    uint32_t Y[9];
    if (KIND == 0) core_cxx(Y, X, B); else mont_core29<FrParams>(Y, X, B);
#pragma unroll
    for (int i = 0; i < 9; i++) X[i] = Y[i];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) out[t * 9 + i] = X[i];
}
int main() {
  const size_t n = 256 * 256 * 16; const int iters = 512;
  std::vector<uint32_t> h(n * 18);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)s & 0x1fffffffu; }
  uint32_t *din, *d0, *d1;
  hipMalloc(&din, h.size() * 4); hipMalloc(&d0, n * 36); hipMalloc(&d1, n * 36);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; kind++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k_probe<0>, dim3(n / 256), dim3(256), 0, 0, din, d0, iters);
      else hipLaunchKernelGGL(k_probe<1>, dim3(n / 256), dim3(256), 0, 0, din, d1, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("{\"kind\": \"%s\", \"ms\": %.3f, \"G_products_per_s\": %.1f}\n", kind ? "asm_single_chain" : "compiler", ms, (double)n * iters / ms / 1e6);
    }
  }
  std::vector<uint32_t> r0(n * 9), r1(n * 9);
  hipMemcpy(r0.data(), d0, n * 36, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 36, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < n * 9; i++) bad += r0[i] != r1[i];
  printf("{\"mismatching_limbs\": %zu}\n", bad);
  return bad != 0;
}
