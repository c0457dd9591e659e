// Raw issue-rate probe for the integer instructions the field arithmetic is made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 valu_probe.hip -o valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHAINS 8
template <int KIND>
__global__ __launch_bounds__(256) void probe(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[CHAINS];
  for (int i = 0; i < CHAINS; i++) acc[i] = threadIdx.x + i;
  uint32_t x = a + threadIdx.x, y = b;
  double d[CHAINS];
  for (int i = 0; i < CHAINS; i++) d[i] = (double)(threadIdx.x + i);
  double dx = (double)a, dy = (double)b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
      if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
      if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"((uint64_t)x));
      if (KIND == 2) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); acc[i] = lo; }
      if (KIND == 3) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); acc[i] = lo; }
      if (KIND == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
      if (KIND == 5) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(x)); acc[i] = lo; }
      if (KIND == 6) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); acc[i] = lo; }
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < CHAINS; i++) s += acc[i] + (uint64_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name) {
  const int blocks = 256 * 16, iters = 4096;
  uint64_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 12345u, 678u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * CHAINS;
  printf("%-18s %8.2f T lane-ops/s   (%.3f ms)\n", name, ops / (ms * 1e-3) / 1e12, ms);
  hipFree(out);
}
int main() {
  run<2>("v_add_u32");
  run<1>("v_lshl_add_u64");
  run<0>("v_mad_u64_u32");
  run<3>("v_mul_lo_u32");
  run<6>("v_mul_hi_u32");
  run<5>("v_mad_u32_u24");
  run<4>("v_fma_f64");
  return 0;
}
