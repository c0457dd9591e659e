// Probe: how fast can 64 lanes that each own a private output run (the witness kernels' pattern: one lane = one gadget instance,
// consecutive 32-byte cells at the lane's own stream offset) write to HBM — (A) as the kernels do today, two 16-byte stores per
// cell and lane, 64 cache lines per store instruction; (B) after a transpose inside each quad of lanes, so that the four lanes of
// a quad write the 64 contiguous bytes (two cells) of ONE of them per store instruction; (C) the same over 8 lanes and four
// cells (one whole 128-byte line per 8 lanes); (D) plain coalesced streaming as the ceiling.  Values are synthetic.
// Build: hipcc -O3 --offload-arch=gfx950 -o store_probe store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int CELLS = 4096;            // cells per lane (32 B each): 128 KiB per lane
__global__ __launch_bounds__(64) void k_a(uint4* out, size_t lane_stride /* in uint4 */) {
  const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
  uint4* p = out + lane * lane_stride;
  uint4 v = make_uint4(lane, 1, 2, 3);
  for (int c = 0; c < CELLS; c++) {
    v.x += 7; v.y ^= v.x;
    p[2 * c] = v;
    p[2 * c + 1] = v;
  }
}
// quad transpose: per pair of cells each lane holds four 16-byte pieces; store i moves lane i's pieces to the quad's four lanes
__global__ __launch_bounds__(64) void k_b(uint4* out, size_t lane_stride) {
  const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int q = threadIdx.x & 3;
  uint4* base[4];
  for (int i = 0; i < 4; i++) base[i] = out + (lane - q + i) * lane_stride + q;   // lane i's run, my piece slot
  uint4 v = make_uint4(lane, 1, 2, 3);
  for (int c = 0; c < CELLS; c += 2) {
    uint4 pc[4];
    for (int k = 0; k < 4; k++) { v.x += 7; v.y ^= v.x; pc[k] = v; }
    for (int i = 0; i < 4; i++) {
      uint4 r;
      // piece q of lane i: fetch all four pieces from lane i of the quad and keep mine
      uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
      for (int w = 0; w < 4; w++) {
        uint32_t got[4];
        for (int k = 0; k < 4; k++) got[k] = __shfl(reinterpret_cast<uint32_t*>(&pc[k])[w], (threadIdx.x & ~3) + i, 64);
        dst[w] = q == 0 ? got[0] : q == 1 ? got[1] : q == 2 ? got[2] : got[3];
      }
      base[i][2 * c] = r;
    }
  }
}
__global__ __launch_bounds__(64) void k_d(uint4* out, size_t total) {
  size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 64;
  uint4 v = make_uint4(i, 1, 2, 3);
  for (; i < total; i += stride) out[i] = v;
}
int main() {
  const size_t lanes = 64 * 4096, lane_stride = 2 * CELLS + 16 * 1024 / 16;   // runs 16 KiB apart beyond their own length
  const size_t total = lanes * lane_stride;
  uint4* d;
  if (hipMalloc(&d, total * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double bytes = (double)lanes * CELLS * 32;
  for (int kind = 0; kind < 3; kind++) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k_a, dim3(lanes / 64), dim3(64), 0, 0, d, lane_stride);
      if (kind == 1) hipLaunchKernelGGL(k_b, dim3(lanes / 64), dim3(64), 0, 0, d, lane_stride);
      if (kind == 2) hipLaunchKernelGGL(k_d, dim3(256 * 32), dim3(64), 0, 0, d, (size_t)(bytes / 16));
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("{\"kind\": \"%s\", \"ms\": %.3f, \"TB_per_s\": %.2f}\n", kind == 0 ? "lane_private_16B_stores" : kind == 1 ? "quad_transposed_64B_runs" : "coalesced_stream", best, bytes / best / 1e9);
  }
  return 0;
}
