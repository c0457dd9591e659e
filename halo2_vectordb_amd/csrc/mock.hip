// Device-side MockProver: what the reference's Mock stage does with halo2's MockProver::run(..).assert_satisfied()
// (/root/reference/src/scaffold/mod.rs:263-266) — every gate row, every lookup cell and every copy constraint of the
// circuit checked row by row, no commitment, no transform — as one pass over the witness where it already lies in HBM.
// Works on the flat streams (halo2-base Context.advice / cells_to_lookup): the column layout only duplicates the cell at
// each break point (a copy that holds by construction), so stream rows and column rows carry the same constraints.
//   gate      every cell flagged as a gate start (bit 0 of the keygen run's flag byte): a[i] + a[i+1] a[i+2] == a[i+3]
//   lookup    every lookup cell is a canonical value below 2^lookup_bits (the range table 0 .. 2^L - 1)
//   copies    copy_of[i] != i: a[i] == a[copy_of[i]]  (Existing cells and the layout's own ties);
//             lookup_src[j]: lookup[j] == a[lookup_src[j]]  (cells_to_lookup are copies of advice cells)
//   constants every cell flagged constant (bit 1) equals the same cell of the keygen-time stream (what the fixed column holds);
//             or, with the circuit's constraint map: const_idx[i] = r >= 0 -> a[i] == const_table[r] (Constant cells and
//             assert_is_const ties alike: both are copies of the fixed column's cell r)
// Pure HBM streaming: 32 B per cell read once (+ 1 flag byte, + 8 B per copy index); the products are one per gate row.
#include "common.hpp"

namespace vdb {

struct MockCounters {
  unsigned long long n[5];      // gate, lookup range, copy, lookup copy, constant
  unsigned long long first[5];  // smallest offending index of each kind (~0 when none)
};

__device__ __forceinline__ void mock_report(MockCounters* m, int kind, uint64_t idx) {
  atomicAdd(&m->n[kind], 1ull);
  atomicMin(&m->first[kind], (unsigned long long)idx);
}

__global__ __launch_bounds__(256) void k_mock_cells(const u256* __restrict__ a, uint64_t n_cells, const uint8_t* __restrict__ flags,
                                                   const int64_t* __restrict__ copy_of, const u256* __restrict__ consts, const int64_t* __restrict__ const_idx,
                                                   const u256* __restrict__ const_table, uint64_t n_consts, MockCounters* m) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
    const uint8_t f = flags ? flags[i] : 0;
    u256 v = ld256(a + i);
    if (f & 1) {
      if (i + 3 >= n_cells) {
        mock_report(m, 0, i);
      } else {
        const u256 lhs = fr_add(v, fr_mul(ld256(a + i + 1), ld256(a + i + 2)));
        if (!u256_eq(lhs, ld256(a + i + 3))) mock_report(m, 0, i);
      }
    }
    if (copy_of) {
      const int64_t src = copy_of[i];
      if (src >= 0 && (uint64_t)src != i && ((uint64_t)src >= n_cells || !u256_eq(v, ld256(a + src)))) mock_report(m, 2, i);
    }
    if ((f & 2) && consts && !u256_eq(v, ld256(consts + i))) mock_report(m, 4, i);
    if (const_idx) {
      const int64_t r = const_idx[i];
      if (r >= 0 && ((uint64_t)r >= n_consts || !u256_eq(v, ld256(const_table + r)))) mock_report(m, 4, i);
    }
  }
}
__global__ __launch_bounds__(256) void k_mock_lookups(const u256* __restrict__ lk, uint64_t n_lookup, uint32_t lookup_bits, const u256* __restrict__ a,
                                                     uint64_t n_cells, const int64_t* __restrict__ lookup_src, MockCounters* m) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lookup; j += stride) {
    const u256 v = ld256(lk + j);
    if (u256_bits(from_mont<Fr>(v)) > lookup_bits) mock_report(m, 1, j);
    if (lookup_src) {
      const int64_t src = lookup_src[j];
      if (src < 0 || (uint64_t)src >= n_cells || !u256_eq(v, ld256(a + src))) mock_report(m, 3, j);
    }
  }
}

__global__ __launch_bounds__(256) void k_mock_instances(const u256* __restrict__ a, uint64_t n_cells, const int64_t* __restrict__ cells,
                                                       const u256* __restrict__ values, uint64_t n, unsigned long long* __restrict__ out /* count, first */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cells[i];
  if (c < 0 || (uint64_t)c >= n_cells || !u256_eq(ld256(a + c), ld256(values + i))) {
    atomicAdd(&out[0], 1ull);
    atomicMin(&out[1], (unsigned long long)i);
  }
}

}  // namespace vdb

using namespace vdb;

extern "C" int vdb_mock_check_dev(const vdb_fr* stream_dev, uint64_t n_cells, const uint8_t* flags_dev, const vdb_fr* lookup_dev, uint64_t n_lookup,
                                  uint32_t lookup_bits, const int64_t* copy_of_dev, const int64_t* lookup_src_dev, const vdb_fr* const_stream_dev,
                                  const int64_t* const_idx_dev, const vdb_fr* const_table_dev, uint64_t n_consts, vdb_mock_report* out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(out && (stream_dev || n_cells == 0) && (lookup_dev || n_lookup == 0) && lookup_bits >= 1 && lookup_bits <= 32, "bad argument");
  VDB_ARG((const_idx_dev == nullptr) == (const_table_dev == nullptr), "constant indices and the constants' table go together");
  Context& c = ctx();
  MockCounters* d = (MockCounters*)scratch_get(5, sizeof(MockCounters));
  if (!d) return VDB_ERR_OOM;
  MockCounters h;
  for (int i = 0; i < 5; i++) h.n[i] = 0, h.first[i] = ~0ull;
  VDB_HIP(hipMemcpyAsync(d, &h, sizeof(h), hipMemcpyHostToDevice, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));  // `h` is on the stack
  const unsigned grid = (unsigned)(c.cu_count * 16);
  if (n_cells) {
    VDB_PROF("k_mock_cells");
    hipLaunchKernelGGL(k_mock_cells, dim3(grid), dim3(256), 0, c.stream, as_u256(stream_dev), n_cells, flags_dev, copy_of_dev, as_u256(const_stream_dev),
                       const_idx_dev, as_u256(const_table_dev), n_consts, d);
  }
  VDB_LAUNCH_CHECK();
  if (n_lookup) {
    VDB_PROF("k_mock_lookups");
    hipLaunchKernelGGL(k_mock_lookups, dim3(grid), dim3(256), 0, c.stream, as_u256(lookup_dev), n_lookup, lookup_bits, as_u256(stream_dev), n_cells, lookup_src_dev, d);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  out->gate_rows_violated = h.n[0], out->first_gate_row = h.first[0];
  out->lookup_cells_out_of_table = h.n[1], out->first_lookup_cell = h.first[1];
  out->copies_unequal = h.n[2], out->first_copy = h.first[2];
  out->lookup_copies_unequal = h.n[3], out->first_lookup_copy = h.first[3];
  out->constants_changed = h.n[4], out->first_constant = h.first[4];
  out->instances_unequal = 0, out->first_instance = ~0ull;
  return VDB_OK;
}

extern "C" int vdb_mock_check_instances_dev(const vdb_fr* stream_dev, uint64_t n_cells, const int64_t* instance_cells_dev, const vdb_fr* instances_dev,
                                            uint64_t n_instances, vdb_mock_report* out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(out && (n_instances == 0 || (stream_dev && instance_cells_dev && instances_dev)), "bad argument");
  out->instances_unequal = 0, out->first_instance = ~0ull;
  if (n_instances == 0) return VDB_OK;
  Context& c = ctx();
  unsigned long long* d = (unsigned long long*)scratch_get(5, 2 * sizeof(unsigned long long));
  if (!d) return VDB_ERR_OOM;
  unsigned long long h[2] = {0ull, ~0ull};
  VDB_HIP(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));  // `h` is on the stack
  {
    VDB_PROF("k_mock_instances");
    hipLaunchKernelGGL(k_mock_instances, dim3((unsigned)((n_instances + 255) / 256)), dim3(256), 0, c.stream, as_u256(stream_dev), n_cells, instance_cells_dev,
                       as_u256(instances_dev), n_instances, d);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  out->instances_unequal = h[0], out->first_instance = h[1];
  return VDB_OK;
}
