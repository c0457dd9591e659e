// Keygen on the device: the permutation of a circuit's copy constraints.
//
// Replaces what halo2's keygen_vk / keygen_pk do with the equalities halo2-base recorded (plonk/permutation/keygen.rs
// Assembly::copy: union of cycles on the host; reached from /root/reference/src/scaffold/mod.rs:273) for circuits of 10^8 - 10^9
// cells, where a host-side sort of every (column, row) takes minutes and a hundred GB of RAM:
//   1. pointer jumping over the copy forest (every cell points at an earlier cell, at itself, or at a cell of the constants'
//      fixed column) until every cell holds the root of its class;
//   2. one (root, position) record per grid position a class occupies — a stream cell's (column, row), the duplicate of the
//      overlap cell at the end of each column, a lookup cell's position in the lookup columns, the fixed cells, and the rows of
//      the instance column the circuit's public cells are tied to (halo2-base's RangeWithInstanceCircuitBuilder:
//      layouter.constrain_instance(cell, instance_column, i), /root/reference/src/scaffold/mod.rs:400);
//   3. a radix sort of the records by root (rocPRIM device radix sort: the sort is HBM-bound, 4 passes over 16 B records);
//   4. every record points at the next one of its class, the last at the first: the cycles.  Positions in no class keep the
//      identity.
// Output: the mapping words col << 32 | row that vdb_permutation_sigma_dev turns into the sigma columns.
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace vdb {

__global__ __launch_bounds__(256) void k_pm_jump(int64_t* __restrict__ root, uint64_t n_cells, int* __restrict__ changed) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  bool any = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
    const int64_t r = root[i];
    if (r >= 0 && (uint64_t)r < n_cells) {
      const int64_t rr = root[r];
      if (rr != r) {
        root[i] = rr;
        any = true;
      }
    }
  }
  if (any) *changed = 1;
}
__global__ __launch_bounds__(256) void k_pm_identity(uint64_t* __restrict__ mapping, uint64_t n_cols, uint64_t rows) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_cols * rows) mapping[i] = ((i / rows) << 32) | (i % rows);
}
struct PmShape {
  uint64_t n_cells, n_dup, n_lookup, n_consts, n_inst;  // records: stream cells, overlap duplicates, lookup cells, fixed cells, instance rows
  uint64_t n_adv, lookup_rows, fixed_col;               // the instance column is column fixed_col + 1
};
// record t -> (root, col << 32 | row); starts[c] = stream offset of row 0 of advice column c (n_adv + 1 entries, the last = n_cells)
__global__ __launch_bounds__(256) void k_pm_records(PmShape s, const int64_t* __restrict__ root, const uint64_t* __restrict__ starts,
                                                   const uint64_t* __restrict__ bp, const int64_t* __restrict__ lookup_src,
                                                   const int64_t* __restrict__ inst_cells, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
  const uint64_t total = s.n_cells + s.n_dup + s.n_lookup + s.n_consts + s.n_inst;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    uint64_t key, col, row;
    if (t < s.n_cells) {
      uint64_t lo = 0, hi = s.n_adv;  // largest c with starts[c] <= t
      while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (starts[mid] <= t) lo = mid;
        else hi = mid;
      }
      col = lo;
      row = t - starts[lo];
      key = (uint64_t)root[t];
    } else if (t < s.n_cells + s.n_dup) {
      const uint64_t d = t - s.n_cells;  // the cell that starts column d + 1 also ends column d, at row bp[d]
      col = d;
      row = bp[d];
      key = (uint64_t)root[starts[d + 1]];
    } else if (t < s.n_cells + s.n_dup + s.n_lookup) {
      const uint64_t j = t - s.n_cells - s.n_dup;
      col = s.n_adv + j / s.lookup_rows;
      row = j % s.lookup_rows;
      key = (uint64_t)root[lookup_src[j]];
    } else if (t < s.n_cells + s.n_dup + s.n_lookup + s.n_consts) {
      const uint64_t r = t - s.n_cells - s.n_dup - s.n_lookup;
      col = s.fixed_col;
      row = r;
      key = s.n_cells + r;
    } else {  // row i of the instance column joins the class of the cell that was made public i-th
      const uint64_t r = t - s.n_cells - s.n_dup - s.n_lookup - s.n_consts;
      col = s.fixed_col + 1;
      row = r;
      key = (uint64_t)root[inst_cells[r]];
    }
    keys[t] = key;
    vals[t] = (col << 32) | row;
  }
}
__global__ __launch_bounds__(256) void k_pm_link(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, uint64_t total, uint64_t rows,
                                                uint64_t* __restrict__ mapping) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t key = keys[t], v = vals[t];
    uint64_t nxt;
    if (t + 1 < total && keys[t + 1] == key) {
      nxt = vals[t + 1];
    } else {  // the last of its class closes the cycle at the first
      uint64_t lo = 0, hi = t;  // first index with keys[idx] == key: keys[lo .. ] sorted
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
      }
      nxt = vals[lo];
    }
    mapping[(v >> 32) * rows + (v & 0xffffffffull)] = nxt;
  }
}

}  // namespace vdb

using namespace vdb;

extern "C" int vdb_permutation_mapping_dev(int64_t* parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t* break_points, uint64_t n_bp, uint32_t k,
                                           const int64_t* lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                           const int64_t* instance_cells_dev, uint64_t n_instances, uint64_t* mapping_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(parent_dev && mapping_dev && (break_points || n_bp == 0) && k <= 28 && n_cells >= 1, "bad argument");
  const uint64_t rows = 1ull << k, n_adv = n_bp + 1;
  VDB_ARG(n_cols >= n_adv && n_consts <= rows && (n_lookup == 0 || (lookup_src_dev && lookup_rows >= 1 && lookup_rows <= rows)), "bad shape");
  VDB_ARG(n_lookup == 0 || n_adv + (n_lookup + lookup_rows - 1) / lookup_rows <= n_cols, "lookup cells do not fit the lookup columns");
  VDB_ARG(n_instances <= rows && (n_instances == 0 || instance_cells_dev), "more public cells than rows of the instance column");
  Context& cx = ctx();
  // starts (n_adv + 1) and break points on the device
  std::vector<uint64_t> h(2 * n_adv + 1);
  uint64_t acc = 0;
  for (uint64_t c = 0; c < n_bp; c++) {
    VDB_ARG(break_points[c] < rows, "break point beyond the column");
    h[c] = acc;
    acc += break_points[c];
    h[n_adv + 1 + c] = break_points[c];
  }
  h[n_bp] = acc;
  h[n_adv] = n_cells;
  VDB_ARG(acc < n_cells && n_cells - acc <= rows, "break points do not describe a stream of this length");
  uint64_t* d_starts = nullptr;
  VDB_HIP(hipMalloc(&d_starts, h.size() * sizeof(uint64_t)));
  struct Guard {
    std::vector<void*> p;
    ~Guard() {
      for (void* q : p) (void)hipFree(q);
    }
  } guard;
  guard.p.push_back(d_starts);
  VDB_HIP(hipMemcpyAsync(d_starts, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice, cx.stream));
  const uint64_t* d_bp = d_starts + n_adv + 1;
  int* d_flag = nullptr;
  VDB_HIP(hipMalloc(&d_flag, sizeof(int)));
  guard.p.push_back(d_flag);
  const unsigned grid = (unsigned)(cx.cu_count * 16);
  // 1. roots
  for (int it = 0; it < 64; it++) {
    VDB_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), cx.stream));
    {
      VDB_PROF("k_pm_jump");
      hipLaunchKernelGGL(k_pm_jump, dim3(grid), dim3(256), 0, cx.stream, parent_dev, n_cells, d_flag);
    }
    VDB_LAUNCH_CHECK();
    int changed = 0;
    VDB_HIP(hipMemcpyAsync(&changed, d_flag, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
    VDB_HIP(hipStreamSynchronize(cx.stream));
    if (!changed) break;
    if (it == 63) {
      set_error("copy map does not settle: it holds a cycle");
      return VDB_ERR_ARG;
    }
  }
  // 2. records
  PmShape s{n_cells, n_bp, n_lookup, n_consts, n_instances, n_adv, lookup_rows ? lookup_rows : 1, n_cols};
  const uint64_t total = n_cells + n_bp + n_lookup + n_consts + n_instances;
  uint64_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  for (uint64_t** q : {&keys, &vals, &keys2, &vals2}) {
    hipError_t e = hipMalloc(q, total * sizeof(uint64_t));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(permutation records: 4 x 8 B per grid position of a copy class)", __FILE__, __LINE__);
    guard.p.push_back(*q);
  }
  {
    VDB_PROF("k_pm_records");
    hipLaunchKernelGGL(k_pm_records, dim3(grid), dim3(256), 0, cx.stream, s, parent_dev, d_starts, d_bp, lookup_src_dev, instance_cells_dev, keys, vals);
  }
  VDB_LAUNCH_CHECK();
  // 3. sort by root
  unsigned end_bit = 1;
  while (end_bit < 64 && ((n_cells + n_consts) >> end_bit)) end_bit++;
  // double-buffered: the two copies of the records are the sort's own ping-pong storage, the temporary storage is histograms only
  rocprim::double_buffer<uint64_t> dk(keys, keys2), dv(vals, vals2);
  size_t tmp_bytes = 0;
  VDB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)total, 0u, end_bit, cx.stream));
  void* tmp = nullptr;
  VDB_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8));
  guard.p.push_back(tmp);
  {
    VDB_PROF("rocprim_radix_sort_pairs");
    VDB_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)total, 0u, end_bit, cx.stream));
  }
  keys2 = dk.current();
  vals2 = dv.current();
  // 4. cycles
  const uint64_t n_grid = (n_cols + 2) * rows;  // [advice | lookup | constants | instance]
  hipLaunchKernelGGL(k_pm_identity, dim3((unsigned)((n_grid + 255) / 256)), dim3(256), 0, cx.stream, mapping_dev, n_cols + 2, rows);
  {
    VDB_PROF("k_pm_link");
    hipLaunchKernelGGL(k_pm_link, dim3(grid), dim3(256), 0, cx.stream, keys2, vals2, total, rows, mapping_dev);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}
