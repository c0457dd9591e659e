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
                                                   const int64_t* __restrict__ inst_cells, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                                   int* __restrict__ err) {
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
      const int64_t src = lookup_src[j];
      if (src < 0 || (uint64_t)src >= s.n_cells) {  // a lookup cell that copies nothing inside the stream: refused by the caller
        *err = 1;
        key = s.n_cells + s.n_consts;
      } else {
        key = (uint64_t)root[src];
      }
    } else if (t < s.n_cells + s.n_dup + s.n_lookup + s.n_consts) {
      const uint64_t r = t - s.n_cells - s.n_dup - s.n_lookup;
      col = s.fixed_col;
      row = r;
      key = s.n_cells + r;
    } else {  // row i of the instance column joins the class of the cell that was made public i-th
      const uint64_t r = t - s.n_cells - s.n_dup - s.n_lookup - s.n_consts;
      col = s.fixed_col + 1;
      row = r;
      const int64_t cell = inst_cells[r];
      if (cell < 0 || (uint64_t)cell >= s.n_cells) {  // a public cell outside the stream
        *err = 1;
        key = s.n_cells + s.n_consts;
      } else {
        key = (uint64_t)root[cell];
      }
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

// ---- the circuit's constraint map, instantiated on the device -------------------------------------------------------------
// The gadgets' copy structure is data independent and repeats: one traced block (a distance, a per-vector assignment, a filter, a
// division: circuit_sym.py) is placed at hundreds of thousands of stream offsets.  Block cell codes: src >= 0 copies block cell src,
// src <= MAP_EXT0 is external input number MAP_EXT0 - src of the instance, anything else ties the cell to nothing (itself).
constexpr int64_t MAP_EXT0 = -10;
__global__ __launch_bounds__(256) void k_map_iota(int64_t* __restrict__ a, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = (int64_t)i;
}
__global__ __launch_bounds__(256) void k_map_place(const int64_t* __restrict__ blk_src, const int64_t* __restrict__ blk_cid, const uint8_t* __restrict__ blk_flags,
                                                  uint64_t n_blk, const int64_t* __restrict__ bases, const int64_t* __restrict__ ext, uint64_t m, uint64_t n_ext,
                                                  uint64_t n_cells, int64_t* __restrict__ copy_of, int64_t* __restrict__ const_idx, uint8_t* __restrict__ flags,
                                                  int* __restrict__ err) {
  const uint64_t total = m * n_blk, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t inst = t / n_blk, c = t % n_blk;
    const int64_t b = bases[inst], src = blk_src[c];
    const uint64_t idx = (uint64_t)b + c;
    if (b < 0 || idx >= n_cells) {
      *err = 1;
      continue;
    }
    int64_t val = (int64_t)idx;
    if (src >= 0) {
      if ((uint64_t)src >= n_blk) {  // a block cell that copies a cell outside its block
        *err = 1;
        continue;
      }
      val = b + src;
    } else if (src <= MAP_EXT0) {
      const uint64_t e = (uint64_t)(MAP_EXT0 - src);
      if (e >= n_ext) {
        *err = 1;
        continue;
      }
      val = ext[inst * n_ext + e];
    }
    copy_of[idx] = val;
    const_idx[idx] = blk_cid[c];
    flags[idx] = blk_flags[c];
  }
}
__global__ __launch_bounds__(256) void k_map_place_lookups(const int64_t* __restrict__ blk_lk, uint64_t n_blk_lk, const int64_t* __restrict__ bases,
                                                          const int64_t* __restrict__ lk_bases, const int64_t* __restrict__ ext, uint64_t m, uint64_t n_ext,
                                                          uint64_t n_lookup, int64_t* __restrict__ lookup_src, int* __restrict__ err) {
  const uint64_t total = m * n_blk_lk, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t inst = t / n_blk_lk, c = t % n_blk_lk;
    const int64_t src = blk_lk[c], lb = lk_bases[inst];
    if (lb < 0 || (uint64_t)lb + c >= n_lookup) {
      *err = 1;
      continue;
    }
    int64_t val;
    if (src <= MAP_EXT0) {
      const uint64_t e = (uint64_t)(MAP_EXT0 - src);
      if (e >= n_ext) {
        *err = 1;
        continue;
      }
      val = ext[inst * n_ext + e];
    } else {
      val = bases[inst] + (src > 0 ? src : 0);
    }
    lookup_src[(uint64_t)lb + c] = val;
  }
}
// parent[i] for vdb_permutation_mapping_dev: the fixed cell of its constant for a tied cell (which must copy nothing), else the cell it
// copies; counts[0] = tied cells that copy another cell, counts[1] = lookup cells without a source
__global__ __launch_bounds__(256) void k_map_finish(const int64_t* __restrict__ copy_of, const int64_t* __restrict__ const_idx, uint64_t n_cells,
                                                   const int64_t* __restrict__ lookup_src, uint64_t n_lookup, int64_t* __restrict__ parent,
                                                   unsigned long long* __restrict__ counts) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0, nosrc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
    const int64_t c = const_idx[i], p = copy_of[i];
    if (c >= 0 && p != (int64_t)i) bad++;
    if (parent) parent[i] = c >= 0 ? (int64_t)n_cells + c : p;
  }
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_lookup; j += stride)
    if (lookup_src[j] < 0) nosrc++;
  if (bad) atomicAdd(&counts[0], bad);
  if (nosrc) atomicAdd(&counts[1], nosrc);
}

}  // namespace vdb

using namespace vdb;

extern "C" int vdb_copymap_init_dev(uint64_t n_cells, uint64_t n_lookup, int64_t* copy_of_dev, int64_t* const_idx_dev, uint8_t* flags_dev, int64_t* lookup_src_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(copy_of_dev && const_idx_dev && flags_dev && (lookup_src_dev || n_lookup == 0), "null pointer");
  Context& cx = ctx();
  hipLaunchKernelGGL(k_map_iota, dim3((unsigned)(cx.cu_count * 16)), dim3(256), 0, cx.stream, copy_of_dev, n_cells);
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemsetAsync(const_idx_dev, 0xff, n_cells * sizeof(int64_t), cx.stream));   // -1: tied to no constant
  VDB_HIP(hipMemsetAsync(flags_dev, 0, n_cells, cx.stream));
  if (n_lookup) VDB_HIP(hipMemsetAsync(lookup_src_dev, 0xff, n_lookup * sizeof(int64_t), cx.stream));
  return VDB_OK;
}

extern "C" int vdb_copymap_place_dev(const int64_t* blk_src_dev, const int64_t* blk_cid_dev, const uint8_t* blk_flags_dev, uint64_t n_blk, const int64_t* blk_lk_dev,
                                     uint64_t n_blk_lk, const int64_t* bases_dev, const int64_t* lk_bases_dev, const int64_t* ext_dev, uint64_t m, uint64_t n_ext,
                                     uint64_t n_cells, uint64_t n_lookup, int64_t* copy_of_dev, int64_t* const_idx_dev, uint8_t* flags_dev,
                                     int64_t* lookup_src_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(blk_src_dev && blk_cid_dev && blk_flags_dev && bases_dev && copy_of_dev && const_idx_dev && flags_dev, "null pointer");
  VDB_ARG((n_ext == 0 || ext_dev) && (n_blk_lk == 0 || (blk_lk_dev && lk_bases_dev && lookup_src_dev)), "null pointer");
  if (m == 0 || n_blk == 0) return VDB_OK;
  Context& cx = ctx();
  int* derr = (int*)scratch_get(5, 64);
  if (!derr) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), cx.stream));
  const unsigned grid = (unsigned)(cx.cu_count * 16);
  {
    VDB_PROF("k_map_place");
    hipLaunchKernelGGL(k_map_place, dim3(grid), dim3(256), 0, cx.stream, blk_src_dev, blk_cid_dev, blk_flags_dev, n_blk, bases_dev, ext_dev, m, n_ext, n_cells,
                       copy_of_dev, const_idx_dev, flags_dev, derr);
    if (n_blk_lk)
      hipLaunchKernelGGL(k_map_place_lookups, dim3(grid), dim3(256), 0, cx.stream, blk_lk_dev, n_blk_lk, bases_dev, lk_bases_dev, ext_dev, m, n_ext, n_lookup,
                         lookup_src_dev, derr);
  }
  VDB_LAUNCH_CHECK();
  int herr = 0;
  VDB_HIP(hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));   // also: the caller's small host arrays may be reused after the call
  if (herr) {
    set_error("copy map: a block instance lies outside the stream, names an external input it was not given, or copies a cell outside its block");
    return VDB_ERR_ARG;
  }
  return VDB_OK;
}

extern "C" int vdb_copymap_finish_dev(const int64_t* copy_of_dev, const int64_t* const_idx_dev, uint64_t n_cells, const int64_t* lookup_src_dev, uint64_t n_lookup,
                                      int64_t* parent_dev, uint64_t* tied_not_root, uint64_t* lookups_without_source) {
  VDB_REQUIRE_INIT();
  VDB_ARG(copy_of_dev && const_idx_dev && tied_not_root && lookups_without_source && (lookup_src_dev || n_lookup == 0), "null pointer");
  Context& cx = ctx();
  unsigned long long* d = (unsigned long long*)scratch_get(5, 64);
  if (!d) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), cx.stream));
  {
    VDB_PROF("k_map_finish");
    hipLaunchKernelGGL(k_map_finish, dim3((unsigned)(cx.cu_count * 16)), dim3(256), 0, cx.stream, copy_of_dev, const_idx_dev, n_cells, lookup_src_dev, n_lookup,
                       parent_dev, d);
  }
  VDB_LAUNCH_CHECK();
  unsigned long long h[2];
  VDB_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  *tied_not_root = h[0];
  *lookups_without_source = h[1];
  return VDB_OK;
}

extern "C" int vdb_permutation_mapping_ws_dev(int64_t* parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t* break_points, uint64_t n_bp, uint32_t k,
                                              const int64_t* lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                              const int64_t* instance_cells_dev, uint64_t n_instances, uint64_t* mapping_dev, void* work_dev,
                                              size_t work_bytes);
extern "C" int vdb_permutation_mapping_dev(int64_t* parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t* break_points, uint64_t n_bp, uint32_t k,
                                           const int64_t* lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                           const int64_t* instance_cells_dev, uint64_t n_instances, uint64_t* mapping_dev) {
  return vdb_permutation_mapping_ws_dev(parent_dev, n_cells, n_consts, break_points, n_bp, k, lookup_src_dev, n_lookup, lookup_rows, n_cols, instance_cells_dev,
                                        n_instances, mapping_dev, nullptr, 0);
}
// `work_dev` / `work_bytes`: device memory the caller lends for the sort records (4 x 8 B per grid position of a copy class + the sort's
// histograms: 51 GiB for the k = 16 cosine k-means) — a keygen hands over the buffer its sigma columns and selectors will fill
// afterwards, so that the records' memory is neither mapped nor handed back; too small or null: the call allocates its own
extern "C" int vdb_permutation_mapping_ws_dev(int64_t* parent_dev, uint64_t n_cells, uint64_t n_consts, const uint64_t* break_points, uint64_t n_bp, uint32_t k,
                                              const int64_t* lookup_src_dev, uint64_t n_lookup, uint64_t lookup_rows, uint64_t n_cols,
                                              const int64_t* instance_cells_dev, uint64_t n_instances, uint64_t* mapping_dev, void* work_dev,
                                              size_t work_bytes) {
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
  // the sort's histograms are sized first, so that a lent workspace is known to hold everything
  size_t tmp_bytes = 0;
  {
    rocprim::double_buffer<uint64_t> dk0(nullptr, nullptr), dv0(nullptr, nullptr);
    VDB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk0, dv0, (size_t)total, 0u, 64u, cx.stream));
  }
  const size_t rec_bytes = (total * sizeof(uint64_t) + 255) / 256 * 256, tmp_room = (tmp_bytes + 255) / 256 * 256 + 256;
  void* tmp = nullptr;
  if (work_dev && work_bytes >= 4 * rec_bytes + tmp_room && ((uintptr_t)work_dev & 255) == 0) {
    uint8_t* w = (uint8_t*)work_dev;
    keys = (uint64_t*)w;
    vals = (uint64_t*)(w + rec_bytes);
    keys2 = (uint64_t*)(w + 2 * rec_bytes);
    vals2 = (uint64_t*)(w + 3 * rec_bytes);
    tmp = w + 4 * rec_bytes;
  } else {
    for (uint64_t** q : {&keys, &vals, &keys2, &vals2}) {
      hipError_t e = timed_malloc((void**)q, total * sizeof(uint64_t));
      if (e != hipSuccess) return hip_fail(e, "hipMalloc(permutation records: 4 x 8 B per grid position of a copy class)", __FILE__, __LINE__);
      guard.p.push_back(*q);
    }
    VDB_HIP(timed_malloc(&tmp, tmp_room));
    guard.p.push_back(tmp);
  }
  {
    VDB_PROF("k_pm_records");
    VDB_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), cx.stream));
    hipLaunchKernelGGL(k_pm_records, dim3(grid), dim3(256), 0, cx.stream, s, parent_dev, d_starts, d_bp, lookup_src_dev, instance_cells_dev, keys, vals, d_flag);
  }
  VDB_LAUNCH_CHECK();
  {
    int bad = 0;
    VDB_HIP(hipMemcpyAsync(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, cx.stream));
    VDB_HIP(hipStreamSynchronize(cx.stream));
    if (bad) {
      set_error("permutation mapping: a public cell or a lookup cell's source lies outside the stream");
      return VDB_ERR_ARG;
    }
  }
  // 3. sort by root
  unsigned end_bit = 1;
  while (end_bit < 64 && ((n_cells + n_consts) >> end_bit)) end_bit++;
  // double-buffered: the two copies of the records are the sort's own ping-pong storage, the temporary storage is histograms only
  rocprim::double_buffer<uint64_t> dk(keys, keys2), dv(vals, vals2);
  {
    size_t need = 0;
    VDB_HIP(rocprim::radix_sort_pairs(nullptr, need, dk, dv, (size_t)total, 0u, end_bit, cx.stream));
    if (need > tmp_room) {
      set_error("permutation mapping: the sort asks for more temporary storage than was sized");
      return VDB_ERR_HIP;
    }
    tmp_bytes = need;
  }
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
