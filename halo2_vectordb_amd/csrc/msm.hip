// Batched multi-scalar multiplication over BN254 G1 for gfx950.
//
// Replaces halo2 `arithmetic::best_multiexp` / `ParamsKZG::commit_lagrange` (third-party halo2-axiom,
// reached from /root/reference/src/scaffold/mod.rs:296 and :273): one exact
// group element per column, returned as canonical affine (identity = (0,0)).
//
// MI355X-first design (not the CPU's per-thread chunked Pippenger):
//  * The bases are fixed per circuit size, so vdb_srs_load precomputes T[j][i] = 2^(c*j) * G_i for every
//    c-bit window j (tens of MB in 288 GB of HBM, L2 / Infinity-Cache resident).  All windows of a column
//    then share ONE bucket set: a signed digit d of window j of scalar i adds +-T[j][i] to bucket |d|.
//    No per-window doubling chain, one bucket reduction per column.
//  * Scalars are sign-folded (s > r/2 -> r - s with the point negated) so the many "negative" witness
//    values become short; zero digits cost nothing.
//  * k_msm_sort: one workgroup per column does a counting sort of the (digit -> table index) pairs
//    entirely in LDS (histogram, scan, scatter) and cuts the sorted list into ranges of LCAP entries; bucket
//    changes inside a range open new segments (one partial sum each), which removes the skew of witness columns.
//    With wide windows (thousands of buckets, dense scalars) the scatter goes in two phases: entries to the 256 runs of coarse bins
//    here, k_msm_scatter2 orders each bin in place through LDS — four-byte stores to 8,192 open runs per column do not merge in L2.
//  * k_msm_accum: one thread per range (every lane does exactly LCAP mixed additions), XYZZ accumulator in VGPRs.
//  * k_msm_combine: segmented tree reduction of the partials of every bucket.
//  * k_msm_reduce: one wavefront per column; each lane folds its slice of buckets with the running-sum
//    trick, lanes are combined with wavefront shuffles, lane 0 normalises to affine.
// Roofline: algorithmic HBM traffic is 32 B per scalar (+ bases once); the kernels are bound by 32-bit
// integer multiply-add throughput (v_mad_u64_u32), see DESIGN.md.
#include "common.hpp"
#include "ec.hpp"
#include "limb9.hpp"

struct vdb_srs {
  uint32_t k;
  size_t n;
  uint32_t c, W, B;
  vdb::Affine* table[2];  // [0] monomial, [1] lagrange; each W * n points
  int device;             // the GPU whose HBM holds the tables
};
// a handle only works on the device it was loaded on (the calling thread's current one: vdb_set_device)
#define VDB_SRS_HERE(srs) VDB_ARG((srs)->device == vdb::ctx().device, "this srs handle was loaded on another device (vdb_set_device)")

namespace vdb {

#define MSM_SORT_THREADS 1024
// Range length (entries one thread of k_msm_accum adds up) is chosen per batch: long ranges mean fewer segments, i.e.
// less work for k_msm_partials / k_msm_combine (256: 4 ms of combine on the k = 16 kmeans workload against 20 ms at
// 32), short ranges keep small jobs parallel (a single 2^16 column has only ~130 k entries).
#define MSM_LCAP_MIN 32
#define MSM_LCAP_MAX 256

// A range is `lcap` consecutive entries of one column's bucket-sorted entry list.  Inside a range every change of
// bucket starts a new *segment*; each segment produces one partial sum.  Segment numbering follows the sorted order,
// so the partials of one bucket are contiguous: [seg_off[b], seg_off[b + 1]).
struct MsmRange {
  uint32_t col, bucket, start, len, seg0, pad;
};
struct MsmSegInfo {
  uint32_t idx, len;  // position of the segment among its bucket's segments, and how many the bucket has
};

// ---------------------------------------------------------------- table precompute
__global__ __launch_bounds__(256) void k_srs_table(const Affine* __restrict__ bases, Affine* __restrict__ table, size_t n, uint32_t c, uint32_t W) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine p = ld_affine(bases + i);
  // table entries are stored times 32, i.e. in Montgomery form with R' = 2^261: the form k_msm_accum computes in
  // (the identity (0, 0) is unaffected)
  const u256 m32 = to_mont<Fq>(u256_from_u64(32));
  for (uint32_t j = 0; j < W; j++) {
    Affine ps;
    ps.x = fq_mul(p.x, m32);
    ps.y = fq_mul(p.y, m32);
    st_affine(table + (size_t)j * n + i, ps);
    if (j + 1 < W) {
      XYZZ q = xyzz_double_affine(p);
      for (uint32_t d = 1; d < c; d++) q = xyzz_double(q);
      p = xyzz_to_affine(q);
    }
  }
}

// ---------------------------------------------------------------- digit decomposition
// canonical scalar -> (folded magnitude, sign)
__device__ __forceinline__ bool fold_scalar(u256& s) {
  // (r-1)/2
  const uint32_t H[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
  u256 half;
#pragma unroll
  for (int i = 0; i < 8; i++) half.w[i] = H[i];
  // s > half  <=>  !(half >= s)
  if (!u256_geq(half, s)) {
    u256 r = mod_p<Fr>(), t;
    u256_sub(t, r, s);
    s = t;
    return true;
  }
  return false;
}

// signed window digits of a folded magnitude s (nb = its bit length): f(window, |digit|, negative)
template <class F>
__device__ __forceinline__ void emit_digits(u256 s, bool neg, uint32_t nb, uint32_t c, uint32_t W, F&& f, size_t idx = 0) {
  uint32_t carry = 0;
  const uint32_t half = 1u << (c - 1), full = 1u << c;
  // witness scalars are mostly short: stop after the window that can still receive a carry
  uint32_t wend = nb / c + 2;
  if (wend > W) wend = W;
  for (uint32_t j = 0; j < wend; j++) {
    uint32_t d = (s.w[0] & (full - 1)) + carry;
    s = u256_shr_small(s, c);
    bool dneg = false;
    if (d > half) {
      d = full - d;
      dneg = true;
      carry = 1;
    } else {
      carry = 0;
    }
    if (d) f(j, d, neg != dneg, idx);
  }
}
template <class F>
__device__ __forceinline__ void for_each_digit(const u256& mont_scalar, uint32_t c, uint32_t W, F&& f) {
  if (u256_is_zero(mont_scalar)) return;  // Montgomery form of zero is zero: skip the product for the ~1/3 zero cells
  u256 s = from_mont<Fr>(mont_scalar);
  bool neg = fold_scalar(s);
  emit_digits(s, neg, u256_bits(s), c, W, [&](uint32_t j, uint32_t d, bool ng, size_t) { f(j, d, ng); });
}
// same for a magnitude below 2^32
template <class F>
__device__ __forceinline__ void emit_digits32(uint32_t mag, bool neg, uint32_t nb, uint32_t c, uint32_t W, F&& f, size_t idx) {
  uint32_t carry = 0;
  const uint32_t half = 1u << (c - 1), full = 1u << c;
  uint32_t wend = nb / c + 2;
  if (wend > W) wend = W;
  for (uint32_t j = 0; j < wend; j++) {
    uint32_t d = (mag & (full - 1)) + carry;
    mag >>= c;
    bool dneg = false;
    if (d > half) {
      d = full - d;
      dneg = true;
      carry = 1;
    } else {
      carry = 0;
    }
    if (d) f(j, d, neg != dneg, idx);
  }
}
// Two-speed, two-pass walk over a column's scalars (the counting sort visits them twice: histogram, then scatter).
// Every lane of a wavefront pays for the longest path among its 64 scalars, and witness columns mix a majority of short
// values (bits, 15-bit limbs) with ~100-bit fixed-point values.  First visit (`first`): the Montgomery reduction, sign
// fold and bit length are computed once; a short scalar (magnitude below 2^32 and at most three windows) leaves an
// 8-byte record {magnitude, sign, length} and is processed from registers; a long one is appended to a queue
// (wavefront-aggregated) and all long ones are then processed densely, every lane holding one.  Second visit: the short
// scalars come straight from their records (no 32-byte load, no field arithmetic), the long ones from the queue.
// `queue`, `rec`: n entries of this column each.
#define MSM_REC_VALID (1ull << 63)
// `sc` (a materialised column) or `cs` (a column still lying in the witness stream, see ColSrc) supplies the scalars.
template <class F>
__device__ __forceinline__ void walk_scalars(const u256* __restrict__ sc, const ColSrc* __restrict__ cs, uint32_t n_blind, const uint8_t* __restrict__ mk,
                                             size_t n, uint32_t c, uint32_t W, uint32_t* __restrict__ queue, unsigned long long* __restrict__ rec,
                                             uint32_t* s_qcnt, bool first, F&& f, bool dense = false) {
  ColSrc src;
  if (cs) src = *cs;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  if (dense) {
    // columns of full-width scalars (products, quotient pieces, fixed columns: the wide-window tables): no short / long sorting, every
    // scalar is reduced and decomposed where it is met — one Montgomery reduction per visit instead of two on the first
    for (size_t i0 = 0; i0 < n; i0 += MSM_SORT_THREADS) {
      const size_t i = i0 + tid;
      if (i >= n || (mk && mk[i])) continue;
      u256 s = cs ? colsrc_fetch(src, i, n, n_blind) : ld256(sc + i);
      if (u256_is_zero(s)) continue;
      s = from_mont<Fr>(s);
      const bool neg = fold_scalar(s);
      emit_digits(s, neg, u256_bits(s), c, W, f, i);
    }
    return;
  }
  const uint32_t short_bits = 3 * c - 1 < 32 ? 3 * c - 1 : 32;
  if (first) {
    if (tid == 0) *s_qcnt = 0;
    __syncthreads();
    for (size_t i0 = 0; i0 < n; i0 += MSM_SORT_THREADS) {
      const size_t i = i0 + tid;
      bool live = i < n && !(mk && mk[i]);  // masked: constant cell, its term is part of the precomputed per-column point
      u256 s;
      bool neg = false;
      uint32_t nb = 0;
      if (live) {
        s = cs ? colsrc_fetch(src, i, n, n_blind) : ld256(sc + i);
        live = !u256_is_zero(s);  // Montgomery form of zero is zero: no reduction for the ~1/3 zero cells
      }
      if (live) {
        s = from_mont<Fr>(s);
        neg = fold_scalar(s);
        nb = u256_bits(s);
      }
      const bool is_long = live && nb > short_bits;
      const unsigned long long m = __ballot(is_long);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(s_qcnt, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (is_long) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
      }
      const bool is_short = live && !is_long;
      if (i < n) rec[i] = is_short ? (MSM_REC_VALID | ((unsigned long long)nb << 40) | ((unsigned long long)(neg ? 1 : 0) << 32) | s.w[0]) : 0ull;
      if (is_short) emit_digits32(s.w[0], neg, nb, c, W, f, i);
    }
  } else {
    for (size_t i0 = 0; i0 < n; i0 += MSM_SORT_THREADS) {
      const size_t i = i0 + tid;
      const unsigned long long r = i < n ? rec[i] : 0ull;
      if (r & MSM_REC_VALID) emit_digits32((uint32_t)r, ((r >> 32) & 1) != 0, (uint32_t)(r >> 40) & 0xffu, c, W, f, i);
    }
  }
  __syncthreads();  // queue complete (written and read by this workgroup only)
  const uint32_t nq = *s_qcnt;
  for (uint32_t qi = tid; qi < nq; qi += MSM_SORT_THREADS) {
    const size_t i = queue[qi];
    u256 s = from_mont<Fr>(cs ? colsrc_fetch(src, i, n, n_blind) : ld256(sc + i));
    const bool neg = fold_scalar(s);
    emit_digits(s, neg, u256_bits(s), c, W, f, i);
  }
}

// block-wide exclusive scan of one value per thread (blockDim.x = MSM_SORT_THREADS); returns the
// exclusive prefix, *total gets the block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, 64);
    if ((int)lane >= o) inc += t;
  }
  if (lane == 63) wave_sums[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    uint32_t ws = lane < (MSM_SORT_THREADS / 64) ? wave_sums[lane] : 0;
    uint32_t winc = ws;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(winc, o, 64);
      if ((int)lane >= o) winc += t;
    }
    if (lane < (MSM_SORT_THREADS / 64)) wave_sums[lane] = winc - ws;  // exclusive wave offsets
    if (lane == (MSM_SORT_THREADS / 64) - 1) wave_sums[MSM_SORT_THREADS / 64] = winc;
  }
  __syncthreads();
  uint32_t res = wave_sums[wave] + inc - v;
  *total = wave_sums[MSM_SORT_THREADS / 64];
  __syncthreads();
  return res;
}

// per column: the number of (scalar, window) entries its MSM sorts and accumulates — the non-zero signed digits of the
// cells that are not masked out
__global__ __launch_bounds__(256) void k_msm_count_entries(const u256* __restrict__ scalars, const uint8_t* __restrict__ mask, uint64_t rows, uint32_t c,
                                                           uint32_t W, unsigned long long* __restrict__ counts) {
  const uint64_t col = blockIdx.x;
  uint32_t n = 0;
  for (uint64_t r = threadIdx.x; r < rows; r += 256) {
    const uint64_t i = col * rows + r;
    if (mask && mask[i]) continue;
    for_each_digit(ld256(scalars + i), c, W, [&](uint32_t, uint32_t, bool) { n++; });
  }
  for (int o = 32; o >= 1; o >>= 1) n += __shfl_down(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&counts[col], (unsigned long long)n);
}

// One workgroup per column: counting sort of (bucket -> table index|sign), segment numbering and range cutting.
__global__ __launch_bounds__(MSM_SORT_THREADS) void k_msm_sort(const u256* __restrict__ scalars, const ColSrc* __restrict__ srcs, uint32_t n_blind, size_t n,
                                                               size_t table_n, uint32_t c, uint32_t W,
                                                               uint32_t* __restrict__ entries, size_t ent_cap,
                                                               uint32_t* __restrict__ seg_off, uint32_t* __restrict__ bucket_off, MsmRange* __restrict__ ranges,
                                                               uint32_t* __restrict__ counters /* [0]=segments, [1]=overflow, [2]=max segs/bucket, [3]=ranges */,
                                                               uint32_t seg_cap, uint32_t range_cap,
                                                               const uint8_t* __restrict__ skip_mask /* optional: n per column */,
                                                               uint32_t lcap /* range length, a power of two */,
                                                               uint32_t* __restrict__ longq /* n per column: queue of the long scalars */,
                                                               unsigned long long* __restrict__ recs /* n per column: short-scalar records */,
                                                               uint32_t fine_bits, uint32_t idx_bits, uint32_t bin_cap) {
  extern __shared__ uint32_t sh[];
  const uint32_t B = 1u << (c - 1);
  uint32_t* hist = sh;            // B
  uint32_t* cursor = sh + B;      // B
  uint32_t* ucnt = sh + 2 * B;    // B
  uint32_t* wave_sums = sh + 3 * B;  // 18
  __shared__ uint32_t s_base, s_rbase, s_qcnt;
  const uint32_t col = blockIdx.x, tid = threadIdx.x;
  const u256* sc = scalars ? scalars + (size_t)col * n : nullptr;
  const ColSrc* cs = srcs ? srcs + col : nullptr;
  const uint8_t* mk = skip_mask ? skip_mask + (size_t)col * n : nullptr;
  for (uint32_t b = tid; b < B; b += MSM_SORT_THREADS) hist[b] = 0;
  __syncthreads();
  uint32_t* queue = longq + (size_t)col * n;
  unsigned long long* rec = recs + (size_t)col * n;
  walk_scalars(sc, cs, n_blind, mk, n, c, W, queue, rec, &s_qcnt, true, [&](uint32_t, uint32_t d, bool, size_t) { atomicAdd(&hist[d - 1], 1u); }, fine_bits != 0);
  __syncthreads();
  // scan: thread owns buckets [tid*ipt, (tid+1)*ipt)
  const uint32_t ipt = (B + MSM_SORT_THREADS - 1) / MSM_SORT_THREADS;
  uint32_t cnt_local = 0;
  for (uint32_t q = 0; q < ipt; q++) {
    uint32_t b = tid * ipt + q;
    if (b < B) cnt_local += hist[b];
  }
  uint32_t total_cnt, total_ua;
  const uint32_t cnt_pre0 = block_exclusive_scan(cnt_local, wave_sums, &total_cnt);
  // buckets whose first entry is not LCAP-aligned open one extra segment
  uint32_t ua_local = 0;
  {
    uint32_t off = cnt_pre0;
    for (uint32_t q = 0; q < ipt; q++) {
      uint32_t b = tid * ipt + q;
      if (b < B) {
        if (hist[b] && (off % lcap)) ua_local++;
        off += hist[b];
      }
    }
  }
  const uint32_t ua_pre0 = block_exclusive_scan(ua_local, wave_sums, &total_ua);
  const uint32_t nranges = (total_cnt + lcap - 1) / lcap, nseg = nranges + total_ua;
  if (tid == 0) {
    uint32_t base = atomicAdd(&counters[0], nseg);
    uint32_t rb = atomicAdd(&counters[3], nranges);
    if (base + nseg > seg_cap || rb + nranges > range_cap || total_cnt > ent_cap) {
      atomicExch(&counters[1], 1u);
      base = 0xffffffffu;
    }
    s_base = base;
    s_rbase = rb;
  }
  __syncthreads();
  const uint32_t base = s_base, rbase = s_rbase;
  uint32_t* soff = seg_off + (size_t)col * (B + 1);
  uint32_t* boff = bucket_off + (size_t)col * (B + 1);
  if (base == 0xffffffffu) {  // overflow: publish an empty column so later kernels stay in bounds
    for (uint32_t b = tid; b <= B; b += MSM_SORT_THREADS) {
      soff[b] = 0;
      boff[b] = 0;
    }
    return;
  }
  {
    uint32_t off = cnt_pre0, ua = ua_pre0;
    for (uint32_t q = 0; q < ipt; q++) {
      uint32_t b = tid * ipt + q;
      if (b < B) {
        uint32_t cnt = hist[b];
        cursor[b] = off;
        ucnt[b] = ua;
        boff[b] = off;
        soff[b] = base + (off + lcap - 1) / lcap + ua;
        if (cnt) {
          uint32_t segs = (off + cnt + lcap - 1) / lcap - off / lcap;
          if (segs > 1) atomicMax(&counters[2], segs);  // lets k_msm_combine skip passes nobody needs
          if (off % lcap) ua++;
        }
        off += cnt;
      }
    }
  }
  if (tid == 0) {
    soff[B] = base + nseg;
    boff[B] = total_cnt;
  }
  __syncthreads();
  // range descriptors: starting bucket by binary search over the (monotone) bucket offsets in LDS
  for (uint32_t r = tid; r < nranges; r += MSM_SORT_THREADS) {
    const uint32_t x = r * lcap;
    uint32_t lo_b = 0, hi_b = B;  // last b with cursor[b] <= x
    while (hi_b - lo_b > 1) {
      uint32_t mid = (lo_b + hi_b) >> 1;
      if (cursor[mid] <= x) lo_b = mid;
      else hi_b = mid;
    }
    const uint32_t bo = cursor[lo_b];
    MsmRange rg;
    rg.col = col;
    rg.bucket = lo_b;
    rg.start = x;
    rg.len = total_cnt - x < lcap ? total_cnt - x : lcap;
    rg.seg0 = base + r + ucnt[lo_b] + ((bo % lcap) && bo < x ? 1u : 0u);
    rg.pad = 0;
    ranges[rbase + r] = rg;
  }
  __syncthreads();
  uint32_t* ent = entries + (size_t)col * ent_cap;
  // Two-phase scatter (fine_bits > 0): 1.2 M four-byte stores to 8,192 open bucket runs per column never merge in L2 (a workgroup per CU,
  // every one with thousands of partly written lines) and cost two thirds of this kernel.  Here the entries go to the few hundred runs of
  // the coarse bins — bucket >> fine_bits, contiguous because a bin is a range of buckets — carrying their fine bucket number above the
  // table index; k_msm_scatter2 then orders every bin in place through LDS.
  if (fine_bits) {
    const uint32_t n_bins = B >> fine_bits;
    for (uint32_t q = tid; q < n_bins; q += MSM_SORT_THREADS) {
      const uint32_t lo = cursor[q << fine_bits], hi = q + 1 < n_bins ? cursor[(q + 1) << fine_bits] : total_cnt;
      ucnt[q] = lo;                          // the bins' cursors (ucnt is free: the range descriptors are written)
      hist[q] = hi - lo > bin_cap ? 1u : 0u; // a bin too large for a workgroup of the second phase (the top window's few small digits;
                                             // one value repeated down a witness column) is scattered directly: its runs are few
    }
    __syncthreads();
    const uint32_t fmask = (1u << fine_bits) - 1u;
    walk_scalars(sc, cs, n_blind, mk, n, c, W, queue, rec, &s_qcnt, false, [&](uint32_t j, uint32_t d, bool neg, size_t i) {
      const uint32_t b = d - 1, bin = b >> fine_bits;
      const uint32_t payload = (uint32_t)(j * table_n + i) | (neg ? 0x80000000u : 0u);
      if (hist[bin]) {
        ent[atomicAdd(&cursor[b], 1u)] = payload;
      } else {
        ent[atomicAdd(&ucnt[bin], 1u)] = payload | ((b & fmask) << idx_bits);
      }
    }, true);
    return;
  }
  walk_scalars(sc, cs, n_blind, mk, n, c, W, queue, rec, &s_qcnt, false, [&](uint32_t j, uint32_t d, bool neg, size_t i) {
    uint32_t pos = atomicAdd(&cursor[d - 1], 1u);
    ent[pos] = (uint32_t)(j * table_n + i) | (neg ? 0x80000000u : 0u);
  });
}

// second phase of the two-phase scatter: one workgroup per (bin, column) loads the bin's entries (coarse order) into LDS and writes
// them back bucket by bucket — 2^fine_bits open runs per workgroup, which L2 merges into whole lines
#define MSM_SCATTER2_THREADS 256
#define MSM_BIN_CAP (6 * 1024)   // entries of a bin a workgroup holds (24 registers per thread, 24 KB of LDS)
__global__ __launch_bounds__(MSM_SCATTER2_THREADS) void k_msm_scatter2(uint32_t* __restrict__ entries, size_t ent_cap, const uint32_t* __restrict__ bucket_off,
                                                                       uint32_t B, uint32_t fine_bits, uint32_t idx_bits, uint32_t bin_cap,
                                                                       const uint32_t* __restrict__ counters) {
  extern __shared__ uint32_t sh2[];
  if (counters[1]) return;
  const uint32_t col = blockIdx.y, tid = threadIdx.x;
  const uint32_t* bo = bucket_off + (size_t)col * (B + 1);
  const uint32_t nf = 1u << fine_bits, n_bins = B >> fine_bits;
  uint32_t* outb = sh2;           // bin_cap: the bin in bucket order, written back in whole lines
  uint32_t* fcur = sh2 + bin_cap; // nf
  uint32_t* ent = entries + (size_t)col * ent_cap;
  const uint32_t fmask = nf - 1u, imask = (1u << idx_bits) - 1u;
  const uint32_t lane = tid & 63u;
  for (uint32_t bin = blockIdx.x; bin < n_bins; bin += gridDim.x) {   // (uniform per workgroup: the barriers below are reached by all)
    const uint32_t b0 = bin << fine_bits;
    const uint32_t lo = bo[b0], hi = bo[b0 + nf];
    const uint32_t S = hi - lo;
    if (S == 0 || S > bin_cap) continue;   // (a larger bin was scattered to its buckets directly)
    // the bin as it lies (coarse order), in registers: every load of the thread in flight at once (a loop of unknown length would
    // wait for each one: ~20 round trips to HBM per bin)
    uint32_t r[MSM_BIN_CAP / MSM_SCATTER2_THREADS];
#pragma unroll
    for (uint32_t q = 0; q < MSM_BIN_CAP / MSM_SCATTER2_THREADS; q++) {
      const uint32_t i = tid + q * MSM_SCATTER2_THREADS;
      r[q] = i < S ? ent[lo + i] : 0u;
    }
    for (uint32_t f = tid; f < nf; f += MSM_SCATTER2_THREADS) fcur[f] = bo[b0 + f] - lo;
    __syncthreads();
    // positions by wavefront-level ranking: the lanes that hold the same bucket are found with one ballot per bucket bit, their
    // leader draws the run of positions with one atomic, the others take theirs by rank
#pragma unroll
    for (uint32_t q = 0; q < MSM_BIN_CAP / MSM_SCATTER2_THREADS; q++) {
      const uint32_t i = tid + q * MSM_SCATTER2_THREADS;
      if (q * MSM_SCATTER2_THREADS >= S) break;        // uniform
      const bool active = i < S;
      const uint32_t e = r[q];
      const uint32_t f = (e >> idx_bits) & fmask;
      unsigned long long peers = __ballot(active);
      for (uint32_t bit = 0; bit < fine_bits; bit++) {
        const unsigned long long m = __ballot(active && ((f >> bit) & 1u));
        peers &= ((f >> bit) & 1u) ? m : ~m;
      }
      uint32_t base = 0;
      const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
      if (active && rank == 0) base = atomicAdd(&fcur[f], (uint32_t)__popcll(peers));
      base = (uint32_t)__shfl((int)base, active ? (int)__ffsll((long long)peers) - 1 : (int)lane, 64);
      if (active) outb[base + rank] = (e & 0x80000000u) | (e & imask);
    }
    __syncthreads();
    for (uint32_t i = tid; i < S; i += MSM_SCATTER2_THREADS) ent[lo + i] = outb[i];
    __syncthreads();
  }
}

// ---- accumulation in nine-limb form ------------------------------------------------------------------------------
// Inside k_msm_accum coordinates are L9 values in Montgomery form with R' = 2^261 (what a 9 x 29-bit product divides
// by), so products need no operand shift, additions and subtractions are carry-free, and nothing is packed / split
// between the ten products of a mixed addition.  Bounds (multiples of q) are tracked in the comments of madd_l9.
struct MsmL9Consts {
  uint32_t c2[9], c8[9];  // 2q and 8q with dominating limbs (l9_sub)
  u256 one_rp;            // 2^261 mod q            (1 in R' form)
  u256 to_std;            // 2^256 mod q: x R' form -> standard Montgomery form (R = 2^256)
  u256 to_rp;             // 2^266 mod q: standard form -> R' form
};
struct AccL9 {
  L9 x, y, zz, zzz;  // exactly normalised; x < 7.5 q, y < 3.6 q, zz, zzz < 1.1 q
};

// rare exact path, acc == p as points: acc = 2p from the affine point   [mdbl-2008-s-1, 5M + 2S]
// x, y exactly normalised, x < q, y <= 2q
__device__ __forceinline__ void mdbl_l9(AccL9& acc, bool& ident, const L9& x, const L9& y, const MsmL9Consts& K) {
  if (l9_is_zero_mod<Fq>(y)) {  // 2-torsion (not on BN254 G1; kept for exactness of the group law)
    ident = true;
    return;
  }
  const L9 u = l9_add(y, y);               // limbs < 2 * 2^29, value <= 4q
  L9 un = u;
  l9_renorm(un);
  const L9 v = l9_sqr<Fq>(un);             // < 1.1 q
  const L9 w = l9_mul<Fq>(u, v);           // < 1.03 q
  const L9 sv = l9_mul<Fq>(x, v);          // < 1.01 q
  const L9 xx = l9_mul<Fq>(x, x);          // < 1.01 q
  const L9 m = l9_add(l9_add(xx, xx), xx); // limbs < 3 * 2^29, value < 3.1 q
  L9 mn = m;
  l9_renorm(mn);
  const L9 mm = l9_sqr<Fq>(mn);            // < 1.06 q
  const L9 ns = l9_neg(sv, K.c2);
  L9 x3 = l9_add(l9_add(mm, ns), ns);      // mm - 2 sv + 4q: limbs < 5 * 2^29, value < 5.1 q
  l9_carry(x3);
  const L9 td = l9_sub(sv, x3, K.c8);      // value < 9.1 q
  const L9 m1 = l9_mul<Fq>(td, mn);        // < 1.2 q
  const L9 m2 = l9_mul<Fq>(w, y);          // < 1.02 q
  L9 y3 = l9_sub(m1, m2, K.c2);
  l9_carry(y3);
  acc.x = x3;
  acc.y = y3;
  acc.zz = v;
  acc.zzz = w;
}

// acc += (neg ? -p : p), p affine in R' form and not the identity   [madd-2008-s, 8M + 2S]
__device__ __forceinline__ void madd_l9(AccL9& acc, bool& ident, const Affine& p, bool neg, const MsmL9Consts& K) {
  L9 x2 = l9_split(p.x), y2 = l9_split(p.y);  // canonical: exactly normalised, below q
  if (neg) {
    y2 = l9_neg(y2, K.c2);  // 2q - y: limbs below 2 * 2^29
    l9_carry(y2);
  }
  if (ident) {
    acc.x = x2;
    acc.y = y2;
    acc.zz = l9_split(K.one_rp);
    acc.zzz = acc.zz;
    ident = false;
    return;
  }
  const L9 u2 = l9_mul<Fq>(acc.zz, x2);    // < 1.01 q
  const L9 s2 = l9_mul<Fq>(acc.zzz, y2);   // < 1.02 q
  L9 pd = l9_sub(u2, acc.x, K.c8);         // u2 - x1 + 8q: limbs < 3 * 2^29, value < 9.1 q
  L9 rd = l9_sub(s2, acc.y, K.c8);         // s2 - y1 + 8q: same bounds
  L9 pn = pd, rn = rd;
  l9_renorm(pn);
  l9_renorm(rn);
  const L9 pp = l9_sqr<Fq>(pn);            // < 1.5 q
  const L9 r2 = l9_sqr<Fq>(rn);            // < 1.5 q
  if (l9_is_zero_mod<Fq>(pp)) {            // same x (exact test on the product: q is prime)
    if (l9_is_zero_mod<Fq>(r2)) mdbl_l9(acc, ident, x2, y2, K);  // same point
    else ident = true;                                            // opposite points
    return;
  }
  const L9 ppp = l9_mul<Fq>(pd, pp);       // < 1.09 q
  const L9 qq = l9_mul<Fq>(acc.x, pp);     // < 1.07 q
  const L9 nq = l9_neg(qq, K.c2);          // 2q - qq
  L9 x3 = l9_add(l9_add(l9_sub(r2, ppp, K.c2), nq), nq);  // r2 - ppp - 2 qq + 6q: limbs < 6.9 * 2^29, value < 7.5 q
  l9_carry(x3);
  const L9 td = l9_sub(qq, x3, K.c8);      // qq - x3 + 8q: limbs < 3 * 2^29, value < 9.1 q
  // y3 = td * rn - y1 * ppp as ONE reduced sum of two products: td * rn + (8q - y1) * ppp  (limbs 3 + 2 units,
  // values 9.1 q * 9.1 q + 8 q * 1.1 q: the result is exactly normalised and below 1.6 q)
  const L9 ny = l9_neg(acc.y, K.c8);
  const L9 y3 = l9_mul2<Fq>(td, rn, ny, ppp);
  acc.zz = l9_mul<Fq>(acc.zz, pp);
  acc.zzz = l9_mul<Fq>(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}
// A segment's partial sum leaves k_msm_accum as raw limbs (36 words; zz = 0 encodes the identity): the conversion to
// the standard form costs four products, and inside the accumulation loop a flush by ANY lane of a wavefront makes the
// whole wavefront walk the flush code, so it has to be cheap.  k_msm_partials converts all segments afterwards.
#define MSM_RAW_WORDS 36
__device__ __forceinline__ void flush_l9(uint4* dst, const AccL9& acc, bool ident) {
  const uint32_t z = ident ? 0u : 0xffffffffu;
  dst[0] = make_uint4(acc.x.l[0], acc.x.l[1], acc.x.l[2], acc.x.l[3]);
  dst[1] = make_uint4(acc.x.l[4], acc.x.l[5], acc.x.l[6], acc.x.l[7]);
  dst[2] = make_uint4(acc.y.l[0], acc.y.l[1], acc.y.l[2], acc.y.l[3]);
  dst[3] = make_uint4(acc.y.l[4], acc.y.l[5], acc.y.l[6], acc.y.l[7]);
  dst[4] = make_uint4(acc.zz.l[0] & z, acc.zz.l[1] & z, acc.zz.l[2] & z, acc.zz.l[3] & z);
  dst[5] = make_uint4(acc.zz.l[4] & z, acc.zz.l[5] & z, acc.zz.l[6] & z, acc.zz.l[7] & z);
  dst[6] = make_uint4(acc.zzz.l[0], acc.zzz.l[1], acc.zzz.l[2], acc.zzz.l[3]);
  dst[7] = make_uint4(acc.zzz.l[4], acc.zzz.l[5], acc.zzz.l[6], acc.zzz.l[7]);
  dst[8] = make_uint4(acc.x.l[8], acc.y.l[8], acc.zz.l[8] & z, acc.zzz.l[8]);
}
__global__ __launch_bounds__(256) void k_msm_partials(const uint4* __restrict__ raw, XYZZ* __restrict__ partials, const uint32_t* __restrict__ counters,
                                                      uint32_t seg_cap, u256 to_std) {
  if (counters[1]) return;
  const uint32_t total = counters[0] < seg_cap ? counters[0] : seg_cap;
  const uint32_t stride = gridDim.x * blockDim.x;
  const L9 ts = l9_split(to_std);
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint4* src = raw + (size_t)t * (MSM_RAW_WORDS / 4);
    uint4 w[9];
#pragma unroll
    for (int i = 0; i < 9; i++) w[i] = src[i];
    L9 c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c[j].l[0] = w[2 * j].x; c[j].l[1] = w[2 * j].y; c[j].l[2] = w[2 * j].z; c[j].l[3] = w[2 * j].w;
      c[j].l[4] = w[2 * j + 1].x; c[j].l[5] = w[2 * j + 1].y; c[j].l[6] = w[2 * j + 1].z; c[j].l[7] = w[2 * j + 1].w;
    }
    c[0].l[8] = w[8].x; c[1].l[8] = w[8].y; c[2].l[8] = w[8].z; c[3].l[8] = w[8].w;
    uint32_t zz_any = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) zz_any |= c[2].l[k];
    XYZZ a;
    if (!zz_any) {
      a = xyzz_identity();
    } else {
      a.x = l9_canon<Fq>(l9_mul<Fq>(c[0], ts));
      a.y = l9_canon<Fq>(l9_mul<Fq>(c[1], ts));
      a.zz = l9_canon<Fq>(l9_mul<Fq>(c[2], ts));
      a.zzz = l9_canon<Fq>(l9_mul<Fq>(c[3], ts));
    }
    st_xyzz(partials + t, a);
  }
}

// One thread per range: exactly `lcap` mixed additions per lane (full lane utilisation); the accumulator is
// flushed to the next segment slot whenever the sorted entry list moves on to another bucket.
__global__ __launch_bounds__(256) void k_msm_accum(const Affine* __restrict__ table, const uint32_t* __restrict__ entries, size_t ent_cap,
                                                   const MsmRange* __restrict__ ranges, const uint32_t* __restrict__ bucket_off, uint32_t B,
                                                   const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ counters, uint4* __restrict__ raw,
                                                   MsmSegInfo* __restrict__ seginfo, uint32_t range_cap, MsmL9Consts K) {
  if (counters[1]) return;  // sort overflowed (never expected: capacities are worst-case)
  const uint32_t total = counters[3] < range_cap ? counters[3] : range_cap;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    MsmRange rg = ranges[t];
    const uint32_t* e = entries + (size_t)rg.col * ent_cap;
    const uint32_t* bo = bucket_off + (size_t)rg.col * (B + 1);
    const uint32_t* so = seg_off + (size_t)rg.col * (B + 1);
    uint32_t b = rg.bucket, seg = rg.seg0, next = bo[b + 1];
    AccL9 acc;
    bool ident = true;
    // software prefetch: the gather of the next table point is in flight during the current addition
    uint32_t v = e[rg.start];
    Affine nxt = ld_affine(table + (v & 0x7fffffffu));
    for (uint32_t q = 0; q < rg.len; q++) {
      const uint32_t pos = rg.start + q;
      if (pos == next) {  // bucket boundary: close the segment
        flush_l9(raw + (size_t)seg * (MSM_RAW_WORDS / 4), acc, ident);
        {
          const uint32_t s0 = so[b];
          seginfo[seg].idx = seg - s0;
          seginfo[seg].len = so[b + 1] - s0;
        }
        seg++;
        ident = true;
        do {
          b++;
          next = bo[b + 1];
        } while (next == pos);  // skip empty buckets
      }
      Affine p = nxt;
      bool neg = (v >> 31) != 0;
      if (q + 1 < rg.len) {
        v = e[pos + 1];
        nxt = ld_affine(table + (v & 0x7fffffffu));
      }
      if (!affine_is_identity(p)) madd_l9(acc, ident, p, neg, K);
    }
    flush_l9(raw + (size_t)seg * (MSM_RAW_WORDS / 4), acc, ident);
    {
      const uint32_t s0 = so[b];
      seginfo[seg].idx = seg - s0;
      seginfo[seg].len = so[b + 1] - s0;
    }
  }
}

// Segmented tree reduction of the partial sums of every bucket: in pass p a bucket that still holds
// len_p = ceil(len / 2^p) > 1 partials folds its upper half onto its lower half.  A thread needs only its own
// (idx, len) record — no lookups — so a pass in which a segment has nothing to do costs one coalesced 8-byte read, and
// the lanes that do work are contiguous.  After ceil(log2(max segments per bucket)) passes the sum of bucket b sits in
// its first segment slot.  This is what removes the witness-column skew from the reduce kernel.
// (Measured alternatives: radix-8 in-place folding — 2x slower, one active lane in eight; one thread per bucket — 8x
// slower, serial tails of the heavy buckets.)
__global__ __launch_bounds__(256) void k_msm_combine(XYZZ* __restrict__ partials, const MsmSegInfo* __restrict__ seginfo, const uint32_t* __restrict__ counters,
                                                     uint32_t pass, uint32_t task_cap) {
  if (counters[1]) return;
  if ((1u << pass) >= counters[2]) return;  // every bucket is already folded to one partial
  const uint32_t total = counters[0] < task_cap ? counters[0] : task_cap;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const MsmSegInfo si = seginfo[t];
    const uint32_t len = (si.len + (1u << pass) - 1) >> pass, half = (len + 1) >> 1;
    if (len <= 1 || si.idx + half >= len) continue;
    XYZZ a = ld_xyzz(partials + t), b = ld_xyzz(partials + t + half);
    xyzz_add(a, b);
    st_xyzz(partials + t, a);
  }
}

__device__ __forceinline__ u256 shfl_u256(const u256& v, int src) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = (uint32_t)__shfl((int)v.w[i], src, 64);
  return r;
}
__device__ __forceinline__ XYZZ shfl_xyzz(const XYZZ& p, int src) {
  XYZZ r;
  r.x = shfl_u256(p.x, src);
  r.y = shfl_u256(p.y, src);
  r.zz = shfl_u256(p.zz, src);
  r.zzz = shfl_u256(p.zzz, src);
  return r;
}

// One wavefront per column: result = sum_b b * B_b with B_b = sum of the partials of bucket b.
__global__ __launch_bounds__(64) void k_msm_reduce(const XYZZ* __restrict__ partials, const uint32_t* __restrict__ task_off, uint32_t c,
                                                   const uint32_t* __restrict__ counters, const Affine* __restrict__ add_points, Affine* __restrict__ out) {
  if (counters[1]) return;
  const uint32_t B = 1u << (c - 1);
  const uint32_t col = blockIdx.x, lane = threadIdx.x;
  const uint32_t log_per = c - 1 >= 6 ? c - 7 : 0, per = 1u << log_per;
  const uint32_t* toff = task_off + (size_t)col * (B + 1);
  XYZZ running = xyzz_identity(), total = xyzz_identity();
  const uint32_t lo = lane * per;  // bucket indices [lo, lo+per) <-> bucket ids lo+1 .. lo+per
  if (lo < B) {
    for (int b = (int)(lo + per) - 1; b >= (int)lo; b--) {
      uint32_t t0 = toff[b], t1 = toff[b + 1];
      if (t1 > t0) {  // bucket sum was folded into its first slot by k_msm_combine
        XYZZ p = ld_xyzz(partials + t0);
        xyzz_add(running, p);
      }
      xyzz_add(total, running);
    }
  }
  // total_L = sum (b - lo) * B_b over the lane's ids; running_L = sum B_b.
  // result = sum_L total_L + per * sum_L L * running_L
  // suffix scan of running over lanes: suf_L = sum_{L' >= L} running_L'
  XYZZ suf = running;
  for (int o = 1; o < 64; o <<= 1) {
    XYZZ other = shfl_xyzz(suf, (int)lane + o < 64 ? (int)lane + o : (int)lane);
    if ((int)lane + o < 64) xyzz_add(suf, other);
  }
  // sum_{L>=1} suf_L = sum_L L * running_L
  XYZZ t2 = lane >= 1 ? suf : xyzz_identity();
  XYZZ t1 = total;
  for (int o = 32; o >= 1; o >>= 1) {
    XYZZ a = shfl_xyzz(t1, (int)lane + o < 64 ? (int)lane + o : (int)lane);
    XYZZ b = shfl_xyzz(t2, (int)lane + o < 64 ? (int)lane + o : (int)lane);
    if ((int)lane < o) {
      xyzz_add(t1, a);
      xyzz_add(t2, b);
    }
  }
  if (lane == 0) {
    for (uint32_t d = 0; d < log_per; d++) t2 = xyzz_double(t2);
    xyzz_add(t1, t2);
    if (add_points) xyzz_add_mixed(t1, ld_affine(add_points + col), false);  // + precomputed constant-cell part
    st_affine(out + col, xyzz_to_affine(t1));
  }
}


// ---------------------------------------------------------------- test / bench SRS ("unsafe" setup with a known tau)
// g[i] = tau^i * G,  g_lagrange[i] = L_i(tau) * G with L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i))
__global__ __launch_bounds__(64) void k_srs_setup(u256 tau, u256 omega, u256 tn_minus_1_over_n, uint64_t n, Affine* __restrict__ g, Affine* __restrict__ gl) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine gen;
  gen.x = to_mont<Fq>(u256_from_u64(1));
  gen.y = to_mont<Fq>(u256_from_u64(2));
  u256 e = u256_from_u64(i);
  u256 ti = mont_pow<Fr>(tau, e), wi = mont_pow<Fr>(omega, e);
  u256 li = fr_mul(fr_mul(wi, tn_minus_1_over_n), mont_inv<Fr>(fr_sub(tau, wi)));
  for (int which = 0; which < 2; which++) {
    u256 k = from_mont<Fr>(which == 0 ? ti : li);
    XYZZ acc = xyzz_identity();
    for (int b = (int)u256_bits(k) - 1; b >= 0; b--) {
      acc = xyzz_double(acc);
      if (u256_bit(k, (unsigned)b)) xyzz_add_mixed(acc, gen, false);
    }
    st_affine((which == 0 ? g : gl) + i, xyzz_to_affine(acc));
  }
}

// out[c] = sum_m parts[m][c]: the combine step of a point-sharded MSM (each rank commits its slice of the rows of every
// column; the partial commitments are gathered and added — SURVEY §8e "alternative")
__global__ __launch_bounds__(64) void k_g1_sum(const Affine* __restrict__ parts, uint32_t m, uint64_t n, Affine* __restrict__ out) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  XYZZ acc = xyzz_identity();
  for (uint32_t i = 0; i < m; i++) xyzz_add_mixed(acc, ld_affine(parts + (uint64_t)i * n + c), false);
  st_affine(out + c, xyzz_to_affine(acc));
}

static uint32_t pick_window(uint32_t k) {
  const char* env = getenv("VDB_MSM_C");
  if (env) {
    int v = atoi(env);
    if (v >= 2 && v <= 14) return (uint32_t)v;
  }
  // measured on the C4 witness columns (profiles/): per-task overhead of the many light buckets outweighs the
  // extra digits of a smaller window; c = 11 is the optimum for k = 16
  if (k <= 8) return 8;
  return 11;
}

// device-level batched MSM: scalars_dev = n_cols x n (contiguous), out_dev = n_cols affine points
// `defer_tail`: the bucket folding of the LAST batch (k_msm_combine, k_msm_reduce: short latency-bound launches that
// leave most of the chip idle) goes to the context's auxiliary stream and the function returns without waiting;
// msm_collect() joins.  The scalars are no longer read at that point, so the caller may overwrite them (NTT in place).
int msm_batch_dev(const vdb_srs* srs, int basis, const u256* scalars_dev, size_t n_cols, size_t n, Affine* out_dev, const uint8_t* skip_mask = nullptr,
                  const Affine* add_points = nullptr, bool defer_tail = false, const ColSrc* srcs = nullptr, uint32_t n_blind = 0) {
  Context& cx = ctx();
  if (g_prof_mode == 1) defer_tail = false;  // per-kernel timing serialises on the main stream
  if (cx.msm_pending) {
    set_error("msm: a deferred batch has not been collected (vdb_msm_batch_end)");
    return VDB_ERR_ARG;
  }
  if (n_cols == 0) return VDB_OK;
  const Affine* table = srs->table[basis];
  const uint32_t c = srs->c, W = srs->W, B = srs->B;
  const size_t ent_cap = n * W;
  uint32_t lcap = MSM_LCAP_MIN;
  while (lcap < MSM_LCAP_MAX && (n_cols * n) / lcap >= ((size_t)1 << 19)) lcap <<= 1;  // keep >= ~2^20 ranges in a large job
  const size_t range_cap_col = (ent_cap + lcap - 1) / lcap;  // worst case: every digit non-zero
  const size_t seg_cap_col = B + range_cap_col;
  // large batches keep thousands of independent column reductions in flight
  size_t per_col = ent_cap * 4 + range_cap_col * sizeof(MsmRange) + seg_cap_col * (sizeof(MsmSegInfo) + sizeof(XYZZ) + MSM_RAW_WORDS * 4) + 2 * (B + 1) * 4 +
                   n * 12;
  // two-phase scatter of the sort: bins of 2^fine_bits buckets; the table index and the fine bucket number share an entry's 31 bits
  static const bool two_phase_on = !(getenv("VDB_MSM_TWO_PHASE") && getenv("VDB_MSM_TWO_PHASE")[0] == '0');
  uint32_t idx_bits = 1;
  while (((uint64_t)1 << idx_bits) < (uint64_t)W * srs->n) idx_bits++;
  uint32_t fine_bits = 0;
  const uint32_t bin_cap = MSM_BIN_CAP;
  // (only where the direct scatter is what the sort waits for: thousands of buckets per column and dense scalars — the 14-bit windows
  //  of the product / quotient / fixed columns; with the 11-bit windows of the witness columns, 1,024 buckets and mostly short scalars,
  //  the second phase costs more than it saves: 15.4 + 4.2 against 15.2 ms per C4 step)
  if (two_phase_on && c >= 13) {
    fine_bits = c - 1 - 8;              // 256 bins
    if (fine_bits + idx_bits > 31) fine_bits = idx_bits < 31 ? 31 - idx_bits : 0;
    if (fine_bits < 2) fine_bits = 0;
  }
  // budget: half of what is free on the card (counting the slot this buffer already holds), between 8 and 96 GiB —
  // on a 288 GB MI355X the 8,146 columns of the kmeans k = 16 job go through in ONE batch (76 GB), so the whole
  // bucket-folding tail can run beside the NTTs (defer_tail)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)80 << 30;
  size_t budget = (free_b + cx.scratch_bytes[2]) / 2;
  if (budget < ((size_t)8 << 30)) budget = (size_t)8 << 30;
  if (budget > ((size_t)96 << 30)) budget = (size_t)96 << 30;
  // a caller that is about to allocate tens of GB of its own (keygen) bounds the work space: growing it to half of a still empty
  // card and releasing it again costs seconds of page mapping (vdb_msm_set_scratch_cap)
  if (cx.msm_scratch_cap && budget > cx.msm_scratch_cap) budget = cx.msm_scratch_cap > cx.scratch_bytes[2] ? cx.msm_scratch_cap : cx.scratch_bytes[2];
  size_t nb = budget / per_col;
  if (nb < 1) nb = 1;
  if (nb > n_cols) nb = n_cols;
  if (nb > 16384) nb = 16384;
  if (nb * seg_cap_col > 0x7fffffffull) nb = 0x7fffffffull / seg_cap_col;
  uint8_t* buf = (uint8_t*)scratch_get(2, nb * per_col + 512);
  if (!buf) return VDB_ERR_OOM;
  uint32_t* entries = (uint32_t*)buf;
  XYZZ* partials = (XYZZ*)(buf + nb * ent_cap * 4);
  uint4* raw = (uint4*)((uint8_t*)partials + nb * seg_cap_col * sizeof(XYZZ));
  MsmRange* ranges = (MsmRange*)((uint8_t*)raw + nb * seg_cap_col * MSM_RAW_WORDS * 4);
  MsmSegInfo* seginfo = (MsmSegInfo*)((uint8_t*)ranges + nb * range_cap_col * sizeof(MsmRange));
  uint32_t* seg_off = (uint32_t*)((uint8_t*)seginfo + nb * seg_cap_col * sizeof(MsmSegInfo));
  uint32_t* bucket_off = seg_off + nb * (B + 1);
  uint32_t* counters = bucket_off + nb * (B + 1);
  unsigned long long* recs = (unsigned long long*)(((uintptr_t)(counters + 64) + 7) & ~(uintptr_t)7);  // n records per column
  uint32_t* longq = (uint32_t*)(recs + nb * n);                       // then n queue entries per column
  const uint32_t seg_cap = (uint32_t)(nb * seg_cap_col), range_cap = (uint32_t)(nb * range_cap_col);
  size_t lds = (3 * (size_t)B + 32) * sizeof(uint32_t);
  MsmL9Consts l9k;
  l9_offset_limbs<FqParams>(2, l9k.c2);
  l9_offset_limbs<FqParams>(8, l9k.c8);
  l9k.one_rp = to_mont<Fq>(u256_from_u64(32));
  l9k.to_std = mont_one<Fq>();
  l9k.to_rp = to_mont<Fq>(u256_from_u64(1024));
  for (size_t c0 = 0; c0 < n_cols; c0 += nb) {
    size_t nc = n_cols - c0 < nb ? n_cols - c0 : nb;
    VDB_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), cx.stream));
    {
      VDB_PROF("k_msm_sort");
      hipLaunchKernelGGL(k_msm_sort, dim3((unsigned)nc), dim3(MSM_SORT_THREADS), lds, cx.stream, scalars_dev ? scalars_dev + c0 * n : nullptr, srcs ? srcs + c0 : nullptr,
                       n_blind, n, srs->n, c, W, entries,
                       ent_cap, seg_off, bucket_off, ranges, counters, seg_cap, range_cap, skip_mask ? skip_mask + c0 * n : nullptr,
                       lcap, longq, recs, fine_bits, idx_bits, bin_cap);
    }
    VDB_LAUNCH_CHECK();
    if (fine_bits) {
      VDB_PROF("k_msm_scatter2");
      const unsigned bins_x = (B >> fine_bits) < 64u ? (B >> fine_bits) : 64u;   // (16 .. 256 workgroups per column measure the same)
      hipLaunchKernelGGL(k_msm_scatter2, dim3(bins_x, (unsigned)nc), dim3(MSM_SCATTER2_THREADS), (bin_cap + (1u << fine_bits)) * sizeof(uint32_t), cx.stream,
                       entries, ent_cap, bucket_off, B, fine_bits, idx_bits, bin_cap, counters);
    }
    VDB_LAUNCH_CHECK();
    {
      VDB_PROF("k_msm_accum");
      hipLaunchKernelGGL(k_msm_accum, dim3((unsigned)(cx.cu_count * 8)), dim3(256), 0, cx.stream, table, entries, ent_cap, ranges, bucket_off, B,
                       seg_off, counters, raw, seginfo, range_cap, l9k);
    }
    VDB_LAUNCH_CHECK();
    {
      VDB_PROF("k_msm_partials");
      hipLaunchKernelGGL(k_msm_partials, dim3((unsigned)(cx.cu_count * 8)), dim3(256), 0, cx.stream, raw, partials, counters, seg_cap, l9k.to_std);
    }
    VDB_LAUNCH_CHECK();
    hipStream_t ts = cx.stream;
    if (defer_tail && c0 + nc == n_cols) {
      VDB_HIP(hipEventRecord(cx.ev_tail, cx.stream));
      VDB_HIP(hipStreamWaitEvent(cx.aux, cx.ev_tail, 0));
      ts = cx.aux;
    }
    {
      // at most ceil(log2(max segments per bucket)) passes do work; the rest exit on the device-side maximum
      uint32_t max_nt = (uint32_t)range_cap_col + 1, passes = 0;
      while ((1u << passes) < max_nt) passes++;
      for (uint32_t ps = 0; ps < passes; ps++) {
        VDB_PROF_ON("k_msm_combine", ts);
        hipLaunchKernelGGL(k_msm_combine, dim3((unsigned)(cx.cu_count * 8)), dim3(256), 0, ts, partials, seginfo, counters, ps, seg_cap);
      }
      VDB_LAUNCH_CHECK();
    }
    {
      VDB_PROF_ON("k_msm_reduce", ts);
      hipLaunchKernelGGL(k_msm_reduce, dim3((unsigned)nc), dim3(64), 0, ts, partials, seg_off, c, counters, add_points ? add_points + c0 : nullptr, out_dev + c0);
    }
    VDB_LAUNCH_CHECK();
  }
  if (defer_tail) {
    cx.msm_pending = true;
    cx.msm_counters = counters;
    return VDB_OK;
  }
  uint32_t h_counters[2] = {0, 0};
  VDB_HIP(hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  if (h_counters[1]) {
    set_error("msm: internal task buffer overflow");
    return VDB_ERR_HIP;
  }
  return VDB_OK;
}

// joins a deferred batch: both streams drained, overflow flag checked
int msm_collect() {
  Context& cx = ctx();
  if (!cx.msm_pending) return VDB_OK;
  cx.msm_pending = false;
  if (!cx.msm_counters) {  // the batch ran synchronously (profiling): nothing left to check
    VDB_HIP(hipStreamSynchronize(cx.aux));
    VDB_HIP(hipStreamSynchronize(cx.stream));
    return VDB_OK;
  }
  // the auxiliary stream is ordered after the batch's main-stream kernels (ev_tail): reading the flag there does not wait
  // for whatever the caller queued on the main stream behind the MSM (the NTTs of the same columns)
  uint32_t h_counters[2] = {0, 0};
  VDB_HIP(hipMemcpyAsync(h_counters, cx.msm_counters, sizeof(h_counters), hipMemcpyDeviceToHost, cx.aux));
  cx.msm_counters = nullptr;
  VDB_HIP(hipStreamSynchronize(cx.aux));
  if (h_counters[1]) {
    set_error("msm: internal task buffer overflow");
    return VDB_ERR_HIP;
  }
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_srs_setup_unsafe(uint32_t k, const vdb_fr* tau, vdb_g1* g_out, vdb_g1* g_lagrange_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(tau && g_out && g_lagrange_out && k >= 1 && k <= 22, "bad argument");
  Context& cx = ctx();
  const uint64_t n = 1ull << k;
  u256 t;
  memcpy(&t, tau, 32);
  u256 omega = host_root_of_unity(k);
  u256 tn = mont_pow<Fr>(t, u256_from_u64(n));
  u256 c = fr_mul(fr_sub(tn, mont_one<Fr>()), mont_inv<Fr>(host_fr_from_u64(n)));
  Affine* d = (Affine*)scratch_get(0, 2 * n * sizeof(Affine));
  if (!d) return VDB_ERR_OOM;
  {
    VDB_PROF("k_srs_setup");
    hipLaunchKernelGGL(k_srs_setup, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, cx.stream, t, omega, c, n, d, d + n);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(g_out, d, n * sizeof(Affine), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipMemcpyAsync(g_lagrange_out, d + n, n * sizeof(Affine), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}

int vdb_srs_load(uint32_t k, const vdb_g1* g, const vdb_g1* g_lagrange, vdb_srs** out) { return vdb_srs_load_window(k, g, g_lagrange, 0, out); }
int vdb_srs_load_window(uint32_t k, const vdb_g1* g, const vdb_g1* g_lagrange, uint32_t window_bits, vdb_srs** out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(out && (g || g_lagrange) && k >= 1 && k <= 24, "bad argument");
  // the counting sort keeps three counters per bucket in LDS: 2^(c-1) buckets must fit 160 KB
  VDB_ARG(window_bits == 0 || (window_bits >= 2 && window_bits <= 14), "window_bits must be 0 (default) or 2..14");
  Context& cx = ctx();
  vdb_srs* s = new vdb_srs();
  s->device = cx.device;
  s->k = k;
  s->n = (size_t)1 << k;
  s->c = window_bits ? window_bits : pick_window(k);
  s->W = (254 + s->c - 1) / s->c;
  s->B = 1u << (s->c - 1);
  s->table[0] = s->table[1] = nullptr;
  if ((size_t)s->W * s->n > 0x7fffffffull) {
    delete s;
    set_error("srs: table index does not fit 31 bits");
    return VDB_ERR_ARG;
  }
  const vdb_g1* src[2] = {g, g_lagrange};
  for (int b = 0; b < 2; b++) {
    if (!src[b]) continue;
    Affine* bases = (Affine*)scratch_get(0, s->n * sizeof(Affine));
    if (!bases) {
      vdb_srs_free(s);
      return VDB_ERR_OOM;
    }
    hipError_t e = hipMalloc(&s->table[b], (size_t)s->W * s->n * sizeof(Affine));
    if (e != hipSuccess) {
      vdb_srs_free(s);
      return hip_fail(e, "hipMalloc(srs table)", __FILE__, __LINE__);
    }
    VDB_HIP(hipMemcpyAsync(bases, src[b], s->n * sizeof(Affine), hipMemcpyHostToDevice, cx.stream));
    {
      VDB_PROF("k_srs_table");
      hipLaunchKernelGGL(k_srs_table, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, cx.stream, bases, s->table[b], s->n, s->c, s->W);
    }
    VDB_LAUNCH_CHECK();
    VDB_HIP(hipStreamSynchronize(cx.stream));
  }
  *out = s;
  return VDB_OK;
}
int vdb_g1_sum(const vdb_g1* parts, size_t m, size_t n, vdb_g1* out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(parts && out && m >= 1 && m <= 0xffffffffull, "bad argument");
  if (n == 0) return VDB_OK;
  Context& cx = ctx();
  Affine* d = (Affine*)scratch_get(0, (m + 1) * n * sizeof(Affine));
  if (!d) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(d, parts, m * n * sizeof(Affine), hipMemcpyHostToDevice, cx.stream));
  {
    VDB_PROF("k_g1_sum");
    hipLaunchKernelGGL(k_g1_sum, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, cx.stream, d, (uint32_t)m, (uint64_t)n, d + m * n);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(out, d + m * n, n * sizeof(Affine), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}
void vdb_srs_free(vdb_srs* s) {
  if (!s) return;
  int cur = -1;
  const bool hop = hipGetDevice(&cur) == hipSuccess && cur != vdb::phys_of(s->device) && hipSetDevice(vdb::phys_of(s->device)) == hipSuccess;
  for (int b = 0; b < 2; b++)
    if (s->table[b]) (void)hipFree(s->table[b]);
  if (hop) (void)hipSetDevice(cur);
  delete s;
}
int vdb_srs_device(const vdb_srs* s, int* device) {
  VDB_ARG(s && device, "null pointer");
  *device = s->device;
  return VDB_OK;
}
int vdb_srs_info(const vdb_srs* s, uint32_t* k, uint32_t* window_bits, uint32_t* windows) {
  VDB_ARG(s, "null srs");
  if (k) *k = s->k;
  if (window_bits) *window_bits = s->c;
  if (windows) *windows = s->W;
  return VDB_OK;
}

// The results of a deferred batch stay in a buffer of their own until vdb_msm_batch_end: the shared scratch slots remain
// free for whatever the caller queues in between (slot 2, where the batch's bucket folding is still at work on the second
// stream, is refused to everybody else while the batch is open: scratch_get).
static void* deferred_out(size_t bytes) {
  Context& c = ctx();
  if (c.msm_out_bytes >= bytes && c.msm_out_buf) return c.msm_out_buf;
  if (c.msm_out_buf) (void)hipFree(c.msm_out_buf);
  c.msm_out_buf = nullptr;
  c.msm_out_bytes = 0;
  hipError_t e = hipMalloc(&c.msm_out_buf, bytes + bytes / 8 + 64);
  if (e != hipSuccess) {
    hip_fail(e, "hipMalloc(deferred MSM output)", __FILE__, __LINE__);
    return nullptr;
  }
  c.msm_out_bytes = bytes + bytes / 8 + 64;
  return c.msm_out_buf;
}
int vdb_msm_batch_dev(const vdb_srs* srs, int basis, const vdb_fr* scalars_dev, size_t n_cols, size_t n, vdb_g1* out_host) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && scalars_dev && out_host && (basis == 0 || basis == 1), "bad argument");
  VDB_SRS_HERE(srs);
  VDB_ARG(srs->table[basis], "srs was loaded without this basis");
  VDB_ARG(n <= srs->n && n > 0, "n exceeds the loaded SRS size (shorter columns are allowed)");
  if (n_cols == 0) return VDB_OK;
  Affine* dout = (Affine*)scratch_get(1, n_cols * sizeof(Affine));
  if (!dout) return VDB_ERR_OOM;
  int rc = msm_batch_dev(srs, basis, as_u256(scalars_dev), n_cols, n, dout);
  if (rc) return rc;
  VDB_HIP(hipMemcpyAsync(out_host, dout, n_cols * sizeof(Affine), hipMemcpyDeviceToHost, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_msm_count_entries_dev(const vdb_srs* srs, const vdb_fr* scalars_dev, size_t n_cols, size_t n, const uint8_t* skip_mask_dev, uint64_t* counts_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && scalars_dev && counts_out && n <= srs->n, "bad argument");
  VDB_SRS_HERE(srs);
  if (n_cols == 0) return VDB_OK;
  Context& cx = ctx();
  unsigned long long* d = (unsigned long long*)scratch_get(2, n_cols * sizeof(unsigned long long));
  if (!d) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(d, 0, n_cols * sizeof(unsigned long long), cx.stream));
  {
    VDB_PROF("k_msm_count_entries");
    hipLaunchKernelGGL(k_msm_count_entries, dim3((unsigned)n_cols), dim3(256), 0, cx.stream, as_u256(scalars_dev), skip_mask_dev, (uint64_t)n, srs->c, srs->W, d);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(counts_out, d, n_cols * sizeof(uint64_t), hipMemcpyDeviceToHost, cx.stream));
  VDB_HIP(hipStreamSynchronize(cx.stream));
  return VDB_OK;
}
int vdb_msm_batch_masked_dev_begin(const vdb_srs* srs, int basis, const vdb_fr* scalars_dev, size_t n_cols, size_t n, const uint8_t* skip_mask_dev,
                                   const vdb_g1* const_points_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && scalars_dev && (basis == 0 || basis == 1), "bad argument");
  VDB_SRS_HERE(srs);
  VDB_ARG(srs->table[basis], "srs was loaded without this basis");
  VDB_ARG(n <= srs->n && n > 0, "n exceeds the loaded SRS size");
  VDB_ARG((skip_mask_dev == nullptr) == (const_points_dev == nullptr), "mask and constant points go together");
  if (n_cols == 0) return VDB_OK;
  Affine* dout = (Affine*)deferred_out(n_cols * sizeof(Affine));
  if (!dout) return VDB_ERR_OOM;
  ctx().msm_out = dout;
  int rc = msm_batch_dev(srs, basis, as_u256(scalars_dev), n_cols, n, dout, skip_mask_dev, reinterpret_cast<const Affine*>(const_points_dev), true);
  if (rc == VDB_OK && !ctx().msm_pending) ctx().msm_pending = true;  // profiling mode ran it synchronously: _end still copies out
  return rc;
}
int vdb_msm_batch_src_dev_begin(const vdb_srs* srs, int basis, const vdb_colsrc* src_dev, size_t n_cols, size_t n, uint32_t n_blind,
                                const uint8_t* skip_mask_dev, const vdb_g1* const_points_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && src_dev && (basis == 0 || basis == 1), "bad argument");
  VDB_SRS_HERE(srs);
  VDB_ARG(srs->table[basis], "srs was loaded without this basis");
  VDB_ARG(n <= srs->n && n > 0 && n_blind <= n, "n exceeds the loaded SRS size");
  VDB_ARG((skip_mask_dev == nullptr) == (const_points_dev == nullptr), "mask and constant points go together");
  if (n_cols == 0) return VDB_OK;
  Affine* dout = (Affine*)deferred_out(n_cols * sizeof(Affine));
  if (!dout) return VDB_ERR_OOM;
  ctx().msm_out = dout;
  int rc = msm_batch_dev(srs, basis, nullptr, n_cols, n, dout, skip_mask_dev, reinterpret_cast<const Affine*>(const_points_dev), true,
                         reinterpret_cast<const ColSrc*>(src_dev), n_blind);
  if (rc == VDB_OK && !ctx().msm_pending) ctx().msm_pending = true;
  return rc;
}
int vdb_msm_batch_end(vdb_g1* out_host, size_t n_cols) {
  VDB_REQUIRE_INIT();
  VDB_ARG(out_host || n_cols == 0, "null pointer");
  int rc = msm_collect();
  if (rc) return rc;
  if (n_cols == 0) return VDB_OK;
  const void* dout = ctx().msm_out;
  VDB_ARG(dout, "no deferred MSM to collect");
  // on the auxiliary stream, where the points were produced: work queued on the main stream after _begin keeps running
  VDB_HIP(hipMemcpyAsync(out_host, dout, n_cols * sizeof(Affine), hipMemcpyDeviceToHost, ctx().aux));
  VDB_HIP(hipStreamSynchronize(ctx().aux));
  return VDB_OK;
}
int vdb_msm_batch_masked_dev(const vdb_srs* srs, int basis, const vdb_fr* scalars_dev, size_t n_cols, size_t n, const uint8_t* skip_mask_dev,
                             const vdb_g1* const_points_dev, vdb_g1* out_host) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && scalars_dev && out_host && (basis == 0 || basis == 1) && skip_mask_dev && const_points_dev, "bad argument");
  VDB_SRS_HERE(srs);
  VDB_ARG(srs->table[basis], "srs was loaded without this basis");
  VDB_ARG(n <= srs->n && n > 0, "n exceeds the loaded SRS size");
  if (n_cols == 0) return VDB_OK;
  Affine* dout = (Affine*)scratch_get(1, n_cols * sizeof(Affine));
  if (!dout) return VDB_ERR_OOM;
  int rc = msm_batch_dev(srs, basis, as_u256(scalars_dev), n_cols, n, dout, skip_mask_dev, reinterpret_cast<const Affine*>(const_points_dev));
  if (rc) return rc;
  VDB_HIP(hipMemcpyAsync(out_host, dout, n_cols * sizeof(Affine), hipMemcpyDeviceToHost, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_msm_batch(const vdb_srs* srs, int basis, const vdb_fr* const* cols, size_t n_cols, size_t n, vdb_g1* out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && cols && out, "null pointer");
  if (n_cols == 0) return VDB_OK;
  u256* d = (u256*)scratch_get(0, n_cols * n * sizeof(u256));
  if (!d) return VDB_ERR_OOM;
  for (size_t i = 0; i < n_cols; i++) VDB_HIP(hipMemcpyAsync(d + i * n, cols[i], n * sizeof(u256), hipMemcpyHostToDevice, ctx().stream));
  return vdb_msm_batch_dev(srs, basis, reinterpret_cast<const vdb_fr*>(d), n_cols, n, out);
}
int vdb_msm(const vdb_srs* srs, int basis, const vdb_fr* scalars, size_t n, vdb_g1* out) {
  const vdb_fr* cols[1] = {scalars};
  return vdb_msm_batch(srs, basis, cols, 1, n, out);
}

}  // extern "C"
