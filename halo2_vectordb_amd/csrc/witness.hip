// Witness generation (b5), stream->column layout (b4) and fixed-point staging (a2/a3/a18) on gfx950.
//
// Every kernel writes cells of the flat advice stream at statically known offsets (gadgets.hpp), so
// independent gadget instances — distance evaluations, Poseidon permutations, per-dimension folds —
// run as independent threads; long sequential gadgets are cut into position windows, one window per
// wavefront of 64 instances.  The stream is written once (32 B per cell: the HBM-write roofline of
// this stage) and read once by the layout kernel.
#include <cmath>

#include "common.hpp"
#include "gadgets.hpp"
#include "poseidon.hpp"

namespace vdb {

int poseidon_merkle_dev(const u256* vectors_dev, size_t n, size_t dim, u256* lv, u256* tmp);

// ------------------------------------------------------------------ fixed-point staging (host)
// fixed_point.rs:104-119: round(|x| * 2^P) as u128 (saturating), negative -> r - q
static u256 quantize_host(uint32_t P, double x) {
  bool neg = !std::isnan(x) && std::signbit(x);
  double y = std::round(std::fabs(x) * std::ldexp(1.0, (int)P));
  unsigned __int128 q;
  if (std::isnan(y) || y <= 0.0) q = 0;
  else if (y >= 340282366920938463463374607431768211456.0) q = ~(unsigned __int128)0;
  else q = (unsigned __int128)y;
  u256 c = u256_zero();
  for (int i = 0; i < 4; i++) c.w[i] = (uint32_t)(q >> (32 * i));
  u256 m = to_mont<Fr>(c);
  return neg ? fr_neg(m) : m;
}
// fixed_point.rs:121-136 (including the "-(|v| - 2)" quirk for negatives)
static double dequantize_host(uint32_t P, const u256& x) {
  u256 c = from_mont<Fr>(x);
  u256 np, t, one = u256_from_u64(1);
  u256_sub(np, mod_p<Fr>(), u256_shl(one, 2 * P + 1));
  double sign = 1.0;
  if (!u256_geq(np, c)) {  // x > negative_point
    u256 bm;
    u256_sub(bm, mod_p<Fr>(), one);                // bn254_max
    u256 xm = fr_sub(fr_sub(to_mont<Fr>(bm), x), mont_one<Fr>());
    c = from_mont<Fr>(xm);
    sign = -1.0;
  }
  (void)t;
  unsigned __int128 lo = 0;
  for (int i = 0; i < 4; i++) lo |= (unsigned __int128)c.w[i] << (32 * i);
  unsigned __int128 sc = (unsigned __int128)1 << P;
  double xi = (double)(lo / sc);
  double xf = (double)(lo % sc) / (double)sc;
  return sign * (xi + xf);
}

// ------------------------------------------------------------------ FixedPointChip tables
struct FpEntry {
  FpTables host;
  FpTables* dev;
  u256* limb_tab;
};
// cached per device in Context::fp_tables (released by witness_release on vdb_shutdown)
void witness_release(Context& c) {
  for (auto& kv : c.fp_tables) {
    FpEntry* e = static_cast<FpEntry*>(kv.second);
    if (e->limb_tab) (void)hipFree(e->limb_tab);
    if (e->dev) (void)hipFree(e->dev);
    delete e;
  }
  c.fp_tables.clear();
}

__global__ void k_limb_table(u256* tab, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tab[i] = to_mont<Fr>(u256_from_u64(i));
}

static int get_fp(uint32_t P, uint32_t L, FpEntry** out) {
  if (P < 32 || P > 63) {
    set_error("PRECISION_BITS must be in [32, 63] (fixed_point.rs:55-56)");
    return VDB_ERR_ARG;
  }
  if (L < 2 || L > 20) {
    set_error("lookup_bits must be in [2, 20]");
    return VDB_ERR_ARG;
  }
  uint64_t key = ((uint64_t)P << 32) | L;
  auto& cache = ctx().fp_tables;
  auto it = cache.find(key);
  if (it != cache.end()) {
    *out = static_cast<FpEntry*>(it->second);
    return VDB_OK;
  }
  // owned by the cache from here on, so that an error path below leaks nothing (vdb_shutdown frees what was allocated)
  FpEntry* e = new FpEntry();
  e->dev = nullptr;
  e->limb_tab = nullptr;
  cache[key] = e;
  struct Guard {
    std::map<uint64_t, void*>& cache;
    uint64_t key;
    FpEntry* e;
    bool keep = false;
    ~Guard() {
      if (keep) return;
      if (e->limb_tab) (void)hipFree(e->limb_tab);
      if (e->dev) (void)hipFree(e->dev);
      cache.erase(key);
      delete e;
    }
  } guard{cache, key, e};
  FpTables& T = e->host;
  memset(&T, 0, sizeof(T));
  T.P = P;
  T.L = L;
  T.one = mont_one<Fr>();
  T.pow2[0] = T.one;
  for (int i = 1; i < 254; i++) T.pow2[i] = fr_add(T.pow2[i - 1], T.pow2[i - 1]);
  T.scale = T.pow2[P];
  static const double exp2c[13] = {3.6240421303547230336183979205877e-11, 4.1284327467833130245549169910389e-10,
                                   0.0000000071086385644026346316624185550542, 0.00000010172297085296590958930245291448,
                                   0.0000013215904023658396206789543841996, 0.000015252713316417140696221389106544,
                                   0.00015403531076657894204857389177279, 0.0013333558131297097698435464957392,
                                   0.0096181291078409107025643582456283, 0.055504108664804181586140094858174,
                                   0.24022650695910142332414229540187, 0.69314718055994529934452147700678, 1.0};
  static const double logc[15] = {-3.319586265362338e-08, 1.4957235315170112e-06, -3.1350053389526744e-05,
                                  0.00040554177582512901, -0.0036218342998850703, 0.023663846121538389,
                                  -0.11691877183255484, 0.44524062371564499, -1.3195777548208449,
                                  3.0518128028712077, -5.4904626000399528, 7.6298580090181591,
                                  -8.1653313719804235, 7.1389971101896279, -3.1937385492842112};
  for (int i = 0; i < 13; i++) T.exp2_poly[i] = quantize_host(P, exp2c[i]);
  for (int i = 0; i < 15; i++) T.log_poly[i] = quantize_host(P, logc[i]);
  T.c_half = quantize_host(P, 0.5);
  T.c_ln2 = quantize_host(P, 0.693147180559945309417232121458176568);
  T.c_log2e = quantize_host(P, 1.44269504088896340735992468100189214);
  T.c_one_q = quantize_host(P, 1.0);
  // generate_sin_poly (fixed_point.rs:189-211: "lolremez -d 14 -r 0:pi sin(x)", highest power first) and the constants of qsin, qcos, qsinh
  static const double sinc[15] = {-1.1008071636607462e-11, 2.4208013888629323e-10, -3.8584805817996712e-10, -2.3786993104309845e-08,
                                  -2.9795813710683115e-09, 2.7608543130047009e-06, -6.4467066994122565e-09, -0.00019840680551418068,
                                  -3.839555844512214e-09, 0.0083333350601673614, -5.0943769725466814e-10, -0.16666666657583049,
                                  -8.5029878414113731e-12, 1.0000000000003146, -1.9323057584419828e-15};
  for (int i = 0; i < 15; i++) T.sin_poly[i] = quantize_host(P, sinc[i]);
  const double pi = 3.14159265358979323846264338327950288;
  T.c_pi = quantize_host(P, pi);
  T.c_two_pi = quantize_host(P, pi * 2.0);
  T.c_half_pi = quantize_host(P, 1.57079632679489661923132169163975144);
  T.c_two = quantize_host(P, 2.0);
  for (uint32_t i = 0; i < 260; i++) {
    T.small[i] = host_fr_from_u64(i);
    T.small_inv[i] = i ? mont_inv<Fr>(T.small[i]) : u256_zero();
  }
  compute_sizes(T);
  VDB_HIP(hipMalloc(&e->limb_tab, ((size_t)1 << L) * sizeof(u256)));
  {
    VDB_PROF("k_limb_table");
    hipLaunchKernelGGL(k_limb_table, dim3((unsigned)(((1u << L) + 255) / 256)), dim3(256), 0, ctx().stream, e->limb_tab, 1u << L);
  }
  VDB_LAUNCH_CHECK();
  T.limb_tab = e->limb_tab;
  VDB_HIP(hipMalloc(&e->dev, sizeof(FpTables)));
  VDB_HIP(hipMemcpyAsync(e->dev, &T, sizeof(FpTables), hipMemcpyHostToDevice, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  guard.keep = true;
  *out = e;
  return VDB_OK;
}

// ------------------------------------------------------------------ distance layout
enum { M_EUCLID = 0, M_COSINE = 1, M_MANHATTAN = 2, M_HAMMING = 3 };
struct DistLayout {
  uint32_t metric, D;
  uint64_t head_cells, head_lk, tail_cells, tail_lk, total_cells, total_lk;
};
static int dist_layout(const FpTables& T, int metric, size_t D, DistLayout* o) {
  const Sizes& z = T.sz;
  o->metric = (uint32_t)metric;
  o->D = (uint32_t)D;
  uint64_t ip_c = 4 + D * ((uint64_t)z.qmul[0] + 4), ip_l = D * (uint64_t)z.qmul[1];
  switch (metric) {
    case M_EUCLID:
      o->head_cells = 4 * D + ip_c;
      o->head_lk = ip_l;
      o->tail_cells = z.qsqrt[0];
      o->tail_lk = z.qsqrt[1];
      break;
    case M_COSINE:
      o->head_cells = 3 * ip_c;
      o->head_lk = 3 * ip_l;
      o->tail_cells = 2ull * z.qsqrt[0] + z.qmul[0] + z.qdiv[0] + 1 + 4;
      o->tail_lk = 2ull * z.qsqrt[1] + z.qmul[1] + z.qdiv[1];
      break;
    case M_MANHATTAN:
      o->head_cells = 4 * D + D * (uint64_t)z.qabs[0] + (D == 0 ? 0 : (D == 1 ? 1 : 1 + 3 * (D - 1)));
      o->head_lk = D * (uint64_t)z.qabs[1];
      o->tail_cells = 0;
      o->tail_lk = 0;
      break;
    case M_HAMMING:  // distance.rs:146-175: is_equal per element (sub + is_zero), gate().sum, two load_witness, qdiv, load_constant, qsub
      o->head_cells = 12 * D + (D == 0 ? 0 : (D == 1 ? 1 : 1 + 3 * (D - 1)));
      o->head_lk = 0;
      o->tail_cells = 2ull + z.qdiv[0] + 1 + 4;
      o->tail_lk = z.qdiv[1];
      break;
    default:
      set_error("unsupported metric %d (0 euclidean, 1 cosine, 2 manhattan, 3 hamming)", metric);
      return VDB_ERR_ARG;
  }
  o->total_cells = o->head_cells + o->tail_cells;
  o->total_lk = o->head_lk + o->tail_lk;
  return VDB_OK;
}

// where instance t lives in the streams and which operand vectors it uses
struct InstMap {
  uint64_t adv_base, lk_base;
  uint32_t grp;                       // instances per group
  uint64_t grp_adv_stride, grp_lk_stride;
  uint32_t a_mod, b_div;              // a = A[(t % a_mod)], b = B[(t / b_div)]
  __device__ __forceinline__ uint64_t adv(uint32_t t, const DistLayout& dl) const { return adv_base + (uint64_t)(t / grp) * grp_adv_stride + (uint64_t)(t % grp) * dl.total_cells; }
  __device__ __forceinline__ uint64_t lk(uint32_t t, const DistLayout& dl) const { return lk_base + (uint64_t)(t / grp) * grp_lk_stride + (uint64_t)(t % grp) * dl.total_lk; }
};

struct Streams {
  u256* adv;
  uint8_t* sel;
  u256* lk;
  int* err;
  // deferred-inversion list (see WCtx)
  uint64_t* inv_pos;
  u256* inv_val;
  uint32_t* inv_cnt;
  uint32_t inv_cap;
  // rank window in the coordinates of adv / lk (see WCtx); full range by default
  uint64_t rlo, rhi, rllo, rlhi;
  __device__ __forceinline__ bool touches(uint64_t a0, uint64_t a1, uint64_t l0, uint64_t l1) const {
    return (a0 < rhi && a1 > rlo) || (l0 < rlhi && l1 > rllo && l1 > l0);
  }
};

// (the streams, the rank window, the tables and the inversion list are not part of the context: they are the call's g_winv, set_winv)
__device__ __forceinline__ WCtx make_ctx(const Streams&, const FpTables*, uint64_t pos, uint64_t lpos) {
  WCtx c;
  c.pos = pos;
  c.lpos = lpos;
  c.lo = 0;
  c.hi = ~0ull;
  c.count_only = false;
  c.err = 0;
  return c;
}
// every host entry publishes its call's invariant context before it launches a kernel (stream ordered: kernels of an earlier call
// still read the earlier one)
static int set_winv(const Streams& st, const FpTables* T) {
  // (the source outlives the call: a pageable source is staged before hipMemcpyToSymbolAsync returns on this runtime, as the other
  //  small uploads of this file rely on too; the ring only makes that assumption harmless should a runtime ever defer the read)
  static thread_local WInv ring[8];
  static thread_local unsigned slot = 0;
  WInv& h = ring[slot++ & 7u];
  h = WInv{};
  h.adv = st.adv;
  h.sel = st.sel;
  h.lk = st.lk;
  h.rlo = st.rlo;
  h.rhi = st.rhi;
  h.rllo = st.rllo;
  h.rlhi = st.rlhi;
  h.T = T;
  h.inv_pos = st.inv_pos;
  h.inv_val = st.inv_val;
  h.inv_cnt = st.inv_cnt;
  h.inv_cap = st.inv_cap;
  VDB_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_winv), &h, sizeof(WInv), 0, hipMemcpyHostToDevice, ctx().stream));
  return VDB_OK;
}

#define HEAD_TB 128
// inner_product(a, b) of fixed_point.rs:854-874 for one instance, cooperatively by the block:
// thread i emits qmul(a_i, b_i) and the following qadd; running sums come from an LDS prefix.
// mode 0: operands x_i = A[i], y_i = B[i];  mode 1: x_i = y_i = A[i] - B[i] (already emitted qsub)
__device__ void block_inner_product(const Streams& st, const FpTables* T, const u256* A, const u256* Bv, int mode, uint32_t D,
                                    uint64_t ipbase, uint64_t iplbase, u256* sh /* HEAD_TB + 1 */, u256* result) {
  const uint32_t tid = threadIdx.x;
  const uint32_t qm = T->sz.qmul[0], qml = T->sz.qmul[1];
  if (tid == 0) {
    WCtx c = make_ctx(st, T, ipbase, iplbase);
    Gadgets g(c);
    g.g_add(u256_zero(), u256_zero());  // res = qadd(0, 0)
    sh[HEAD_TB] = u256_zero();
  }
  __syncthreads();
  for (uint32_t c0 = 0; c0 < D; c0 += HEAD_TB) {
    uint32_t i = c0 + tid;
    u256 q = u256_zero();
    WCtx c = make_ctx(st, T, ipbase + 4 + (uint64_t)i * (qm + 4), iplbase + (uint64_t)i * qml);
    Gadgets g(c);
    if (i < D) {
      u256 x = A[i], y = Bv[i];
      if (mode == 1) {
        x = fr_sub(x, y);
        y = x;
      }
      q = g.fp_qmul(x, y);
      if (c.err) atomicOr(st.err, c.err);
    }
    sh[tid] = q;
    __syncthreads();
    if (i < D) {
      u256 prev = sh[HEAD_TB];
      for (uint32_t j = 0; j < tid; j++) prev = fr_add(prev, sh[j]);
      g.g_add(prev, q);  // res = qadd(res, a_i b_i)
    }
    __syncthreads();
    if (tid == 0) {
      u256 carry = sh[HEAD_TB];
      uint32_t lim = D - c0 < HEAD_TB ? D - c0 : HEAD_TB;
      for (uint32_t j = 0; j < lim; j++) carry = fr_add(carry, sh[j]);
      sh[HEAD_TB] = carry;
    }
    __syncthreads();
  }
  if (tid == 0) *result = sh[HEAD_TB];
  __syncthreads();
}

// head of a distance: everything before the sequential tail.  One block per instance.
__global__ __launch_bounds__(HEAD_TB) void k_dist_head(Streams st, const FpTables* __restrict__ T, DistLayout dl, InstMap im,
                                                       const u256* __restrict__ A, const u256* __restrict__ Bv, u256* __restrict__ mid /* 3 per inst */,
                                                       u256* __restrict__ result, int have_values) {
  __shared__ u256 sh[HEAD_TB + 1];
  const uint32_t t = blockIdx.x, tid = threadIdx.x, D = dl.D;
  const u256* a = A + (size_t)(t % im.a_mod) * D;
  const u256* b = Bv + (size_t)(t / im.b_div) * D;
  const uint64_t base = im.adv(t, dl), lbase = im.lk(t, dl);
  // sharded run: k_dist_values has left this instance's sums in `mid` already, so a block none of whose cells lie in the rank's
  // window has nothing to do (without it every rank walked every instance's head in value-only mode)
  if (have_values && !st.touches(base, base + dl.head_cells, lbase, lbase + dl.head_lk)) return;
  if (dl.metric == M_EUCLID) {  // distance.rs:97-119
    for (uint32_t i = tid; i < D; i += HEAD_TB) {
      WCtx c = make_ctx(st, T, base + 4ull * i, lbase);
      Gadgets g(c);
      g.g_sub(a[i], b[i]);
    }
    block_inner_product(st, T, a, b, 1, D, base + 4ull * D, lbase, sh, &mid[3 * (size_t)t]);
  } else if (dl.metric == M_COSINE) {  // distance.rs:121-144
    const uint64_t ipc = 4 + (uint64_t)D * (T->sz.qmul[0] + 4), ipl = (uint64_t)D * T->sz.qmul[1];
    block_inner_product(st, T, a, b, 0, D, base, lbase, sh, &mid[3 * (size_t)t]);
    block_inner_product(st, T, a, a, 0, D, base + ipc, lbase + ipl, sh, &mid[3 * (size_t)t + 1]);
    block_inner_product(st, T, b, b, 0, D, base + 2 * ipc, lbase + 2 * ipl, sh, &mid[3 * (size_t)t + 2]);
  } else {  // manhattan, distance.rs:177-195; hamming, distance.rs:146-175: one value per element, then gate().sum over them
    const bool ham = dl.metric == M_HAMMING;
    const uint32_t qa = T->sz.qabs[0], qal = T->sz.qabs[1];
    const uint64_t sumbase = ham ? base + 12ull * D : base + 4ull * D + (uint64_t)D * qa;
    if (tid == 0) sh[HEAD_TB] = u256_zero();
    __syncthreads();
    for (uint32_t c0 = 0; c0 < D; c0 += HEAD_TB) {
      uint32_t i = c0 + tid;
      u256 v = u256_zero();
      if (i < D) {
        if (ham) {  // gate().is_equal(a_i, b_i): [a - b, b, 1, a] then is_zero's eight cells
          WCtx c = make_ctx(st, T, base + 12ull * i, lbase);
          Gadgets g(c);
          v = g.g_is_equal(a[i], b[i]);
        } else {
          WCtx c = make_ctx(st, T, base + 4ull * i, lbase);
          Gadgets g(c);
          u256 d = g.g_sub(a[i], b[i]);
          c.pos = base + 4ull * D + (uint64_t)i * qa;
          c.lpos = lbase + (uint64_t)i * qal;
          v = g.fp_qabs(d);
        }
      }
      sh[tid] = v;
      __syncthreads();
      if (i < D) {  // gate().sum: [a0, a1, 1, s1, a2, 1, s2, ...]
        WCtx c = make_ctx(st, T, 0, 0);
        if (i == 0) {
          c.pos = sumbase;
          c.push(v, D > 1);
        } else {
          u256 s = sh[HEAD_TB];
          for (uint32_t j = 0; j <= tid; j++) s = fr_add(s, sh[j]);
          c.pos = sumbase + 1 + 3ull * (i - 1);
          c.push(v, false);
          c.push(mont_one<Fr>(), false, true);
          c.push(s, i + 1 < D);
        }
      }
      __syncthreads();
      if (tid == 0) {
        u256 carry = sh[HEAD_TB];
        uint32_t lim = D - c0 < HEAD_TB ? D - c0 : HEAD_TB;
        for (uint32_t j = 0; j < lim; j++) carry = fr_add(carry, sh[j]);
        sh[HEAD_TB] = carry;
      }
      __syncthreads();
    }
    if (tid == 0) {
      if (ham)
        mid[3 * (size_t)t] = sh[HEAD_TB];   // the number of equal elements: the tail quantizes it
      else
        result[t] = sh[HEAD_TB];
    }
  }
}

// sequential tail of a distance, cut into `gridDim.y` position windows; lanes = instances
__global__ __launch_bounds__(64) void k_dist_tail(Streams st, const FpTables* __restrict__ T, DistLayout dl, InstMap im, uint32_t n_inst,
                                                  const u256* __restrict__ mid, u256* __restrict__ result, int have_values) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const bool live = t < n_inst;
  if (!live) t = n_inst - 1;
  const uint32_t S = gridDim.y, s = blockIdx.y;
  const uint64_t tb = im.adv(t, dl) + dl.head_cells, tlb = im.lk(t, dl) + dl.head_lk;
  // segment S-1 carries the result every rank needs; the other segments only emit cells and leave at once
  // when none of the wavefront's instances lies in this rank's window
  // (`have_values`: k_dist_tail_values has computed every result already — the last segment leaves like the others)
  if ((s != S - 1 || have_values) && !__any((int)(live && st.touches(tb, tb + dl.tail_cells, tlb, tlb + dl.tail_lk)))) return;
  WCtx c = make_ctx(st, T, tb, tlb);
  c.lo = tb + dl.tail_cells * s / S;
  c.hi = tb + dl.tail_cells * (s + 1) / S;
  if (!live) c.lo = c.hi = tb;  // padding lanes follow the same control flow but store nothing
  Gadgets g(c);
  u256 r;
  if (dl.metric == M_EUCLID) {
    r = g.fp_qsqrt(mid[3 * (size_t)t]);
  } else if (dl.metric == M_HAMMING) {
    // len = load_witness(quantization(D)), ab_sum_q = load_witness(quantization(the count as f64)): both exact (an integer times 2^P),
    // neither constrained by the reference (distance.rs:165-169); 1 - ab_sum_q / len
    const u256 len = fr_mul(to_mont<Fr>(u256_from_u64(dl.D)), T->scale);
    const u256 sq = fr_mul(mid[3 * (size_t)t], T->scale);
    c.push(len, false);
    c.push(sq, false);
    u256 sim = g.fp_qdiv(sq, len);
    u256 one = g.load_constant(T->c_one_q);
    r = g.g_sub(one, sim);
  } else {
    u256 ab = mid[3 * (size_t)t], aa = mid[3 * (size_t)t + 1], bb = mid[3 * (size_t)t + 2];
    u256 as = g.fp_qsqrt(aa);
    u256 bs = g.fp_qsqrt(bb);
    u256 den = g.fp_qmul(as, bs);
    u256 sim = g.fp_qdiv(ab, den);
    u256 one = g.load_constant(T->c_one_q);
    r = g.g_sub(one, sim);
  }
  if (live && s == S - 1) {
    result[t] = r;
    if (c.err) atomicOr(st.err, c.err);
  }
}

// ---- the distances' VALUES alone, for sharded runs (every rank needs every distance — assignments and centroids follow
// from them — but stores only the cells of its own columns).  One wavefront per instance: lane i takes dimensions i, i + 64, ...
// through the gadgets' value-only helpers, the sums are folded across the wavefront (field addition: the order is free), lane 0
// writes what the head would have left in `mid` (Manhattan: the result itself).  No emission context, no stores of cells.
__device__ __forceinline__ u256 wave_sum_fr(u256 v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    u256 o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.w[k] = __shfl_xor(v.w[k], off, 64);
    v = fr_add(v, o);
  }
  return v;
}
__global__ __launch_bounds__(64) void k_dist_values(const FpTables* __restrict__ T, DistLayout dl, InstMap im, const u256* __restrict__ A,
                                                    const u256* __restrict__ Bv, u256* __restrict__ mid, u256* __restrict__ result) {
  const uint32_t t = blockIdx.x, lane = threadIdx.x, D = dl.D;
  const u256* a = A + (size_t)(t % im.a_mod) * D;
  const u256* b = Bv + (size_t)(t / im.b_div) * D;
  WCtx c{};
  Gadgets g(c);
  u256 s0 = u256_zero(), s1 = u256_zero(), s2 = u256_zero();
  for (uint32_t i = lane; i < D; i += 64) {
    const u256 x = a[i], y = b[i];
    if (dl.metric == M_EUCLID) {
      const u256 d = fr_sub(x, y);
      s0 = fr_add(s0, g.v_qmul(d, d));
    } else if (dl.metric == M_COSINE) {
      s0 = fr_add(s0, g.v_qmul(x, y));
      s1 = fr_add(s1, g.v_qmul(x, x));
      s2 = fr_add(s2, g.v_qmul(y, y));
    } else if (dl.metric == M_HAMMING) {
      if (u256_eq(x, y)) s0 = fr_add(s0, mont_one<Fr>());
    } else {
      s0 = fr_add(s0, g.v_qabs(fr_sub(x, y)));
    }
  }
  s0 = wave_sum_fr(s0);
  if (dl.metric == M_COSINE) {
    s1 = wave_sum_fr(s1);
    s2 = wave_sum_fr(s2);
  }
  if (lane == 0) {
    if (dl.metric == M_MANHATTAN) {
      result[t] = s0;
    } else {
      mid[3 * (size_t)t] = s0;
      if (dl.metric == M_COSINE) {
        mid[3 * (size_t)t + 1] = s1;
        mid[3 * (size_t)t + 2] = s2;
      }
    }
  }
}
// ... and the sequential tails' values: lanes = instances, the tail's own code with an emission window that holds no position
// (every sub-gadget takes its value-only path; domain errors — a division by zero — are reported as the emitting walk reports them)
__global__ __launch_bounds__(64) void k_dist_tail_values(Streams st, const FpTables* __restrict__ T, DistLayout dl, uint32_t n_inst,
                                                         const u256* __restrict__ mid, u256* __restrict__ result) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  const bool live = t < n_inst;
  if (!live) t = n_inst - 1;
  WCtx c = make_ctx(st, T, 1, 0);
  c.lo = c.hi = 0;
  Gadgets g(c);
  u256 r;
  if (dl.metric == M_EUCLID) {
    r = g.fp_qsqrt(mid[3 * (size_t)t]);
  } else if (dl.metric == M_HAMMING) {
    // len = load_witness(quantization(D)), ab_sum_q = load_witness(quantization(the count as f64)): both exact (an integer times 2^P),
    // neither constrained by the reference (distance.rs:165-169); 1 - ab_sum_q / len
    const u256 len = fr_mul(to_mont<Fr>(u256_from_u64(dl.D)), T->scale);
    const u256 sq = fr_mul(mid[3 * (size_t)t], T->scale);
    c.push(len, false);
    c.push(sq, false);
    u256 sim = g.fp_qdiv(sq, len);
    u256 one = g.load_constant(T->c_one_q);
    r = g.g_sub(one, sim);
  } else {
    u256 ab = mid[3 * (size_t)t], aa = mid[3 * (size_t)t + 1], bb = mid[3 * (size_t)t + 2];
    u256 as = g.fp_qsqrt(aa);
    u256 bs = g.fp_qsqrt(bb);
    u256 den = g.fp_qmul(as, bs);
    u256 sim = g.fp_qdiv(ab, den);
    u256 one = g.load_constant(T->c_one_q);
    r = g.g_sub(one, sim);
  }
  if (live) {
    result[t] = r;
    if (c.err) atomicOr(st.err, c.err);
  }
}

// Batched evaluation of the deferred inverse cells: Montgomery's trick over runs of 16 entries
#define INVFIX_CH 16
__global__ __launch_bounds__(64) void k_inv_fixup(u256* __restrict__ adv, const uint64_t* __restrict__ pos, const u256* __restrict__ val,
                                                  const uint32_t* __restrict__ cnt, uint32_t cap) {
  const uint32_t n = *cnt < cap ? *cnt : cap;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; (uint64_t)t * INVFIX_CH < n; t += stride) {
    const uint32_t lo = t * INVFIX_CH, hi = lo + INVFIX_CH < n ? lo + INVFIX_CH : n;
    u256 pre[INVFIX_CH];
    u256 acc = mont_one<Fr>();
    for (uint32_t i = lo; i < hi; i++) {
      pre[i - lo] = acc;
      acc = fr_mul(acc, val[i]);  // entries are non-zero by construction
    }
    acc = mont_inv<Fr>(acc);
    for (uint32_t i = hi; i-- > lo;) {
      adv[pos[i]] = fr_mul(acc, pre[i - lo]);
      acc = fr_mul(acc, val[i]);
    }
  }
}
// attaches the context-owned deferred-inversion list to `st` and resets its counter
static int inv_list_attach(Streams& st, uint64_t cells) {
  uint64_t cap = cells / 16 + 4096;
  if (cap > (16u << 20)) cap = 16u << 20;
  uint8_t* buf = (uint8_t*)scratch_get(4, cap * (sizeof(u256) + sizeof(uint64_t)) + 64);
  if (!buf) return VDB_ERR_OOM;
  st.inv_cnt = (uint32_t*)buf;
  st.inv_val = (u256*)(buf + 64);
  st.inv_pos = (uint64_t*)(buf + 64 + cap * sizeof(u256));
  st.inv_cap = (uint32_t)cap;
  VDB_HIP(hipMemsetAsync(st.inv_cnt, 0, sizeof(uint32_t), ctx().stream));
  return VDB_OK;
}
static int inv_list_fixup(const Streams& st) {
  if (!st.inv_cnt) return VDB_OK;
  {
    VDB_PROF("k_inv_fixup");
    hipLaunchKernelGGL(k_inv_fixup, dim3((unsigned)(ctx().cu_count * 8)), dim3(64), 0, ctx().stream, st.adv, st.inv_pos, st.inv_val, st.inv_cnt, st.inv_cap);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

static uint32_t tail_segments(uint32_t n_inst) {
  uint32_t groups = (n_inst + 63) / 64;
  uint32_t s = 4096 / (groups ? groups : 1);
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  if (const char* e = getenv("VDB_TAIL_SEGMENTS")) {
    int v = atoi(e);
    if (v >= 1 && v <= 1024) s = (uint32_t)v;
  }
  return s;
}

// runs head + tail for n_inst instances; distances land in `result` (device)
static int run_distances(const Streams& st, FpEntry* fp, const DistLayout& dl, const InstMap& im, uint32_t n_inst, const u256* A,
                         const u256* Bv, u256* mid, u256* result) {
  if (n_inst == 0) return VDB_OK;
  // a rank that stores only a window of the streams first computes every distance's value with the value-only kernels; the emitting
  // kernels then leave at once wherever none of their cells lie in the window (VDB_WIT_VALUES=0: the walk of rounds 1-3, every rank
  // running every head and the last tail segment in value-only mode)
  static const bool values_on = !(getenv("VDB_WIT_VALUES") && getenv("VDB_WIT_VALUES")[0] == '0');
  const bool windowed = !(st.rlo == 0 && st.rhi == ~0ull && st.rllo == 0 && st.rlhi == ~0ull);
  const int have_values = (windowed && values_on && st.sel == nullptr) ? 1 : 0;
  if (have_values) {
    {
      VDB_PROF("k_dist_values");
      hipLaunchKernelGGL(k_dist_values, dim3(n_inst), dim3(64), 0, ctx().stream, fp->dev, dl, im, A, Bv, mid, result);
    }
    VDB_LAUNCH_CHECK();
    if (dl.tail_cells) {
      {
        VDB_PROF("k_dist_tail_values");
        hipLaunchKernelGGL(k_dist_tail_values, dim3((n_inst + 63) / 64), dim3(64), 0, ctx().stream, st, fp->dev, dl, n_inst, mid, result);
      }
      VDB_LAUNCH_CHECK();
    }
  }
  {
    VDB_PROF("k_dist_head");
    hipLaunchKernelGGL(k_dist_head, dim3(n_inst), dim3(HEAD_TB), 0, ctx().stream, st, fp->dev, dl, im, A, Bv, mid, result, have_values);
  }
  VDB_LAUNCH_CHECK();
  if (dl.tail_cells) {
    {
      VDB_PROF("k_dist_tail");
      hipLaunchKernelGGL(k_dist_tail, dim3((n_inst + 63) / 64, tail_segments(n_inst)), dim3(64), 0, ctx().stream, st, fp->dev, dl, im, n_inst,
                       mid, result, have_values);
    }
    VDB_LAUNCH_CHECK();
  }
  return VDB_OK;
}

// ------------------------------------------------------------------ nearest_vector (vectordb.rs:122-163)
__global__ void k_nv_prefix_min(const FpTables* __restrict__ T, const u256* __restrict__ d, uint32_t n, u256* __restrict__ pm) {
  if (blockIdx.x || threadIdx.x) return;
  WCtx c{};
  Gadgets g(c);
  u256 m = d[0];
  pm[0] = m;
  for (uint32_t i = 1; i < n; i++) {
    u256 x = d[i];
    if (!g.v_is_neg(from_mont<Fr>(fr_sub(m, x)))) m = x;  // qmin(m, x) = is_neg(m - x) ? m : x
    pm[i] = m;
  }
}
__global__ __launch_bounds__(64) void k_nv_qmin(Streams st, const FpTables* __restrict__ T, uint64_t base, uint64_t lbase, const u256* __restrict__ d,
                                                const u256* __restrict__ pm, uint32_t n) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x + 1;
  if (i >= n) return;
  WCtx c = make_ctx(st, T, base + (uint64_t)(i - 1) * T->sz.qmin[0], lbase + (uint64_t)(i - 1) * T->sz.qmin[1]);
  Gadgets g(c);
  g.fp_qmin(pm[i - 1], d[i]);
}
__global__ __launch_bounds__(64) void k_nv_is_equal(Streams st, const FpTables* __restrict__ T, uint64_t base, const u256* __restrict__ d,
                                                    const u256* __restrict__ pm, uint32_t n, u256* __restrict__ ind) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  WCtx c = make_ctx(st, T, base + 12ull * i, 0);
  Gadgets g(c);
  ind[i] = g.g_is_equal(pm[n - 1], d[i]);
}
// select_by_indicator per dimension: [0, a0, ind0, s0, a1, ind1, s1, ...]
__global__ __launch_bounds__(64) void k_nv_select(Streams st, const FpTables* __restrict__ T, uint64_t base, const u256* __restrict__ vectors,
                                                  const u256* __restrict__ ind, uint32_t n, uint32_t D, u256* __restrict__ result) {
  uint32_t j = blockIdx.x * 64 + threadIdx.x;
  if (j >= D) return;
  WCtx c = make_ctx(st, T, base + (uint64_t)j * (1 + 3ull * n), 0);
  u256 s = u256_zero();
  c.push(s, n > 0);
  for (uint32_t i = 0; i < n; i++) {
    u256 a = vectors[(size_t)i * D + j], in = ind[i];
    if (!u256_is_zero(in)) s = a;
    c.push(a, false);
    c.push(in, false);
    c.push(s, i + 1 < n);
  }
  result[j] = s;
}

// ------------------------------------------------------------------ kmeans (vectordb.rs:225-362)
struct KmLayout {
  uint32_t N, D, K;
  uint64_t per_vec, per_vec_l, assign, assign_l, sizes, per_cluster, per_cluster_l, iter, iter_l;
};
__global__ __launch_bounds__(64) void k_km_assign(Streams st, const FpTables* __restrict__ T, KmLayout kl, DistLayout dl, uint64_t ibase, uint64_t ilbase,
                                                  const u256* __restrict__ dist, u256* __restrict__ ind) {
  // lanes = vectors, blockIdx.y = position window of the per-vector assignment block
  uint32_t v = blockIdx.x * 64 + threadIdx.x;
  const bool live = v < kl.N;
  if (!live) v = kl.N - 1;
  const uint32_t K = kl.K, S = gridDim.y, s = blockIdx.y;
  const uint64_t base = ibase + (uint64_t)v * kl.per_vec + (uint64_t)K * dl.total_cells;
  const uint64_t cells = (uint64_t)(K - 1) * T->sz.qmin[0] + 20ull * K;
  const uint64_t lbase_v = ilbase + (uint64_t)v * kl.per_vec_l + (uint64_t)K * dl.total_lk;
  if (s != S - 1 && !__any((int)(live && st.touches(base, base + cells, lbase_v, lbase_v + (uint64_t)(K - 1) * T->sz.qmin[1])))) return;
  WCtx c = make_ctx(st, T, base, lbase_v);
  c.lo = base + cells * s / S;
  c.hi = base + cells * (s + 1) / S;
  if (!live) c.lo = c.hi = base;
  Gadgets g(c);
  const u256* d = dist + (size_t)v * K;
  u256 m = d[0];
  for (uint32_t k = 1; k < K; k++) m = g.fp_qmin(m, d[k]);
  for (uint32_t k = 0; k < K; k++) {
    u256 eq = g.g_is_equal(m, d[k]);
    u256 r = g.g_select(T->c_one_q, u256_zero(), eq);
    if (live && s == S - 1) ind[(size_t)v * K + k] = r;
  }
}
#define KM_PF 8
__device__ __forceinline__ u256 shfl_up_u256(const u256& v, int delta) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = (uint32_t)__shfl_up((int)v.w[i], delta, 64);
  return r;
}
__device__ __forceinline__ u256 shfl_u256w(const u256& v, int src) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.w[i] = (uint32_t)__shfl((int)v.w[i], src, 64);
  return r;
}
// One wavefront folds one chain s_v = s_{v-1} + x_v (s_0 = x_0) and emits the qadd cells of steps v = 1 .. N-1 at
// base + (v - 1) * stride: ORDER 0 = [s_{v-1}, x_v, 1, s_v], ORDER 1 = [x_v, s_{v-1}, 1, s_v].  The chain is
// sequential in the reference; here lane l owns the contiguous steps [l * per, (l + 1) * per): lane totals (loads
// only), a wavefront scan, then every lane emits its steps from its own prefix.  Loads always run ahead of the stores
// of a chunk (a load issued after stores waits for their acknowledgement), and lanes whose cells lie outside the
// rank window skip the emission altogether.  Returns the chain total (all lanes).
template <int ORDER, class LoadFn>
__device__ __forceinline__ u256 chain_fold_wave(WCtx& c, uint64_t base, uint64_t stride, uint32_t N, LoadFn&& x) {
  const uint32_t lane = threadIdx.x & 63, per = (N + 63) / 64;
  const uint32_t lo = lane * per < N ? lane * per : N, hi = lo + per < N ? lo + per : N;
  u256 tot = u256_zero();
  for (uint32_t v0 = lo; v0 < hi; v0 += KM_PF) {
    u256 xv[KM_PF];
#pragma unroll
    for (uint32_t q = 0; q < KM_PF; q++) xv[q] = x(v0 + q < hi ? v0 + q : hi - 1);
#pragma unroll
    for (uint32_t q = 0; q < KM_PF; q++)
      if (v0 + q < hi) tot = fr_add(tot, xv[q]);
  }
  u256 inc = tot;  // inclusive scan over lanes
  for (int o = 1; o < 64; o <<= 1) {
    u256 t = shfl_up_u256(inc, o);
    if ((int)lane >= o) inc = fr_add(inc, t);
  }
  const u256 total = shfl_u256w(inc, 63);
  u256 run = fr_sub(inc, tot);  // exclusive prefix = s_{lo-1}
  // the lane's cells: steps max(lo, 1) .. hi-1
  const uint32_t first = lo ? lo : 1;
  if (first >= hi) return total;
  const uint64_t c_lo = base + (uint64_t)(first - 1) * stride, c_hi = base + (uint64_t)(hi - 2) * stride + 4;
  if (!c.count_only && (c_hi <= winv().rlo || c_lo >= winv().rhi)) return total;
  for (uint32_t v0 = lo; v0 < hi; v0 += KM_PF) {
    u256 xv[KM_PF];
#pragma unroll
    for (uint32_t q = 0; q < KM_PF; q++) xv[q] = x(v0 + q < hi ? v0 + q : hi - 1);
#pragma unroll
    for (uint32_t q = 0; q < KM_PF; q++) {
      const uint32_t v = v0 + q;
      if (v < hi) {
        const u256 nx = fr_add(run, xv[q]);
        if (v >= 1) {
          c.pos = base + (uint64_t)(v - 1) * stride;
          c.push(ORDER == 0 ? run : xv[q], true);
          c.push(ORDER == 0 ? xv[q] : run, false);
          c.push(mont_one<Fr>(), false, true);
          c.push(nx, false);
        }
        run = nx;
      }
    }
  }
  return total;
}
// cluster sizes: sizes_k = sum_v indicator[v][k] as a chain of qadd cells (vectordb.rs:316-322); one wavefront per cluster
__global__ __launch_bounds__(64) void k_km_sizes(Streams st, const FpTables* __restrict__ T, KmLayout kl, uint64_t base, const u256* __restrict__ ind,
                                                 u256* __restrict__ sizes) {
  const uint32_t k = blockIdx.x;
  WCtx c = make_ctx(st, T, 0, 0);
  const u256 tot = chain_fold_wave<0>(c, base + 4ull * k, 4ull * kl.K, kl.N, [&](uint32_t v) { return ind[(size_t)v * kl.K + k]; });
  if (threadIdx.x == 0) sizes[k] = tot;
}
// filtered vectors: select(0, vector_j, is_zero(indicator)) per (cluster, vector, dimension); a thread owns KM_PF dimensions
__global__ __launch_bounds__(64) void k_km_filter(Streams st, const FpTables* __restrict__ T, KmLayout kl, uint64_t cbase0, const u256* __restrict__ vectors,
                                                  const u256* __restrict__ ind, u256 scale_inv, u256* __restrict__ filt) {
  const uint32_t JC = (kl.D + KM_PF - 1) / KM_PF;
  const uint64_t id = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (id >= (uint64_t)kl.K * kl.N * JC) return;
  const uint32_t jc = (uint32_t)(id % JC), v = (uint32_t)((id / JC) % kl.N), k = (uint32_t)(id / ((uint64_t)JC * kl.N));
  const uint32_t j0 = jc * KM_PF;
  const u256 sel = ind[(size_t)v * kl.K + k];
  u256 x[KM_PF];
#pragma unroll
  for (uint32_t q = 0; q < KM_PF; q++) x[q] = vectors[(size_t)v * kl.D + (j0 + q < kl.D ? j0 + q : kl.D - 1)];
  const uint64_t cb = cbase0 + (uint64_t)k * kl.per_cluster + (uint64_t)v * (8 + 8ull * kl.D);
  WCtx c = make_ctx(st, T, cb, 0);
  Gadgets g(c);
  u256 iz;
  if (jc == 0) iz = g.g_is_zero_inv(sel, u256_is_zero(sel) ? mont_one<Fr>() : scale_inv);
  else iz = u256_is_zero(sel) ? mont_one<Fr>() : u256_zero();
  c.pos = cb + 8 + 8ull * j0;
#pragma unroll
  for (uint32_t q = 0; q < KM_PF; q++)
    if (j0 + q < kl.D) filt[((size_t)k * kl.N + v) * kl.D + j0 + q] = g.g_select(u256_zero(), x[q], iz);
}
// per-cluster, per-dimension sums of the filtered vectors (vectordb.rs:338-347): one wavefront per (cluster, dimension)
__global__ __launch_bounds__(64) void k_km_sum(Streams st, const FpTables* __restrict__ T, KmLayout kl, uint64_t cbase0, const u256* __restrict__ filt,
                                               u256* __restrict__ sums) {
  const uint32_t id = blockIdx.x, k = id / kl.D, j = id % kl.D;
  WCtx c = make_ctx(st, T, 0, 0);
  const uint64_t base = cbase0 + (uint64_t)k * kl.per_cluster + (uint64_t)kl.N * (8 + 8ull * kl.D) + 4ull * j;
  const u256 tot = chain_fold_wave<1>(c, base, 4ull * kl.D, kl.N, [&](uint32_t v) { return filt[((size_t)k * kl.N + v) * kl.D + j]; });
  if (threadIdx.x == 0) sums[id] = tot;
}
__global__ __launch_bounds__(64) void k_km_div(Streams st, const FpTables* __restrict__ T, KmLayout kl, uint64_t cbase0, uint64_t clbase0,
                                               const u256* __restrict__ sums, const u256* __restrict__ sizes, u256* __restrict__ cent) {
  uint32_t id = blockIdx.x * 64 + threadIdx.x;
  const bool live = id < kl.K * kl.D;
  if (!live) id = kl.K * kl.D - 1;
  uint32_t k = id / kl.D, j = id % kl.D;
  const uint32_t S = gridDim.y, s = blockIdx.y;
  const uint64_t base = cbase0 + (uint64_t)k * kl.per_cluster + (uint64_t)kl.N * (8 + 8ull * kl.D) + (uint64_t)(kl.N - 1) * kl.D * 4 + (uint64_t)j * T->sz.qdiv[0];
  const uint64_t lbase = clbase0 + (uint64_t)k * kl.per_cluster_l + (uint64_t)j * T->sz.qdiv[1];
  if (s != S - 1 && !__any((int)(live && st.touches(base, base + T->sz.qdiv[0], lbase, lbase + T->sz.qdiv[1])))) return;
  WCtx c = make_ctx(st, T, base, lbase);
  c.lo = base + (uint64_t)T->sz.qdiv[0] * s / S;
  c.hi = base + (uint64_t)T->sz.qdiv[0] * (s + 1) / S;
  if (!live) c.lo = c.hi = base;
  Gadgets g(c);
  u256 r = g.fp_qdiv(sums[id], sizes[k]);
  if (live && s == S - 1) {
    cent[id] = r;
    if (c.err) atomicOr(st.err, c.err);
  }
}
__global__ void k_push_cells(Streams st, uint64_t pos, u256 a, u256 b, uint32_t n) {
  if (blockIdx.x || threadIdx.x) return;
  // load_constant / load_zero cells: data-independent; stored by the rank whose window holds them, like every other cell
  if (pos >= st.rlo && pos < st.rhi) {
    st.adv[pos] = a;
    if (st.sel) st.sel[pos] = 2;
  }
  if (n > 1 && pos + 1 >= st.rlo && pos + 1 < st.rhi) {
    st.adv[pos + 1] = b;
    if (st.sel) st.sel[pos + 1] = 2;
  }
}

// ------------------------------------------------------------------ Poseidon trace (merkle_commitment)
// cells of PoseidonChip::permutation (halo2-lib community-edition poseidon chip, [UPSTREAM-RECALL])
// GateChip::sum; cmask bit i set = v[i] is a Constant cell
__device__ u256 trace_sum(WCtx& c, const FpTables* T, const u256* v, int n, unsigned cmask) {
  u256 s = v[0];
  c.push(v[0], n > 1, cmask & 1u);
  for (int i = 1; i < n; i++) {
    s = fr_add(s, v[i]);
    c.push(v[i], false, (cmask >> i) & 1u);
    c.push(mont_one<Fr>(), false, true);
    c.push(s, i + 1 < n);
  }
  return s;
}
__device__ u256 trace_ip_const(WCtx& c, const FpTables* T, const u256* a, const u256* row, int n) {  // inner_product(a, constants)
  u256 s;
  int i0, ng;
  if (u256_eq(row[0], mont_one<Fr>())) {
    s = a[0];
    i0 = 1;
    ng = n - 1;
    c.push(a[0], ng > 0);
  } else {
    s = u256_zero();
    i0 = 0;
    ng = n;
    c.push(u256_zero(), ng > 0, true);
  }
  int gi = 1;
  for (int i = i0; i < n; i++, gi++) {
    s = fr_add(s, fr_mul(a[i], row[i]));
    c.push(a[i], false);
    c.push(row[i], false, true);  // Constant(matrix entry)
    c.push(s, gi < ng);
  }
  return s;
}
__device__ void trace_sbox(Gadgets& g, u256& x, const u256& cst) {
  u256 x2 = g.g_mul(x, x);
  u256 x4 = g.g_mul(x2, x2);
  u256 o = fr_add(fr_mul(x, x4), cst);  // mul_add(x, x4, Constant(c)): [c, x, x4, out]
  g.c.push(cst, true, true); g.c.push(x, false); g.c.push(x4, false); g.c.push(o, false);
  x = o;
}
__device__ void trace_dense(WCtx& c, const FpTables* T, u256 st[PSD_T], const u256 m[PSD_T][PSD_T]) {
  u256 r[PSD_T];
  for (int i = 0; i < PSD_T; i++) r[i] = trace_ip_const(c, T, st, m[i], PSD_T);
  for (int i = 0; i < PSD_T; i++) st[i] = r[i];
}
// (the context by value and back: ten dwords in registers — by reference it lived in the caller's scratch and was re-read around
//  every cell store, which on gfx9 waits for the stores before it)
__device__ __noinline__ WCtx trace_permutation(WCtx c, const FpTables* T, const PoseidonSpec* __restrict__ sp, u256 st[PSD_T], const u256* in, int n_in) {
  Gadgets g(c);
  {
    u256 v[2] = {st[0], sp->start[0][0]};
    st[0] = trace_sum(c, T, v, 2, 2u);
  }
  for (int i = 0; i < n_in; i++) {
    u256 v[3] = {st[1 + i], in[i], sp->start[0][1 + i]};
    st[1 + i] = trace_sum(c, T, v, 3, 4u);
  }
  for (int i = n_in + 1, k = 0; i < PSD_T; i++, k++) {
    u256 cst = sp->start[0][i];
    if (k == 0) cst = fr_add(cst, mont_one<Fr>());
    u256 v[2] = {st[i], cst};
    st[i] = trace_sum(c, T, v, 2, 2u);
  }
  for (int r = 1; r < PSD_HALF; r++) {
    for (int i = 0; i < PSD_T; i++) trace_sbox(g, st[i], sp->start[r][i]);
    trace_dense(c, T, st, sp->mds);
  }
  for (int i = 0; i < PSD_T; i++) trace_sbox(g, st[i], sp->start[PSD_HALF][i]);
  trace_dense(c, T, st, sp->pre_sparse);
  for (int p = 0; p < PSD_RP; p++) {
    trace_sbox(g, st[0], sp->partial[p]);
    u256 r[PSD_T];
    r[0] = trace_ip_const(c, T, st, sp->sparse_row[p], PSD_T);
    for (int i = 1; i < PSD_T; i++) {  // mul_add(s0, Constant(e), s_i): [s_i, s0, e, out]
      r[i] = fr_add(fr_mul(st[0], sp->sparse_col[p][i - 1]), st[i]);
      c.push(st[i], true); c.push(st[0], false); c.push(sp->sparse_col[p][i - 1], false, true); c.push(r[i], false);
    }
    for (int i = 0; i < PSD_T; i++) st[i] = r[i];
  }
  for (int r = 0; r < PSD_HALF - 1; r++) {
    for (int i = 0; i < PSD_T; i++) trace_sbox(g, st[i], sp->end[r][i]);
    trace_dense(c, T, st, sp->mds);
  }
  u256 z = u256_zero();
  for (int i = 0; i < PSD_T; i++) trace_sbox(g, st[i], z);
  trace_dense(c, T, st, sp->mds);
  return c;
}
HD uint32_t perm_cells(int n_in) { return (n_in == 2 ? 18u : (n_in == 1 ? 15u : 12u)) + 2238u; }

// sponge states before every permutation of every leaf (value only)
__global__ __launch_bounds__(64) void k_mk_leaf_states(const PoseidonSpec* __restrict__ sp, const u256* __restrict__ vectors, uint32_t n, uint32_t D,
                                                       uint32_t nperm, u256* __restrict__ states /* n * nperm * 3 */, u256* __restrict__ leaves) {
  uint32_t v = blockIdx.x * 64 + threadIdx.x;
  if (v >= n) return;
  u256 st[PSD_T] = {sp->cap, u256_zero(), u256_zero()};
  const u256* msg = vectors + (size_t)v * D;
  uint32_t off = 0;
  for (uint32_t p = 0; p < nperm; p++) {
    for (int i = 0; i < PSD_T; i++) states[((size_t)v * nperm + p) * PSD_T + i] = st[i];
    int cnt = off < D ? (int)(D - off < 2 ? D - off : 2) : 0;
    u256 in[PSD_RATE] = {cnt > 0 ? msg[off] : u256_zero(), cnt > 1 ? msg[off + 1] : u256_zero()};
    psd_permute_absorb(sp, st, in, cnt);
    off += (uint32_t)cnt;
  }
  leaves[v] = st[1];
}
// one thread per leaf permutation: the trace cells
__global__ __launch_bounds__(64) void k_mk_leaf_trace(Streams stq, const FpTables* __restrict__ T, const PoseidonSpec* __restrict__ sp,
                                                      const u256* __restrict__ vectors, uint32_t n, uint32_t D, uint32_t nperm, uint64_t base,
                                                      uint64_t leaf_cells, const u256* __restrict__ states) {
  uint32_t id = blockIdx.x * 64 + threadIdx.x;
  if (id >= n * nperm) return;
  uint32_t v = id / nperm, p = id % nperm;
  uint32_t off = 2 * p;
  int cnt = off < D ? (int)(D - off < 2 ? D - off : 2) : 0;
  // permutations 0..p-1 of a leaf are full (2 inputs) except possibly the one before the padding-only one
  uint64_t pos = base + (uint64_t)v * leaf_cells;
  for (uint32_t q = 0; q < p; q++) {
    uint32_t o = 2 * q;
    pos += perm_cells(o < D ? (int)(D - o < 2 ? D - o : 2) : 0);
  }
  // a rank that holds a block of columns emits only the permutations whose cells fall into its stretch of the stream
  // (the sponge states they start from were computed by k_mk_leaf_states)
  if (!stq.touches(pos, pos + perm_cells(cnt), 0, 0)) return;
  WCtx c = make_ctx(stq, T, pos, 0);
  u256 st[PSD_T];
  for (int i = 0; i < PSD_T; i++) st[i] = states[((size_t)v * nperm + p) * PSD_T + i];
  const u256* msg = vectors + (size_t)v * D;
  u256 in[PSD_RATE] = {cnt > 0 ? msg[off] : u256_zero(), cnt > 1 ? msg[off + 1] : u256_zero()};
  c = trace_permutation(c, T, sp, st, in, cnt);
}
// one thread per tree node: two permutations (absorb [l, r], then padding only)
__global__ __launch_bounds__(64) void k_mk_node(Streams stq, const FpTables* __restrict__ T, const PoseidonSpec* __restrict__ sp, const u256* __restrict__ in_lv,
                                                uint32_t n_out, uint64_t base, u256* __restrict__ out_lv) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n_out) return;
  const uint64_t p0 = base + (uint64_t)t * (perm_cells(2) + perm_cells(0));
  u256 st[PSD_T] = {sp->cap, u256_zero(), u256_zero()};
  u256 in[PSD_RATE] = {in_lv[2 * t], in_lv[2 * t + 1]};
  if (!stq.touches(p0, p0 + perm_cells(2) + perm_cells(0), 0, 0)) {  // outside the rank's window: the digest only
    psd_permute_absorb(sp, st, in, 2);
    psd_permute_absorb(sp, st, in, 0);
    out_lv[t] = st[1];
    return;
  }
  WCtx c = make_ctx(stq, T, p0, 0);
  c = trace_permutation(c, T, sp, st, in, 2);
  c = trace_permutation(c, T, sp, st, in, 0);
  out_lv[t] = st[1];
}

// The tree without a launch per level's TRACE: the digests of every level first (value only: two permutations of latency per level,
// k_mk_level_values), then every node's two permutations traced in ONE launch — a thread per (node, permutation), the padding-only
// permutation starting from the state its thread recomputes.  (One thread per node tracing 4.5 k cells level after level cost ten
// launches of 2.1 ms each whatever the level's size: 21 of C3's 49 ms of witness.)
__global__ __launch_bounds__(64) void k_mk_level_values(const PoseidonSpec* __restrict__ sp, const u256* __restrict__ in_lv, uint32_t n_out,
                                                        u256* __restrict__ out_lv) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n_out) return;
  u256 st[PSD_T] = {sp->cap, u256_zero(), u256_zero()};
  u256 in[PSD_RATE] = {in_lv[2 * t], in_lv[2 * t + 1]};
  psd_permute_absorb(sp, st, in, 2);
  psd_permute_absorb(sp, st, in, 0);
  out_lv[t] = st[1];
}
// levels: level 0 = the lp (padded) leaf digests, level l at offset lp (2 - 2^(1-l)) ... i.e. one after the other; node g of the
// tree (level-major numbering, g < lp - 1) has its cells at base + g * (perm_cells(2) + perm_cells(0))
__global__ __launch_bounds__(64) void k_mk_tree_trace(Streams stq, const FpTables* __restrict__ T, const PoseidonSpec* __restrict__ sp,
                                                      const u256* __restrict__ levels, uint32_t lp, uint64_t base) {
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  const uint32_t g = id >> 1, j = id & 1u;
  if (g + 1 >= lp) return;
  // level of node g: level sizes lp/2, lp/4, ...; in_off = offset of the level it reads in `levels`
  uint32_t sz = lp >> 1, first = 0, in_off = 0, in_sz = lp;
  while (g >= first + sz) {
    first += sz;
    in_off += in_sz;
    in_sz = sz;
    sz >>= 1;
  }
  const uint32_t t = g - first;
  const uint64_t p0 = base + (uint64_t)g * (perm_cells(2) + perm_cells(0)) + (j ? perm_cells(2) : 0);
  const uint32_t cells = j ? perm_cells(0) : perm_cells(2);
  if (!stq.touches(p0, p0 + cells, 0, 0)) return;
  u256 st[PSD_T] = {sp->cap, u256_zero(), u256_zero()};
  u256 in[PSD_RATE] = {levels[in_off + 2 * t], levels[in_off + 2 * t + 1]};
  if (j) psd_permute_absorb(sp, st, in, 2);   // the padding-only permutation starts where the absorbing one ended
  WCtx c = make_ctx(stq, T, p0, 0);
  c = trace_permutation(c, T, sp, st, in, j ? 0 : 2);
}

// ------------------------------------------------------------------ layout (halo2-base assign_threads_in)
// break points from the gate-start bits: the row walk of GateThreadBuilder::assign_all.  A column that
// starts at stream cell S breaks at the first row r in {M-3, M-2 (if that cell starts a gate), M-1}.
__global__ void k_layout_plan(const uint8_t* __restrict__ sel, uint64_t n_cells, uint64_t max_rows, uint64_t* __restrict__ bp, uint64_t cap,
                              uint64_t* __restrict__ n_bp) {
  if (blockIdx.x || threadIdx.x) return;
  uint64_t S = 0, cnt = 0;
  const uint64_t M = max_rows;
  for (;;) {
    uint64_t r;
    if (M >= 3 && S + M - 3 < n_cells && (sel[S + M - 3] & 1)) r = M - 3;
    else if (M >= 2 && S + M - 2 < n_cells && (sel[S + M - 2] & 1)) r = M - 2;
    else r = M - 1;
    if (S + r >= n_cells) break;
    if (cnt < cap) bp[cnt] = r;
    cnt++;
    S += r;
  }
  *n_bp = cnt;
}
__global__ __launch_bounds__(256) void k_layout_columns(const u256* __restrict__ stream, uint64_t n_cells, const uint64_t* __restrict__ starts,
                                                        const uint64_t* __restrict__ bp, uint64_t n_bp, uint32_t k, u256* __restrict__ cols,
                                                        const u256* __restrict__ blind, uint32_t n_blind, uint64_t col_lo, uint64_t col_hi) {
  const uint64_t rows = 1ull << k;
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t total = (col_hi - col_lo) * rows;
  if (idx >= total) return;
  uint64_t col = col_lo + (idx >> k), row = idx & (rows - 1);
  uint64_t start = starts[col];
  uint64_t len = col < n_bp ? bp[col] + 1 : n_cells - start;  // cells held by this column
  u256 v = u256_zero();
  if (row < len) v = ld256(stream + start + row);
  else if (blind && row >= rows - n_blind) v = ld256(blind + col * n_blind + (row - (rows - n_blind)));
  st256(cols + idx, v);
}
__global__ __launch_bounds__(256) void k_layout_lookup(const u256* __restrict__ lk, uint64_t n_cells, uint64_t max_rows, uint32_t k, uint64_t n_cols,
                                                       u256* __restrict__ cols, const u256* __restrict__ blind, uint32_t n_blind, uint64_t col_lo) {
  const uint64_t rows = 1ull << k;
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_cols * rows) return;
  uint64_t col = col_lo + (idx >> k), row = idx & (rows - 1);
  uint64_t src = col * max_rows + row;
  u256 v = u256_zero();
  if (row < max_rows && src < n_cells) v = ld256(lk + src);
  else if (blind && row >= rows - n_blind) v = ld256(blind + col * n_blind + (row - (rows - n_blind)));
  st256(cols + idx, v);
}

// column-layout image of the constant-cell flags (bit 1 of the keygen flag byte): mask[col][row] = 1 when the cell
// laid out there is a data-independent constant
__global__ __launch_bounds__(256) void k_layout_const_mask(const uint8_t* __restrict__ flags, uint64_t n_cells, const uint64_t* __restrict__ starts,
                                                           const uint64_t* __restrict__ bp, uint64_t n_bp, uint32_t k, uint8_t* __restrict__ mask) {
  const uint64_t rows = 1ull << k;
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (n_bp + 1) * rows) return;
  uint64_t col = idx >> k, row = idx & (rows - 1);
  uint64_t start = starts[col];
  uint64_t len = col < n_bp ? bp[col] + 1 : n_cells - start;
  mask[idx] = row < len ? (flags[start + row] >> 1) & 1 : 0;
}
// column-layout image of the gate selectors as field elements: q[col][row] = 1 where a gate starts (bit 0 of the flag byte)
__global__ __launch_bounds__(256) void k_layout_selectors(const uint8_t* __restrict__ flags, uint64_t n_cells, const uint64_t* __restrict__ starts,
                                                          const uint64_t* __restrict__ bp, uint64_t n_bp, uint32_t k, u256* __restrict__ q) {
  const uint64_t rows = 1ull << k;
  uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (n_bp + 1) * rows) return;
  uint64_t col = idx >> k, row = idx & (rows - 1);
  uint64_t start = starts[col];
  uint64_t len = col < n_bp ? bp[col] + 1 : n_cells - start;
  // the last cell of a column that is not the last one is the cell the next column starts with again (break points sit on
  // gate boundaries: it closes a gate here and opens one there), so its selector is enabled in the next column only
  const uint64_t sel_len = col < n_bp ? len - 1 : len;
  st256(q + idx, (row < sel_len && (flags[start + row] & 1)) ? mont_one<Fr>() : u256_zero());
}
// scalars' = mask ? v : 0 (constant part) or mask ? 0 : v (variable part)
__global__ __launch_bounds__(256) void k_mask_select(const u256* __restrict__ in, const uint8_t* __restrict__ mask, uint64_t n, int keep_const,
                                                     u256* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool m = mask[i] != 0;
  st256(out + i, (m == (keep_const != 0)) ? ld256(in + i) : u256_zero());
}

// ------------------------------------------------------------------ helpers for the host-pointer ABI
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    return VDB_OK;
  }
  template <class U>
  U* as() { return (U*)p; }
};
#define TRY(x)             \
  do {                     \
    int _rc = (x);         \
    if (_rc) return _rc;   \
  } while (0)

static int check_err_flag(int* derr) {
  int h = 0;
  VDB_HIP(hipMemcpyAsync(&h, derr, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  if (h) {
    set_error("data-dependent failure the reference turns into a panic (division by zero / index out of range)");
    return VDB_ERR_DOMAIN;
  }
  return VDB_OK;
}

// ---- device-level drivers -------------------------------------------------------------------
int wit_distance_dev(FpEntry* fp, int metric, const u256* a, const u256* b, size_t n_pairs, size_t dim, Streams st, uint64_t adv_off,
                     uint64_t lk_off, u256* result) {
  DistLayout dl;
  TRY(dist_layout(fp->host, metric, dim, &dl));
  TRY(inv_list_attach(st, n_pairs * dl.total_cells));
  TRY(set_winv(st, fp->dev));
  InstMap im{adv_off, lk_off, 1, dl.total_cells, dl.total_lk, 0xffffffffu, 1};
  u256* mid = (u256*)scratch_get(0, n_pairs * 3 * sizeof(u256) + 64);
  if (!mid) return VDB_ERR_OOM;
  TRY(run_distances(st, fp, dl, im, (uint32_t)n_pairs, a, b, mid, result));
  return inv_list_fixup(st);
}

// ------------------------------------------------------------------ one FixedPointInstructions call per lane (vdb_wit_fp_op*)
__global__ __launch_bounds__(64) void k_fp_op(Streams st, const FpTables* __restrict__ T, int op, const u256* __restrict__ a, const u256* __restrict__ b,
                                              uint32_t n, uint32_t cells, uint32_t lks, uint64_t adv_off, uint64_t lk_off, u256* __restrict__ result) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  WCtx c = make_ctx(st, T, adv_off + (uint64_t)i * cells, lk_off + (uint64_t)i * lks);
  Gadgets g(c);
  const u256 r = fp_op_apply(g, op, a[i], b ? b[i] : u256_zero());
  result[i] = r;
  if (c.err) atomicOr(st.err, c.err);
}
int wit_fp_op_dev(FpEntry* fp, int op, const u256* a, const u256* b, size_t n, Streams st, uint64_t adv_off, uint64_t lk_off, u256* result) {
  uint32_t sz[2];
  fp_op_size(fp->host, op, sz);
  TRY(inv_list_attach(st, n * (uint64_t)sz[0]));
  TRY(set_winv(st, fp->dev));
  {
    VDB_PROF("k_fp_op");
    hipLaunchKernelGGL(k_fp_op, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx().stream, st, fp->dev, op, a, b, (uint32_t)n, sz[0], sz[1], adv_off, lk_off,
                       result);
  }
  VDB_LAUNCH_CHECK();
  return inv_list_fixup(st);
}

struct NvLayout {
  uint64_t dist, dist_l, qmin, qmin_l, iseq, sel, total, total_l;
};
static int nv_layout(FpEntry* fp, int metric, size_t n, size_t dim, DistLayout* dl, NvLayout* o) {
  TRY(dist_layout(fp->host, metric, dim, dl));
  o->dist = n * dl->total_cells;
  o->dist_l = n * dl->total_lk;
  o->qmin = (n - 1) * (uint64_t)fp->host.sz.qmin[0];
  o->qmin_l = (n - 1) * (uint64_t)fp->host.sz.qmin[1];
  o->iseq = 12ull * n;
  o->sel = dim * (1 + 3ull * n);
  o->total = o->dist + o->qmin + o->iseq + o->sel;
  o->total_l = o->dist_l + o->qmin_l;
  return VDB_OK;
}
int wit_nearest_dev(FpEntry* fp, int metric, const u256* query, const u256* vectors, size_t n, size_t dim, Streams st, uint64_t adv_off,
                    uint64_t lk_off, u256* ind, u256* result) {
  DistLayout dl;
  NvLayout nl;
  TRY(nv_layout(fp, metric, n, dim, &dl, &nl));
  TRY(inv_list_attach(st, nl.total));
  TRY(set_winv(st, fp->dev));
  InstMap im{adv_off, lk_off, 1, dl.total_cells, dl.total_lk, 0xffffffffu, 0xffffffffu};  // (vector_i, query)
  u256* mid = (u256*)scratch_get(0, (n * 5 + 8) * sizeof(u256));
  if (!mid) return VDB_ERR_OOM;
  u256* dist = mid + 3 * n;
  u256* pm = dist + n;
  TRY(run_distances(st, fp, dl, im, (uint32_t)n, vectors, query, mid, dist));
  hipStream_t s = ctx().stream;
  {
    VDB_PROF("k_nv_prefix_min");
    hipLaunchKernelGGL(k_nv_prefix_min, dim3(1), dim3(1), 0, s, fp->dev, dist, (uint32_t)n, pm);
  }
  VDB_LAUNCH_CHECK();
  if (n > 1) {
    {
      VDB_PROF("k_nv_qmin");
      hipLaunchKernelGGL(k_nv_qmin, dim3((unsigned)((n - 1 + 63) / 64)), dim3(64), 0, s, st, fp->dev, adv_off + nl.dist, lk_off + nl.dist_l, dist, pm,
                       (uint32_t)n);
    }
    VDB_LAUNCH_CHECK();
  }
  {
    VDB_PROF("k_nv_is_equal");
    hipLaunchKernelGGL(k_nv_is_equal, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, st, fp->dev, adv_off + nl.dist + nl.qmin, dist, pm, (uint32_t)n,
                     ind);
  }
  VDB_LAUNCH_CHECK();
  {
    VDB_PROF("k_nv_select");
    hipLaunchKernelGGL(k_nv_select, dim3((unsigned)((dim + 63) / 64)), dim3(64), 0, s, st, fp->dev, adv_off + nl.dist + nl.qmin + nl.iseq, vectors, ind,
                     (uint32_t)n, (uint32_t)dim, result);
  }
  VDB_LAUNCH_CHECK();
  return inv_list_fixup(st);
}

static int km_layout(FpEntry* fp, int metric, size_t n, size_t dim, size_t K, DistLayout* dl, KmLayout* kl) {
  TRY(dist_layout(fp->host, metric, dim, dl));
  const Sizes& z = fp->host.sz;
  kl->N = (uint32_t)n;
  kl->D = (uint32_t)dim;
  kl->K = (uint32_t)K;
  kl->per_vec = K * dl->total_cells + (K - 1) * (uint64_t)z.qmin[0] + K * 20ull;
  kl->per_vec_l = K * dl->total_lk + (K - 1) * (uint64_t)z.qmin[1];
  kl->assign = n * kl->per_vec;
  kl->assign_l = n * kl->per_vec_l;
  kl->sizes = (n - 1) * K * 4ull;
  kl->per_cluster = n * (8 + 8ull * dim) + (n - 1) * dim * 4ull + dim * (uint64_t)z.qdiv[0];
  kl->per_cluster_l = dim * (uint64_t)z.qdiv[1];
  kl->iter = kl->assign + kl->sizes + K * kl->per_cluster;
  kl->iter_l = kl->assign_l + K * kl->per_cluster_l;
  return VDB_OK;
}
int wit_kmeans_dev(FpEntry* fp, int metric, const u256* vectors, size_t n, size_t dim, size_t K, size_t I, int zero_cached, Streams st,
                   uint64_t adv_off, uint64_t lk_off, u256* cent_out, u256* ind_out) {
  DistLayout dl;
  KmLayout kl;
  TRY(km_layout(fp, metric, n, dim, K, &dl, &kl));
  TRY(inv_list_attach(st, I * kl.iter));
  TRY(set_winv(st, fp->dev));
  hipStream_t s = ctx().stream;
  size_t need = (n * K * 4 + K * dim * 2 + K + K * n * dim + 16) * sizeof(u256);
  u256* buf = (u256*)scratch_get(0, need);
  if (!buf) return VDB_ERR_OOM;
  u256* mid = buf;
  u256* dist = mid + 3 * n * K;
  u256* cent = dist + n * K;
  u256* sums = cent + K * dim;
  u256* sizes = sums + K * dim;
  u256* filt = sizes + K;
  // preamble: load_constant(quantization(1.0)); load_zero()
  {
    VDB_PROF("k_push_cells");
    hipLaunchKernelGGL(k_push_cells, dim3(1), dim3(1), 0, s, st, adv_off, fp->host.c_one_q, u256_zero(), zero_cached ? 1u : 2u);
  }
  VDB_LAUNCH_CHECK();
  uint64_t pos = adv_off + (zero_cached ? 1 : 2), lpos = lk_off;
  VDB_HIP(hipMemcpyAsync(cent, vectors, K * dim * sizeof(u256), hipMemcpyDeviceToDevice, s));
  u256 scale_inv = mont_inv<Fr>(fp->host.scale);
  for (size_t it = 0; it < I; it++) {
    InstMap im{pos, lpos, (uint32_t)K, kl.per_vec, kl.per_vec_l, (uint32_t)K, (uint32_t)K};  // distance(centroid_k, vector_v)
    TRY(run_distances(st, fp, dl, im, (uint32_t)(n * K), cent, vectors, mid, dist));
    {
      VDB_PROF("k_km_assign");
      hipLaunchKernelGGL(k_km_assign, dim3((unsigned)((n + 63) / 64), (unsigned)(2 * K)), dim3(64), 0, s, st, fp->dev, kl, dl, pos, lpos, dist, ind_out);
    }
    VDB_LAUNCH_CHECK();
    {
      VDB_PROF("k_km_sizes");
      hipLaunchKernelGGL(k_km_sizes, dim3((unsigned)K), dim3(64), 0, s, st, fp->dev, kl, pos + kl.assign, ind_out, sizes);
    }
    VDB_LAUNCH_CHECK();
    uint64_t cb = pos + kl.assign + kl.sizes, clb = lpos + kl.assign_l;
    {
      VDB_PROF("k_km_filter");
      hipLaunchKernelGGL(k_km_filter, dim3((unsigned)((K * n * ((dim + KM_PF - 1) / KM_PF) + 63) / 64)), dim3(64), 0, s, st, fp->dev, kl, cb, vectors,
                         ind_out, scale_inv, filt);
    }
    VDB_LAUNCH_CHECK();
    {
      VDB_PROF("k_km_sum");
      hipLaunchKernelGGL(k_km_sum, dim3((unsigned)(K * dim)), dim3(64), 0, s, st, fp->dev, kl, cb, filt, sums);
    }
    VDB_LAUNCH_CHECK();
    {
      VDB_PROF("k_km_div");
      hipLaunchKernelGGL(k_km_div, dim3((unsigned)((K * dim + 63) / 64), 8), dim3(64), 0, s, st, fp->dev, kl, cb, clb, sums, sizes, cent);
    }
    VDB_LAUNCH_CHECK();
    pos += kl.iter;
    lpos += kl.iter_l;
  }
  VDB_HIP(hipMemcpyAsync(cent_out, cent, K * dim * sizeof(u256), hipMemcpyDeviceToDevice, s));
  return inv_list_fixup(st);
}

struct MkLayout {
  uint32_t nperm;
  uint64_t leaf_cells, leaves, n_leaves_pow2, zero_cell, total;
};
static void mk_layout(size_t n, size_t dim, int zero_cached, MkLayout* o) {
  o->nperm = (uint32_t)((dim + 1) / 2 + (dim % 2 == 0 ? 1 : 0));
  o->leaf_cells = 0;
  for (uint32_t p = 0; p < o->nperm; p++) {
    size_t off = 2 * (size_t)p;
    o->leaf_cells += perm_cells(off < dim ? (int)(dim - off < 2 ? dim - off : 2) : 0);
  }
  o->leaves = n * o->leaf_cells;
  uint64_t lp = 1;
  while (lp < n) lp <<= 1;
  o->n_leaves_pow2 = lp;
  o->zero_cell = (lp > n && !zero_cached) ? 1 : 0;
  o->total = o->leaves + o->zero_cell + (lp - 1) * (uint64_t)(perm_cells(2) + perm_cells(0));
}
int wit_merkle_dev(const u256* vectors, size_t n, size_t dim, int zero_cached, Streams st, uint64_t adv_off, u256* root_out) {
  FpEntry* fp;
  TRY(get_fp(48, 13, &fp));  // only GateChip primitives are used: P and L are irrelevant
  TRY(set_winv(st, fp->dev));
  const PoseidonSpec* sp;
  TRY(poseidon_spec_dev(&sp, nullptr));
  MkLayout ml;
  mk_layout(n, dim, zero_cached, &ml);
  hipStream_t s = ctx().stream;
  size_t need = (n * ml.nperm * PSD_T + 2 * ml.n_leaves_pow2 + 8) * sizeof(u256);
  u256* buf = (u256*)scratch_get(0, need);
  if (!buf) return VDB_ERR_OOM;
  u256* states = buf;
  u256* lva = states + n * ml.nperm * PSD_T;
  u256* lvb = lva + ml.n_leaves_pow2;
  VDB_HIP(hipMemsetAsync(lva, 0, ml.n_leaves_pow2 * sizeof(u256), s));
  {
    VDB_PROF("k_mk_leaf_states");
    hipLaunchKernelGGL(k_mk_leaf_states, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, sp, vectors, (uint32_t)n, (uint32_t)dim, ml.nperm, states, lva);
  }
  VDB_LAUNCH_CHECK();
  {
    VDB_PROF("k_mk_leaf_trace");
    hipLaunchKernelGGL(k_mk_leaf_trace, dim3((unsigned)((n * ml.nperm + 63) / 64)), dim3(64), 0, s, st, fp->dev, sp, vectors, (uint32_t)n, (uint32_t)dim,
                     ml.nperm, adv_off, ml.leaf_cells, states);
  }
  VDB_LAUNCH_CHECK();
  uint64_t pos = adv_off + ml.leaves;
  if (ml.zero_cell) {
    {
      VDB_PROF("k_push_cells");
      hipLaunchKernelGGL(k_push_cells, dim3(1), dim3(1), 0, s, st, pos, u256_zero(), u256_zero(), 1u);
    }
    VDB_LAUNCH_CHECK();
    pos += 1;
  }
  // the tree: every level's digests (lva holds the levels one after the other: lp + lp / 2 + ... + 1 < 2 lp entries = lva | lvb), then one
  // launch that traces all lp - 1 nodes (VDB_MK_TREE=0: a launch per level, each thread tracing its node's two permutations)
  static const bool tree_on = !(getenv("VDB_MK_TREE") && getenv("VDB_MK_TREE")[0] == '0');
  const uint64_t lp = ml.n_leaves_pow2;
  if (tree_on && lp > 1 && lp <= (1u << 30)) {
    uint64_t lv = lp, off = 0;
    while (lv > 1) {
      const uint64_t no = lv / 2;
      {
        VDB_PROF("k_mk_level_values");
        hipLaunchKernelGGL(k_mk_level_values, dim3((unsigned)((no + 63) / 64)), dim3(64), 0, s, sp, lva + off, (uint32_t)no, lva + off + lv);
      }
      VDB_LAUNCH_CHECK();
      off += lv;
      lv = no;
    }
    {
      VDB_PROF("k_mk_tree_trace");
      hipLaunchKernelGGL(k_mk_tree_trace, dim3((unsigned)((2 * (lp - 1) + 63) / 64)), dim3(64), 0, s, st, fp->dev, sp, lva, (uint32_t)lp, pos);
    }
    VDB_LAUNCH_CHECK();
    VDB_HIP(hipMemcpyAsync(root_out, lva + off, sizeof(u256), hipMemcpyDeviceToDevice, s));
    return VDB_OK;
  }
  uint64_t lv = ml.n_leaves_pow2;
  while (lv > 1) {
    uint64_t no = lv / 2;
    {
      VDB_PROF("k_mk_node");
      hipLaunchKernelGGL(k_mk_node, dim3((unsigned)((no + 63) / 64)), dim3(64), 0, s, st, fp->dev, sp, lva, (uint32_t)no, pos, lvb);
    }
    VDB_LAUNCH_CHECK();
    pos += no * (uint64_t)(perm_cells(2) + perm_cells(0));
    std::swap(lva, lvb);
    lv = no;
  }
  VDB_HIP(hipMemcpyAsync(root_out, lva, sizeof(u256), hipMemcpyDeviceToDevice, s));
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

// rank window applied by the *_dev witness entry points (vdb_wit_set_window); full range by default; per device context
#define g_win (vdb::ctx().win)

// break points and their prefix sums (column c starts at stream cell starts[c]) -> device scratch slot 1
static int upload_break_points(const uint64_t* break_points, uint64_t n_bp, uint64_t** dbp, uint64_t** dstarts) {
  static thread_local std::vector<uint64_t> h;  // pageable source: hipMemcpyAsync stages it before returning
  h.resize(2 * n_bp + 2);
  uint64_t acc = 0;
  h[n_bp] = 0;
  for (uint64_t i = 0; i < n_bp; i++) {
    h[i] = break_points[i];
    acc += break_points[i];
    h[n_bp + 1 + i] = acc;
  }
  uint64_t* d = (uint64_t*)scratch_get(1, (2 * n_bp + 2) * sizeof(uint64_t));
  if (!d) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(d, h.data(), (2 * n_bp + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx().stream));
  *dbp = d;
  *dstarts = d + n_bp;
  return VDB_OK;
}

// upload helper for the host-pointer entry points
static int upload(DevBuf& d, const void* src, size_t bytes) {
  TRY(d.alloc(bytes));
  if (bytes) VDB_HIP(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx().stream));
  return VDB_OK;
}
static int download(void* dst, const void* src, size_t bytes) {
  if (dst && bytes) VDB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  return VDB_OK;
}
struct HostStreams {
  DevBuf adv, sel, lk, err;
  Streams st;
  int init(uint64_t cells, uint64_t lookups, bool want_sel) {
    TRY(adv.alloc(cells * sizeof(u256)));
    TRY(lk.alloc(lookups * sizeof(u256)));
    if (want_sel) TRY(sel.alloc(cells));
    TRY(err.alloc(sizeof(int)));
    VDB_HIP(hipMemsetAsync(err.p, 0, sizeof(int), ctx().stream));
    st.adv = adv.as<u256>();
    st.sel = want_sel ? sel.as<uint8_t>() : nullptr;
    st.lk = lk.as<u256>();
    st.err = err.as<int>();
    st.inv_pos = nullptr;
    st.inv_val = nullptr;
    st.inv_cnt = nullptr;
    st.inv_cap = 0;
    st.rlo = 0;
    st.rhi = ~0ull;
    st.rllo = 0;
    st.rlhi = ~0ull;
    return VDB_OK;
  }
  int finish(vdb_fr* stream_out, vdb_fr* lookup_out, uint8_t* sel_out, uint64_t cells, uint64_t lookups) {
    TRY(download(stream_out, st.adv, cells * sizeof(u256)));
    TRY(download(lookup_out, st.lk, lookups * sizeof(u256)));
    if (sel_out && st.sel) TRY(download(sel_out, st.sel, cells));
    return check_err_flag(st.err);
  }
};

extern "C" {

int vdb_fp_quantize(uint32_t precision_bits, const double* x, vdb_fr* out, size_t n) {
  VDB_ARG(x && out && precision_bits >= 32 && precision_bits <= 63, "bad argument");
  for (size_t i = 0; i < n; i++) {
    u256 q = quantize_host(precision_bits, x[i]);
    memcpy(&out[i], &q, 32);
  }
  return VDB_OK;
}
int vdb_fp_dequantize(uint32_t precision_bits, const vdb_fr* x, double* out, size_t n) {
  VDB_ARG(x && out && precision_bits >= 32 && precision_bits <= 63, "bad argument");
  for (size_t i = 0; i < n; i++) {
    u256 v;
    memcpy(&v, &x[i], 32);
    out[i] = dequantize_host(precision_bits, v);
  }
  return VDB_OK;
}

int vdb_wit_distance_size(int metric, uint32_t P, uint32_t L, size_t n_pairs, size_t dim, uint64_t* cells, uint64_t* lookups) {
  VDB_REQUIRE_INIT();
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  DistLayout dl;
  TRY(dist_layout(fp->host, metric, dim, &dl));
  if (cells) *cells = n_pairs * dl.total_cells;
  if (lookups) *lookups = n_pairs * dl.total_lk;
  return VDB_OK;
}
int vdb_wit_distance(int metric, uint32_t P, uint32_t L, const vdb_fr* a, const vdb_fr* b, size_t n_pairs, size_t dim, vdb_fr* stream_out,
                     vdb_fr* lookup_out, uint8_t* selector_out, vdb_fr* result_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(a && b && dim > 0, "null pointer or dim == 0");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  uint64_t cells, lookups;
  TRY(vdb_wit_distance_size(metric, P, L, n_pairs, dim, &cells, &lookups));
  if (n_pairs == 0) return VDB_OK;
  DevBuf da, db, dres;
  HostStreams hs;
  TRY(upload(da, a, n_pairs * dim * sizeof(u256)));
  TRY(upload(db, b, n_pairs * dim * sizeof(u256)));
  TRY(dres.alloc(n_pairs * sizeof(u256)));
  TRY(hs.init(cells, lookups, selector_out != nullptr));
  TRY(wit_distance_dev(fp, metric, da.as<u256>(), db.as<u256>(), n_pairs, dim, hs.st, 0, 0, dres.as<u256>()));
  TRY(download(result_out, dres.p, n_pairs * sizeof(u256)));
  return hs.finish(stream_out, lookup_out, selector_out, cells, lookups);
}

int vdb_wit_distance_dev(int metric, uint32_t P, uint32_t L, const vdb_fr* a_dev, const vdb_fr* b_dev, size_t n_pairs, size_t dim, vdb_fr* stream_dev,
                         vdb_fr* lookup_dev, uint8_t* selector_dev, vdb_fr* result_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(a_dev && b_dev && stream_dev && lookup_dev && result_dev && n_pairs > 0 && dim > 0, "null pointer or empty input");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  int* derr = (int*)scratch_get(1, 64);
  if (!derr) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), ctx().stream));
  Streams st{as_u256(stream_dev), selector_dev, as_u256(lookup_dev), derr, nullptr, nullptr, nullptr, 0, g_win[0], g_win[1], g_win[2], g_win[3]};
  TRY(wit_distance_dev(fp, metric, as_u256(a_dev), as_u256(b_dev), n_pairs, dim, st, 0, 0, as_u256(result_dev)));
  return check_err_flag(derr);
}

int vdb_wit_fp_op_size(int op, uint32_t P, uint32_t L, size_t n, uint64_t* cells, uint64_t* lookups) {
  VDB_REQUIRE_INIT();
  VDB_ARG(op >= 0 && op < FP_OP_COUNT, "unknown fixed-point operation");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  uint32_t sz[2];
  fp_op_size(fp->host, op, sz);
  if (cells) *cells = n * (uint64_t)sz[0];
  if (lookups) *lookups = n * (uint64_t)sz[1];
  return VDB_OK;
}
int vdb_wit_fp_op(int op, uint32_t P, uint32_t L, const vdb_fr* a, const vdb_fr* b, size_t n, vdb_fr* stream_out, vdb_fr* lookup_out,
                  uint8_t* selector_out, vdb_fr* result_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(a && result_out, "null pointer");
  uint64_t cells, lookups;
  TRY(vdb_wit_fp_op_size(op, P, L, n, &cells, &lookups));
  if (n == 0) return VDB_OK;
  VDB_ARG(n <= 0xffffffffull, "too many instances for one call");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  DevBuf da, db, dres;
  HostStreams hs;
  TRY(upload(da, a, n * sizeof(u256)));
  if (b) TRY(upload(db, b, n * sizeof(u256)));
  TRY(dres.alloc(n * sizeof(u256)));
  TRY(hs.init(cells, lookups, selector_out != nullptr));
  TRY(wit_fp_op_dev(fp, op, da.as<u256>(), b ? db.as<u256>() : nullptr, n, hs.st, 0, 0, dres.as<u256>()));
  TRY(download(result_out, dres.p, n * sizeof(u256)));
  return hs.finish(stream_out, lookup_out, selector_out, cells, lookups);
}
int vdb_wit_fp_op_dev(int op, uint32_t P, uint32_t L, const vdb_fr* a_dev, const vdb_fr* b_dev, size_t n, vdb_fr* stream_dev, vdb_fr* lookup_dev,
                      uint8_t* selector_dev, vdb_fr* result_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(op >= 0 && op < FP_OP_COUNT, "unknown fixed-point operation");
  VDB_ARG(a_dev && stream_dev && lookup_dev && result_dev && n > 0 && n <= 0xffffffffull, "null pointer or empty input");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  int* derr = (int*)scratch_get(1, 64);
  if (!derr) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), ctx().stream));
  Streams st{as_u256(stream_dev), selector_dev, as_u256(lookup_dev), derr, nullptr, nullptr, nullptr, 0, g_win[0], g_win[1], g_win[2], g_win[3]};
  TRY(wit_fp_op_dev(fp, op, as_u256(a_dev), as_u256(b_dev), n, st, 0, 0, as_u256(result_dev)));
  return check_err_flag(derr);
}

int vdb_wit_nearest_size(int metric, uint32_t P, uint32_t L, size_t n, size_t dim, uint64_t* cells, uint64_t* lookups) {
  VDB_REQUIRE_INIT();
  VDB_ARG(n > 0, "empty database");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  DistLayout dl;
  NvLayout nl;
  TRY(nv_layout(fp, metric, n, dim, &dl, &nl));
  if (cells) *cells = nl.total;
  if (lookups) *lookups = nl.total_l;
  return VDB_OK;
}
int vdb_wit_nearest(int metric, uint32_t P, uint32_t L, const vdb_fr* query, const vdb_fr* vectors, size_t n, size_t dim, vdb_fr* stream_out,
                    vdb_fr* lookup_out, uint8_t* selector_out, vdb_fr* indicator_out, vdb_fr* result_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(query && vectors && n > 0 && dim > 0, "null pointer or empty input");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  uint64_t cells, lookups;
  TRY(vdb_wit_nearest_size(metric, P, L, n, dim, &cells, &lookups));
  DevBuf dq, dv, dind, dres;
  HostStreams hs;
  TRY(upload(dq, query, dim * sizeof(u256)));
  TRY(upload(dv, vectors, n * dim * sizeof(u256)));
  TRY(dind.alloc(n * sizeof(u256)));
  TRY(dres.alloc(dim * sizeof(u256)));
  TRY(hs.init(cells, lookups, selector_out != nullptr));
  TRY(wit_nearest_dev(fp, metric, dq.as<u256>(), dv.as<u256>(), n, dim, hs.st, 0, 0, dind.as<u256>(), dres.as<u256>()));
  TRY(download(indicator_out, dind.p, n * sizeof(u256)));
  TRY(download(result_out, dres.p, dim * sizeof(u256)));
  return hs.finish(stream_out, lookup_out, selector_out, cells, lookups);
}

int vdb_wit_nearest_dev(int metric, uint32_t P, uint32_t L, const vdb_fr* query_dev, const vdb_fr* vectors_dev, size_t n, size_t dim, vdb_fr* stream_dev,
                        vdb_fr* lookup_dev, uint8_t* selector_dev, vdb_fr* indicator_dev, vdb_fr* result_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(query_dev && vectors_dev && stream_dev && lookup_dev && indicator_dev && result_dev && n > 0 && dim > 0, "null pointer or empty input");
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  int* derr = (int*)scratch_get(1, 64);
  if (!derr) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), ctx().stream));
  // the rank window (vdb_wit_set_window): a rank stores the cells of its own columns — the distances, N-way parallel and nearly all of
  // the cells, exit early outside it — while every rank computes every value (the N distances, the short minimum chain)
  Streams st{as_u256(stream_dev), selector_dev, as_u256(lookup_dev), derr, nullptr, nullptr, nullptr, 0, g_win[0], g_win[1], g_win[2], g_win[3]};
  TRY(wit_nearest_dev(fp, metric, as_u256(query_dev), as_u256(vectors_dev), n, dim, st, 0, 0, as_u256(indicator_dev), as_u256(result_dev)));
  return check_err_flag(derr);
}

int vdb_wit_kmeans_size(int metric, uint32_t P, uint32_t L, size_t n, size_t dim, size_t K, size_t I, int zero_cached, uint64_t* cells,
                        uint64_t* lookups) {
  VDB_REQUIRE_INIT();
  if (!(K < n) || K == 0) {
    set_error("kmeans requires 0 < K < #vectors (vectordb.rs:238 assert)");
    return VDB_ERR_DOMAIN;
  }
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  DistLayout dl;
  KmLayout kl;
  TRY(km_layout(fp, metric, n, dim, K, &dl, &kl));
  if (cells) *cells = (zero_cached ? 1 : 2) + I * kl.iter;
  if (lookups) *lookups = I * kl.iter_l;
  return VDB_OK;
}
int vdb_wit_set_window(uint64_t adv_lo, uint64_t adv_hi, uint64_t lookup_lo, uint64_t lookup_hi) {
  VDB_ARG(adv_lo <= adv_hi && lookup_lo <= lookup_hi, "empty or inverted window");
  g_win[0] = adv_lo;
  g_win[1] = adv_hi;
  g_win[2] = lookup_lo;
  g_win[3] = lookup_hi;
  return VDB_OK;
}
int vdb_wit_kmeans_dev(int metric, uint32_t P, uint32_t L, const vdb_fr* vectors_dev, size_t n, size_t dim, size_t K, size_t I, int zero_cached,
                       vdb_fr* stream_dev, vdb_fr* lookup_dev, uint8_t* selector_dev, vdb_fr* centroids_dev, vdb_fr* indicators_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(vectors_dev && stream_dev && lookup_dev && centroids_dev && indicators_dev && dim > 0, "null pointer");
  uint64_t cells, lookups;
  TRY(vdb_wit_kmeans_size(metric, P, L, n, dim, K, I, zero_cached, &cells, &lookups));
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  int* derr = (int*)scratch_get(1, 64);
  if (!derr) return VDB_ERR_OOM;
  VDB_HIP(hipMemsetAsync(derr, 0, sizeof(int), ctx().stream));
  Streams st{as_u256(stream_dev), selector_dev, as_u256(lookup_dev), derr, nullptr, nullptr, nullptr, 0, g_win[0], g_win[1], g_win[2], g_win[3]};
  TRY(wit_kmeans_dev(fp, metric, as_u256(vectors_dev), n, dim, K, I, zero_cached, st, 0, 0, as_u256(centroids_dev), as_u256(indicators_dev)));
  return check_err_flag(derr);
}
int vdb_wit_kmeans(int metric, uint32_t P, uint32_t L, const vdb_fr* vectors, size_t n, size_t dim, size_t K, size_t I, int zero_cached,
                   vdb_fr* stream_out, vdb_fr* lookup_out, uint8_t* selector_out, vdb_fr* centroids_out, vdb_fr* indicators_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(vectors && dim > 0, "null pointer");
  uint64_t cells, lookups;
  TRY(vdb_wit_kmeans_size(metric, P, L, n, dim, K, I, zero_cached, &cells, &lookups));
  FpEntry* fp;
  TRY(get_fp(P, L, &fp));
  DevBuf dv, dc, di;
  HostStreams hs;
  TRY(upload(dv, vectors, n * dim * sizeof(u256)));
  TRY(dc.alloc(K * dim * sizeof(u256)));
  TRY(di.alloc(n * K * sizeof(u256)));
  TRY(hs.init(cells, lookups, selector_out != nullptr));
  TRY(wit_kmeans_dev(fp, metric, dv.as<u256>(), n, dim, K, I, zero_cached, hs.st, 0, 0, dc.as<u256>(), di.as<u256>()));
  TRY(download(centroids_out, dc.p, K * dim * sizeof(u256)));
  TRY(download(indicators_out, di.p, n * K * sizeof(u256)));
  return hs.finish(stream_out, lookup_out, selector_out, cells, lookups);
}

int vdb_wit_merkle_size(size_t n, size_t dim, int zero_cached, uint64_t* cells) {
  VDB_ARG(n > 0 && cells, "empty database");
  MkLayout ml;
  mk_layout(n, dim, zero_cached, &ml);
  *cells = ml.total;
  return VDB_OK;
}
int vdb_wit_merkle_dev(const vdb_fr* vectors_dev, size_t n, size_t dim, int zero_cached, vdb_fr* stream_dev, uint8_t* selector_dev, vdb_fr* root_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(vectors_dev && stream_dev && root_dev && n > 0, "null pointer");
  Streams st{as_u256(stream_dev), selector_dev, nullptr, nullptr, nullptr, nullptr, nullptr, 0, g_win[0], g_win[1], 0, ~0ull};
  return wit_merkle_dev(as_u256(vectors_dev), n, dim, zero_cached, st, 0, as_u256(root_dev));
}
int vdb_wit_merkle(const vdb_fr* vectors, size_t n, size_t dim, int zero_cached, vdb_fr* stream_out, uint8_t* selector_out, vdb_fr* root_out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(vectors && n > 0, "null pointer or empty database");
  uint64_t cells;
  TRY(vdb_wit_merkle_size(n, dim, zero_cached, &cells));
  DevBuf dv, droot;
  HostStreams hs;
  TRY(upload(dv, vectors, n * dim * sizeof(u256)));
  TRY(droot.alloc(sizeof(u256)));
  TRY(hs.init(cells, 0, selector_out != nullptr));
  TRY(wit_merkle_dev(dv.as<u256>(), n, dim, zero_cached, hs.st, 0, droot.as<u256>()));
  TRY(download(root_out, droot.p, sizeof(u256)));
  return hs.finish(stream_out, nullptr, selector_out, cells, 0);
}

// ---- b4 layout ---------------------------------------------------------------------------------
int vdb_layout_plan_dev(const uint8_t* selector_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t* break_points_out, uint64_t cap,
                        uint64_t* n_break_points) {
  VDB_REQUIRE_INIT();
  VDB_ARG(selector_dev && n_break_points && k >= 3 && k <= 28 && ((uint64_t)1 << k) > minimum_rows + 4, "bad argument");
  uint64_t max_rows = ((uint64_t)1 << k) - minimum_rows;
  uint64_t est = n_cells / (max_rows - 3) + 2;
  uint64_t* d = (uint64_t*)scratch_get(1, (est + 1) * sizeof(uint64_t));
  if (!d) return VDB_ERR_OOM;
  {
    VDB_PROF("k_layout_plan");
    hipLaunchKernelGGL(k_layout_plan, dim3(1), dim3(1), 0, ctx().stream, selector_dev, n_cells, max_rows, d + 1, est, d);
  }
  VDB_LAUNCH_CHECK();
  uint64_t nbp = 0;
  VDB_HIP(hipMemcpyAsync(&nbp, d, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  *n_break_points = nbp;
  if (break_points_out) {
    VDB_ARG(cap >= nbp, "break point buffer too small");
    VDB_HIP(hipMemcpy(break_points_out, d + 1, nbp * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return VDB_OK;
}
int vdb_layout_plan(const uint8_t* selector, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t* break_points_out, uint64_t cap,
                    uint64_t* n_break_points) {
  VDB_REQUIRE_INIT();
  VDB_ARG(selector, "null pointer");
  DevBuf ds;
  TRY(upload(ds, selector, n_cells));
  return vdb_layout_plan_dev(ds.as<uint8_t>(), n_cells, k, minimum_rows, break_points_out, cap, n_break_points);
}
int vdb_layout_columns_dev(const vdb_fr* stream_dev, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, uint32_t k, vdb_fr* cols_dev,
                           const vdb_fr* blind_dev, uint32_t n_blind) {
  return vdb_layout_columns_range_dev(stream_dev, n_cells, break_points, n_bp, k, 0, n_bp + 1, cols_dev, blind_dev, n_blind);
}
int vdb_layout_columns_range_dev(const vdb_fr* stream_dev, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, uint32_t k, uint64_t col_lo,
                                 uint64_t col_hi, vdb_fr* cols_dev, const vdb_fr* blind_dev, uint32_t n_blind) {
  VDB_REQUIRE_INIT();
  VDB_ARG(stream_dev && cols_dev && (break_points || n_bp == 0) && k <= 28 && col_lo <= col_hi && col_hi <= n_bp + 1, "bad argument");
  if (col_lo == col_hi) return VDB_OK;
  const uint64_t rows = 1ull << k;
  uint64_t sum = 0;
  for (uint64_t i = 0; i < n_bp; i++) {
    VDB_ARG(break_points[i] < rows, "break point beyond the column height");
    sum += break_points[i];
  }
  VDB_ARG(sum <= n_cells && n_cells - sum <= rows, "break points do not match the stream length");
  uint64_t *dbp, *dst;
  TRY(upload_break_points(break_points, n_bp, &dbp, &dst));
  uint64_t total = (col_hi - col_lo) * rows;
  {
    VDB_PROF("k_layout_columns");
    hipLaunchKernelGGL(k_layout_columns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(stream_dev), n_cells, dst, dbp, n_bp, k,
                     as_u256(cols_dev), blind_dev ? as_u256(blind_dev) : nullptr, n_blind, col_lo, col_hi);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipStreamSynchronize(ctx().stream));  // break_points is a host buffer the caller may free
  return VDB_OK;
}
int vdb_colsrc_build_dev(const vdb_fr* stream_dev, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, uint32_t k, uint64_t col_lo,
                         uint64_t col_hi, const vdb_fr* blind_dev, uint32_t n_blind, vdb_colsrc* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(stream_dev && out_dev && (break_points || n_bp == 0) && k <= 28 && col_lo <= col_hi && col_hi <= n_bp + 1, "bad argument");
  const uint64_t rows = 1ull << k;
  std::vector<vdb_colsrc> h(col_hi - col_lo);
  uint64_t start = 0;
  for (uint64_t c = 0; c < col_hi; c++) {
    if (c < n_bp) VDB_ARG(break_points[c] < rows, "break point beyond the column height");
    const uint64_t len = c < n_bp ? break_points[c] + 1 : n_cells - start;
    VDB_ARG(start <= n_cells && len <= rows && start + len <= n_cells, "break points do not match the stream length");
    if (c >= col_lo) {
      h[c - col_lo].src = stream_dev + start;
      h[c - col_lo].len = len;
      h[c - col_lo].blind = blind_dev ? blind_dev + c * n_blind : nullptr;
    }
    if (c < n_bp) start += break_points[c];
  }
  if (!h.empty()) VDB_HIP(hipMemcpyAsync(out_dev, h.data(), h.size() * sizeof(vdb_colsrc), hipMemcpyHostToDevice, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_colsrc_build_lookup_dev(const vdb_fr* lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t col_lo, uint64_t col_hi,
                                const vdb_fr* blind_dev, uint32_t n_blind, vdb_colsrc* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(out_dev && (lookup_dev || n_cells == 0) && k <= 28 && col_lo <= col_hi && minimum_rows < (1u << k), "bad argument");
  const uint64_t max_rows = (1ull << k) - minimum_rows;
  std::vector<vdb_colsrc> h(col_hi - col_lo);
  for (uint64_t c = col_lo; c < col_hi; c++) {
    const uint64_t start = c * max_rows;
    h[c - col_lo].src = lookup_dev + (start < n_cells ? start : 0);
    h[c - col_lo].len = start < n_cells ? (n_cells - start < max_rows ? n_cells - start : max_rows) : 0;
    h[c - col_lo].blind = blind_dev ? blind_dev + c * n_blind : nullptr;
  }
  if (!h.empty()) VDB_HIP(hipMemcpyAsync(out_dev, h.data(), h.size() * sizeof(vdb_colsrc), hipMemcpyHostToDevice, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_layout_lookup_dev(const vdb_fr* lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, vdb_fr* cols_dev, uint64_t n_cols,
                          const vdb_fr* blind_dev, uint32_t n_blind) {
  VDB_ARG(n_cols * ((((uint64_t)1 << k)) - minimum_rows) >= n_cells, "not enough lookup columns");
  return vdb_layout_lookup_range_dev(lookup_dev, n_cells, k, minimum_rows, 0, n_cols, cols_dev, blind_dev, n_blind);
}
int vdb_layout_lookup_range_dev(const vdb_fr* lookup_dev, uint64_t n_cells, uint32_t k, uint32_t minimum_rows, uint64_t col_lo, uint64_t col_hi,
                                vdb_fr* cols_dev, const vdb_fr* blind_dev, uint32_t n_blind) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols_dev && (lookup_dev || n_cells == 0) && k <= 28 && col_lo <= col_hi, "bad argument");
  uint64_t max_rows = ((uint64_t)1 << k) - minimum_rows;
  const uint64_t n_cols = col_hi - col_lo;
  if (n_cols == 0) return VDB_OK;
  uint64_t total = n_cols << k;
  {
    VDB_PROF("k_layout_lookup");
    hipLaunchKernelGGL(k_layout_lookup, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(lookup_dev), n_cells, max_rows, k, n_cols,
                     as_u256(cols_dev), blind_dev ? as_u256(blind_dev) : nullptr, n_blind, col_lo);
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_layout_const_mask_dev(const uint8_t* flags_dev, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, uint32_t k, uint8_t* mask_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(flags_dev && mask_dev && (break_points || n_bp == 0) && k <= 28, "bad argument");
  const uint64_t rows = 1ull << k;
  uint64_t *dbp, *dst;
  TRY(upload_break_points(break_points, n_bp, &dbp, &dst));
  uint64_t total = (n_bp + 1) * rows;
  {
    VDB_PROF("k_layout_const_mask");
    hipLaunchKernelGGL(k_layout_const_mask, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream, flags_dev, n_cells, dst, dbp, n_bp, k, mask_dev);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_layout_selectors_dev(const uint8_t* flags_dev, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, uint32_t k, vdb_fr* q_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(flags_dev && q_dev && (break_points || n_bp == 0) && k <= 28, "bad argument");
  const uint64_t rows = 1ull << k;
  uint64_t *dbp, *dst;
  TRY(upload_break_points(break_points, n_bp, &dbp, &dst));
  uint64_t total = (n_bp + 1) * rows;
  {
    VDB_PROF("k_layout_selectors");
    hipLaunchKernelGGL(k_layout_selectors, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx().stream, flags_dev, n_cells, dst, dbp, n_bp, k,
                     as_u256(q_dev));
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_mask_select_dev(const vdb_fr* in_dev, const uint8_t* mask_dev, uint64_t n, int keep_const, vdb_fr* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(in_dev && mask_dev && out_dev, "null pointer");
  if (n == 0) return VDB_OK;
  {
    VDB_PROF("k_mask_select");
    hipLaunchKernelGGL(k_mask_select, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, as_u256(in_dev), mask_dev, n, keep_const, as_u256(out_dev));
  }
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}
int vdb_layout_columns(const vdb_fr* stream, uint64_t n_cells, const uint64_t* break_points, uint64_t n_bp, const vdb_fr* lookup, uint64_t n_lookup,
                       uint32_t k, uint32_t minimum_rows, vdb_fr* advice_cols_out, vdb_fr* lookup_cols_out, uint64_t n_lookup_cols) {
  VDB_REQUIRE_INIT();
  VDB_ARG(stream && advice_cols_out, "null pointer");
  const uint64_t rows = 1ull << k;
  DevBuf ds, dc, dl, dlc;
  TRY(upload(ds, stream, n_cells * sizeof(u256)));
  TRY(dc.alloc((n_bp + 1) * rows * sizeof(u256)));
  TRY(vdb_layout_columns_dev(ds.as<vdb_fr>(), n_cells, break_points, n_bp, k, dc.as<vdb_fr>(), nullptr, 0));
  TRY(download(advice_cols_out, dc.p, (n_bp + 1) * rows * sizeof(u256)));
  if (lookup_cols_out && n_lookup_cols) {
    TRY(upload(dl, lookup, n_lookup * sizeof(u256)));
    TRY(dlc.alloc(n_lookup_cols * rows * sizeof(u256)));
    TRY(vdb_layout_lookup_dev(dl.as<vdb_fr>(), n_lookup, k, minimum_rows, dlc.as<vdb_fr>(), n_lookup_cols, nullptr, 0));
    TRY(download(lookup_cols_out, dlc.p, n_lookup_cols * rows * sizeof(u256)));
  }
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}

}  // extern "C"
