// libvdb_hip core: lifecycle (b0), device memory helpers, elementwise Fr kernels, batch inversion,
// and the Montgomery-multiplication micro-benchmark used to calibrate the integer-ALU roofline.
#include <atomic>
#include <chrono>

#include "common.hpp"

namespace vdb {

static Context g_ctxs[VDB_MAX_DEVICES];
static int g_default = 0;               // device of the last vdb_init (0 after vdb_init_devices)
static thread_local int t_cur = -1;     // device this thread chose with vdb_set_device (-1: the process default)
static thread_local int t_hip_dev = -1; // HIP's current device on this thread, as far as this library set it
static thread_local char g_err[512] = "";

Context& ctx() { return g_ctxs[t_cur >= 0 ? t_cur : g_default]; }
bool context_ready(int device) { return device >= 0 && device < VDB_MAX_DEVICES && g_ctxs[device].ready; }
// Test-only switch VDB_TEST_ALIAS_DEVICES=n: the library offers n logical devices where fewer are visible, logical device d
// living on physical device d % visible — separate contexts (streams, work space, tables, srs handles) on one card, so that the
// several-GPUs-in-one-process code runs with n > 1 on a one-GPU box.  Unset (the default): logical = physical.
static int g_phys_count = 0;
int phys_of(int logical) { return g_phys_count > 0 ? logical % g_phys_count : logical; }
int bind_thread() {
  const int d = phys_of(ctx().device);
  if (t_hip_dev != d) {
    VDB_HIP(hipSetDevice(d));
    t_hip_dev = d;
  }
  return VDB_OK;
}
int g_prof_mode = 0;
std::map<std::string, ProfEntry> g_prof;
std::vector<ProfPending> g_prof_pending;
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  return e == hipErrorOutOfMemory ? VDB_ERR_OOM : VDB_ERR_HIP;
}
static std::atomic<uint64_t> g_alloc_ns{0}, g_alloc_bytes{0}, g_alloc_calls{0};
hipError_t timed_malloc(void** p, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc(p, bytes);
  g_alloc_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  g_alloc_bytes += bytes;
  g_alloc_calls += 1;
  return e;
}
void* scratch_get(int slot, size_t bytes) {
  Context& c = ctx();
  if (slot == 2 && c.msm_pending) {
    // the bucket folding of an open deferred MSM is still working in this slot on the second stream
    set_error("scratch slot 2 belongs to the open deferred MSM: call vdb_msm_batch_end first");
    return nullptr;
  }
  if (c.scratch_bytes[slot] >= bytes) return c.scratch[slot];
  if (c.scratch[slot]) (void)hipFree(c.scratch[slot]);
  c.scratch[slot] = nullptr;
  c.scratch_bytes[slot] = 0;
  size_t want = bytes + bytes / 8;
  hipError_t e = timed_malloc(&c.scratch[slot], want);
  if (e != hipSuccess) {
    hip_fail(e, "hipMalloc(scratch)", __FILE__, __LINE__);
    return nullptr;
  }
  c.scratch_bytes[slot] = want;
  return c.scratch[slot];
}

u256 host_fr_from_u64(uint64_t v) { return to_mont<Fr>(u256_from_u64(v)); }
u256 host_root_of_unity(uint32_t k) {
  // ROOT_OF_UNITY = 7^((r-1) >> 28)  (SURVEY App. D); then square down to order 2^k
  u256 e = mod_p<Fr>(), one = u256_from_u64(1), t;
  u256_sub(t, e, one);
  e = u256_shr(t, 28);
  u256 w = mont_pow<Fr>(host_fr_from_u64(7), e);
  for (uint32_t i = k; i < 28; i++) w = fr_mul(w, w);
  return w;
}
u256 host_zeta() {
  // halo2curves bn256 Fr::ZETA = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
  u256 z;
  const uint32_t w[8] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0, 0};
  for (int i = 0; i < 8; i++) z.w[i] = w[i];
  return to_mont<Fr>(z);
}

// ------------------------------------------------------------------ kernels
enum { EW_MUL = 0, EW_ADD = 1, EW_SUB = 2, EW_TO_MONT = 3, EW_FROM_MONT = 4 };
template <int OP>
__global__ __launch_bounds__(256) void k_elementwise(const u256* __restrict__ a, const u256* __restrict__ b, u256* __restrict__ o, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u256 x = ld256(a + i), r;
    if (OP == EW_MUL) r = fr_mul(x, ld256(b + i));
    else if (OP == EW_ADD) r = fr_add(x, ld256(b + i));
    else if (OP == EW_SUB) r = fr_sub(x, ld256(b + i));
    else if (OP == EW_TO_MONT) r = to_mont<Fr>(x);
    else r = from_mont<Fr>(x);
    st256(o + i, r);
  }
}

// Batch inversion: each thread inverts a run of CH elements with Montgomery's trick (one Fermat
// inversion per run).  Zero entries are skipped (0 -> 0) as halo2's BatchInvert does.
#define BINV_CH 32
__global__ __launch_bounds__(256) void k_batch_invert(const u256* __restrict__ in, u256* __restrict__ out, size_t n) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t lo = t * BINV_CH;
  if (lo >= n) return;
  size_t hi = lo + BINV_CH < n ? lo + BINV_CH : n;
  u256 acc = mont_one<Fr>();
  for (size_t i = lo; i < hi; i++) {
    u256 x = ld256(in + i);
    st256(out + i, acc);  // prefix product before i
    if (!u256_is_zero(x)) acc = fr_mul(acc, x);
  }
  acc = mont_inv<Fr>(acc);
  for (size_t i = hi; i-- > lo;) {
    u256 x = ld256(in + i);
    u256 pre = ld256(out + i);
    if (u256_is_zero(x)) {
      st256(out + i, x);
    } else {
      st256(out + i, fr_mul(acc, pre));
      acc = fr_mul(acc, x);
    }
  }
}

// out[i] = (the 512-bit little-endian integer wide[i]) mod r in Montgomery form: how halo2curves' Fr::random / from_u512 turns
// 64 bytes of entropy into a statistically uniform scalar (lo R2 + hi R3 in its terms)
__global__ __launch_bounds__(256) void k_from_wide(const u256* __restrict__ wide, u256* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u256 lo = ld256(wide + 2 * i), hi = ld256(wide + 2 * i + 1);
  st256(out + i, fr_add(to_mont<Fr>(lo), to_mont<Fr>(to_mont<Fr>(hi))));
}

// 4 independent dependency chains per thread so the measurement is throughput-, not latency-bound
__global__ __launch_bounds__(256) void k_bench_mul32(u256* __restrict__ sink, size_t iters) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u256 a = to_mont<Fr>(u256_from_u64(t * 2654435761ull + 12345)), b = fr_add(a, mont_one<Fr>());
  u256 c = fr_add(b, mont_one<Fr>()), d = fr_add(c, mont_one<Fr>());
  for (size_t i = 0; i < iters; i++) {
    a = mont_mul32<Fr>(a, b);
    b = mont_mul32<Fr>(b, c);
    c = mont_mul32<Fr>(c, d);
    d = mont_mul32<Fr>(d, a);
  }
  st256(sink + t, fr_add(fr_add(a, b), fr_add(c, d)));
}
__global__ __launch_bounds__(256) void k_bench_mul(u256* __restrict__ sink, size_t iters) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u256 a = to_mont<Fr>(u256_from_u64(t * 2654435761ull + 12345)), b = fr_add(a, mont_one<Fr>());
  u256 c = fr_add(b, mont_one<Fr>()), d = fr_add(c, mont_one<Fr>());
  for (size_t i = 0; i < iters; i++) {
    a = fr_mul(a, b);
    b = fr_mul(b, c);
    c = fr_mul(c, d);
    d = fr_mul(d, a);
  }
  st256(sink + t, fr_add(fr_add(a, b), fr_add(c, d)));
}

static int grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  size_t cap = (size_t)ctx().cu_count * 8;
  return (int)(g < cap ? (g ? g : 1) : cap);
}

template <int OP>
static int elementwise_host(const vdb_fr* a, const vdb_fr* b, vdb_fr* out, size_t n) {
  VDB_REQUIRE_INIT();
  VDB_ARG(a && out && (b || OP >= EW_TO_MONT), "null pointer");
  if (n == 0) return VDB_OK;
  Context& c = ctx();
  size_t bytes = n * sizeof(u256);
  u256* da = (u256*)scratch_get(0, bytes);
  u256* db = (u256*)scratch_get(1, bytes);
  u256* dout = (u256*)scratch_get(2, bytes);
  if (!da || !db || !dout) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c.stream));
  if (b) VDB_HIP(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c.stream));
  hipLaunchKernelGGL(k_elementwise<OP>, dim3(grid_for(n)), dim3(256), 0, c.stream, da, db, dout, n);
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* vdb_version(void) { return "halo2-vectordb_amd 0.1 (gfx950)"; }
const char* vdb_last_error(void) { return g_err; }

static int init_context(int device) {
  Context& c = g_ctxs[device];
  if (c.ready) return VDB_OK;
  VDB_HIP(hipSetDevice(phys_of(device)));
  t_hip_dev = phys_of(device);
  VDB_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  VDB_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
  VDB_HIP(hipEventCreateWithFlags(&c.ev_tail, hipEventDisableTiming));
  VDB_HIP(hipEventCreate(&c.ev0));
  VDB_HIP(hipEventCreate(&c.ev1));
  hipDeviceProp_t prop;
  VDB_HIP(hipGetDeviceProperties(&prop, phys_of(device)));
  c.cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c.device = device;
  c.ready = true;
  return VDB_OK;
}
static void shutdown_context(int device) {
  Context& c = g_ctxs[device];
  if (!c.ready) return;
  if (hipSetDevice(phys_of(device)) == hipSuccess) t_hip_dev = phys_of(device);
  (void)hipStreamSynchronize(c.stream);
  (void)hipStreamSynchronize(c.aux);
  c.msm_pending = false;
  c.msm_counters = nullptr;
  c.msm_out = nullptr;
  for (auto& kv : c.twiddles) (void)hipFree(kv.second);
  c.twiddles.clear();
  for (int i = 0; i < Context::N_SCRATCH; i++) {
    if (c.scratch[i]) (void)hipFree(c.scratch[i]);
    c.scratch[i] = nullptr;
    c.scratch_bytes[i] = 0;
  }
  if (c.msm_out_buf) (void)hipFree(c.msm_out_buf);
  c.msm_out_buf = nullptr;
  c.msm_out_bytes = 0;
  witness_release(c);
  poseidon_release(c);
  c.ntt_lds_raised = false;
  c.win[0] = 0, c.win[1] = ~0ull, c.win[2] = 0, c.win[3] = ~0ull;
  (void)hipEventDestroy(c.ev0);
  (void)hipEventDestroy(c.ev1);
  (void)hipEventDestroy(c.ev_tail);
  (void)hipStreamDestroy(c.aux);
  (void)hipStreamDestroy(c.stream);
  c.ready = false;
  c.device = -1;
}
static int visible_devices(int* n) {
  *n = 0;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess || *n == 0) {
    set_error("no HIP device visible (hipGetDeviceCount: %s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    return VDB_ERR_NO_DEVICE;
  }
  if (*n > VDB_MAX_DEVICES) *n = VDB_MAX_DEVICES;
  g_phys_count = *n;
  if (const char* e2 = getenv("VDB_TEST_ALIAS_DEVICES")) {
    const int want = atoi(e2);
    if (want > *n && want <= VDB_MAX_DEVICES) *n = want;
  }
  return VDB_OK;
}

int vdb_init(int device) {
  int n = 0;
  if (int rc = visible_devices(&n)) return rc;
  VDB_ARG(device >= 0 && device < n, "device index out of range");
  // one process, one GPU: whatever else was bound is released
  for (int d = 0; d < VDB_MAX_DEVICES; d++)
    if (d != device) shutdown_context(d);
  if (int rc = init_context(device)) return rc;
  g_default = device;
  t_cur = device;
  return bind_thread();
}
int vdb_init_devices(int n_devices) {
  int n = 0;
  if (int rc = visible_devices(&n)) return rc;
  VDB_ARG(n_devices >= 1 && n_devices <= n, "n_devices must be between 1 and the number of visible devices");
  for (int d = n_devices; d < VDB_MAX_DEVICES; d++) shutdown_context(d);
  for (int d = 0; d < n_devices; d++)
    if (int rc = init_context(d)) return rc;
  g_default = 0;
  t_cur = 0;
  return bind_thread();
}
int vdb_devices_bound(void) {
  int n = 0;
  for (int d = 0; d < VDB_MAX_DEVICES; d++) n += g_ctxs[d].ready ? 1 : 0;
  return n;
}
int vdb_set_device(int device) {
  if (device < 0 || device >= VDB_MAX_DEVICES || !g_ctxs[device].ready) {
    set_error("vdb_set_device: device %d has not been bound (vdb_init / vdb_init_devices)", device);
    return VDB_ERR_ARG;
  }
  t_cur = device;
  return bind_thread();
}
int vdb_current_device(void) { return ctx().ready ? ctx().device : -1; }
void vdb_shutdown(void) {
  for (int d = 0; d < VDB_MAX_DEVICES; d++) shutdown_context(d);
  t_cur = -1;
  g_default = 0;
}
int vdb_malloc(void** dptr, size_t bytes) {
  VDB_REQUIRE_INIT();
  VDB_ARG(dptr, "null pointer");
  VDB_HIP(timed_malloc(dptr, bytes ? bytes : 1));
  return VDB_OK;
}
int vdb_free(void* dptr) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  VDB_HIP(hipFree(dptr));
  return VDB_OK;
}
int vdb_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  return VDB_OK;
}
int vdb_mem_info(size_t* free_bytes, size_t* total_bytes) {
  VDB_REQUIRE_INIT();
  VDB_ARG(free_bytes && total_bytes, "null pointer");
  VDB_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return VDB_OK;
}
int vdb_alloc_stats(double* seconds, uint64_t* bytes, uint64_t* calls, int reset) {
  if (seconds) *seconds = (double)g_alloc_ns.load() * 1e-9;
  if (bytes) *bytes = g_alloc_bytes.load();
  if (calls) *calls = g_alloc_calls.load();
  if (reset) {
    g_alloc_ns = 0;
    g_alloc_bytes = 0;
    g_alloc_calls = 0;
  }
  return VDB_OK;
}
int vdb_msm_set_scratch_cap(size_t bytes) {
  VDB_REQUIRE_INIT();
  ctx().msm_scratch_cap = bytes;
  return VDB_OK;
}
int vdb_scratch_held(size_t* bytes) {
  VDB_REQUIRE_INIT();
  VDB_ARG(bytes, "null pointer");
  Context& c = ctx();
  size_t sum = 0;
  for (int i = 0; i < Context::N_SCRATCH; i++) sum += c.scratch_bytes[i];
  *bytes = sum;
  return VDB_OK;
}
int vdb_scratch_release(void) {
  VDB_REQUIRE_INIT();
  Context& c = ctx();
  VDB_ARG(!c.msm_pending, "a deferred MSM is still open (vdb_msm_batch_end)");
  VDB_HIP(hipStreamSynchronize(c.stream));
  VDB_HIP(hipStreamSynchronize(c.aux));
  for (int i = 0; i < Context::N_SCRATCH; i++) {
    if (c.scratch[i]) VDB_HIP(hipFree(c.scratch[i]));
    c.scratch[i] = nullptr;
    c.scratch_bytes[i] = 0;
  }
  return VDB_OK;
}

int vdb_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream));
  return VDB_OK;
}
int vdb_memset_dev(void* dst, int value, size_t bytes) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipMemsetAsync(dst, value, bytes, ctx().stream));
  return VDB_OK;
}
int vdb_sync(void) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipStreamSynchronize(ctx().stream));
  VDB_HIP(hipStreamSynchronize(ctx().aux));  // the tail of a deferred MSM, if any
  return VDB_OK;
}
int vdb_timer_start(void) {
  VDB_REQUIRE_INIT();
  VDB_HIP(hipEventRecord(ctx().ev0, ctx().stream));
  return VDB_OK;
}
int vdb_timer_stop(float* ms) {
  VDB_REQUIRE_INIT();
  VDB_ARG(ms, "null pointer");
  VDB_HIP(hipEventRecord(ctx().ev1, ctx().stream));
  VDB_HIP(hipEventSynchronize(ctx().ev1));
  VDB_HIP(hipEventElapsedTime(ms, ctx().ev0, ctx().ev1));
  return VDB_OK;
}

int vdb_fr_mul(const vdb_fr* a, const vdb_fr* b, vdb_fr* out, size_t n) { return elementwise_host<EW_MUL>(a, b, out, n); }
int vdb_fr_add(const vdb_fr* a, const vdb_fr* b, vdb_fr* out, size_t n) { return elementwise_host<EW_ADD>(a, b, out, n); }
int vdb_fr_sub(const vdb_fr* a, const vdb_fr* b, vdb_fr* out, size_t n) { return elementwise_host<EW_SUB>(a, b, out, n); }
int vdb_fr_from_canonical(const vdb_fr* in, vdb_fr* out, size_t n) { return elementwise_host<EW_TO_MONT>(in, nullptr, out, n); }
int vdb_fr_to_canonical(const vdb_fr* in, vdb_fr* out, size_t n) { return elementwise_host<EW_FROM_MONT>(in, nullptr, out, n); }

int vdb_fr_batch_invert(const vdb_fr* in, vdb_fr* out, size_t n) {
  VDB_REQUIRE_INIT();
  VDB_ARG(in && out, "null pointer");
  if (n == 0) return VDB_OK;
  Context& c = ctx();
  size_t bytes = n * sizeof(u256);
  u256* da = (u256*)scratch_get(0, bytes);
  u256* dout = (u256*)scratch_get(1, bytes);
  if (!da || !dout) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(da, in, bytes, hipMemcpyHostToDevice, c.stream));
  size_t threads = (n + BINV_CH - 1) / BINV_CH;
  hipLaunchKernelGGL(k_batch_invert, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, da, dout, n);
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

int vdb_fr_from_wide_dev(const uint8_t* wide_dev, size_t n, vdb_fr* out_dev) {
  VDB_REQUIRE_INIT();
  VDB_ARG(wide_dev && out_dev, "null pointer");
  if (n == 0) return VDB_OK;
  hipLaunchKernelGGL(k_from_wide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, reinterpret_cast<const u256*>(wide_dev), as_u256(out_dev), n);
  VDB_LAUNCH_CHECK();
  return VDB_OK;
}

int vdb_bench_fr_mul(size_t threads, size_t iters, double* mul_per_sec) {
  VDB_REQUIRE_INIT();
  VDB_ARG(mul_per_sec && threads >= 256 && threads % 256 == 0 && iters > 0, "threads must be a positive multiple of 256");
  Context& c = ctx();
  u256* sink = (u256*)scratch_get(0, threads * sizeof(u256));
  if (!sink) return VDB_ERR_OOM;
  const bool v32 = getenv("VDB_BENCH_MUL32") != nullptr;
  hipLaunchKernelGGL(k_bench_mul, dim3((unsigned)(threads / 256)), dim3(256), 0, c.stream, sink, (size_t)8);  // warm-up
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipEventRecord(c.ev0, c.stream));
  if (v32) hipLaunchKernelGGL(k_bench_mul32, dim3((unsigned)(threads / 256)), dim3(256), 0, c.stream, sink, iters);
  else hipLaunchKernelGGL(k_bench_mul, dim3((unsigned)(threads / 256)), dim3(256), 0, c.stream, sink, iters);
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipEventRecord(c.ev1, c.stream));
  VDB_HIP(hipEventSynchronize(c.ev1));
  float ms = 0;
  VDB_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
  *mul_per_sec = (double)threads * (double)iters * 4.0 / ((double)ms * 1e-3);
  return VDB_OK;
}

static void prof_drop_pending() {
  for (auto& pp : g_prof_pending) {
    (void)hipEventDestroy(pp.e0);
    (void)hipEventDestroy(pp.e1);
  }
  g_prof_pending.clear();
}
int vdb_profile_begin(void) {
  VDB_REQUIRE_INIT();
  g_prof.clear();
  prof_drop_pending();
  g_prof_mode = 1;
  return VDB_OK;
}
int vdb_profile_begin_deferred(void) {
  VDB_REQUIRE_INIT();
  g_prof.clear();
  prof_drop_pending();
  g_prof_mode = 2;
  return VDB_OK;
}
int vdb_profile_end(char* json_out, size_t cap) {
  const int mode = g_prof_mode;
  g_prof_mode = 0;
  if (mode == 2) {
    // every recorded launch has to have finished before its events can be read
    (void)hipStreamSynchronize(ctx().stream);
    (void)hipStreamSynchronize(ctx().aux);
    for (auto& pp : g_prof_pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pp.e0, pp.e1) == hipSuccess) {
        ProfEntry& pe = g_prof[pp.name];
        pe.ms += ms;
        pe.launches += 1;
      }
    }
  }
  prof_drop_pending();
  VDB_ARG(json_out && cap > 2, "null buffer");
  std::string js = "{";
  bool first = true;
  for (auto& kv : g_prof) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s\"%s\": {\"ms\": %.6f, \"launches\": %llu}", first ? "" : ", ", kv.first.c_str(), kv.second.ms,
             (unsigned long long)kv.second.launches);
    js += buf;
    first = false;
  }
  js += "}";
  VDB_ARG(js.size() + 1 <= cap, "profile buffer too small");
  memcpy(json_out, js.c_str(), js.size() + 1);
  return VDB_OK;
}

int vdb_fr_root_of_unity(uint32_t k, vdb_fr* out) {
  VDB_ARG(out && k <= 28, "k must be <= 28");
  u256 w = host_root_of_unity(k);
  memcpy(out, &w, 32);
  return VDB_OK;
}

}  // extern "C"
