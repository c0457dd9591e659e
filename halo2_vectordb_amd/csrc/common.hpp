// Shared host-side plumbing of libvdb_hip: one context per bound GPU (vdb_init binds one, vdb_init_devices several; a host
// thread works on the device it selected with vdb_set_device), error reporting across the C ABI, scratch allocation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vdb.h"
#include "field.hpp"

namespace vdb {

struct Context {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  // second stream for the latency-bound tail of a deferred MSM (vdb_msm_batch_masked_dev_begin/_end): its bucket
  // folding runs beside whatever the caller queues on the main stream next (the NTTs)
  hipStream_t aux = nullptr;
  hipEvent_t ev_tail = nullptr;
  bool msm_pending = false;       // a deferred MSM has not been collected yet
  const void* msm_counters = nullptr;
  const void* msm_out = nullptr;  // device buffer the deferred MSM writes its points to
  void* msm_out_buf = nullptr;    // its allocation (grow-only; not one of the shared scratch slots)
  size_t msm_out_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int cu_count = 256;
  // twiddle tables keyed by (log_n, first limb words of omega): tw[e] = omega^e, e < n
  struct TwKey {
    uint32_t log_n;
    u256 omega;
    bool operator<(const TwKey& o) const {
      if (log_n != o.log_n) return log_n < o.log_n;
      return memcmp(omega.w, o.omega.w, 32) < 0;
    }
  };
  std::map<TwKey, u256*> twiddles;
  // grow-only scratch buffers
  static constexpr int N_SCRATCH = 8;
  void* scratch[N_SCRATCH] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[N_SCRATCH] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t msm_scratch_cap = 0;   // vdb_msm_set_scratch_cap: upper bound of the MSM's work space (0: half of the free HBM, 8 .. 96 GiB)
  // device-resident caches owned by other translation units; released through their hooks in vdb_shutdown
  std::map<uint64_t, void*> fp_tables;   // witness.hip: FixedPointChip tables keyed by (P, L)
  void* poseidon_spec = nullptr;          // poseidon.hip: device copy of the Poseidon spec
  bool ntt_lds_raised = false;            // ntt.hip: hipFuncAttributeMaxDynamicSharedMemorySize raised on this device
  uint64_t win[4] = {0, ~0ull, 0, ~0ull}; // witness.hip: rank window of the *_dev witness entry points (vdb_wit_set_window)
};

#define VDB_MAX_DEVICES 16
// the context of the device the calling thread works on: the one it chose with vdb_set_device, else the process default
// (the device of the last vdb_init, device 0 after vdb_init_devices)
Context& ctx();
// makes that device the calling thread's current HIP device (HIP's current device is per thread)
int bind_thread();
// physical HIP device of a logical one (identical unless the test-only VDB_TEST_ALIAS_DEVICES is set)
int phys_of(int logical);
bool context_ready(int device);
void witness_release(Context& c);   // witness.hip
void poseidon_release(Context& c);  // poseidon.hip
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// returns nullptr (and sets error) on failure
void* scratch_get(int slot, size_t bytes);
// hipMalloc with its wall time and size added to the process's allocation statistics (vdb_alloc_stats): on this driver an allocation
// is instant or costs seconds — fresh HBM is mapped (and HBM another process or an earlier free left behind cleared) at ~30 ms / GiB
hipError_t timed_malloc(void** p, size_t bytes);

#define VDB_HIP(expr)                                                      \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) return vdb::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define VDB_REQUIRE_INIT()                                            \
  do {                                                                \
    if (!vdb::ctx().ready) {                                          \
      vdb::set_error("vdb_init() has not been called (or failed)"); \
      return VDB_ERR_NOT_INIT;                                        \
    }                                                                 \
    if (int _rc = vdb::bind_thread()) return _rc;                     \
  } while (0)

#define VDB_ARG(cond, msg)       \
  do {                           \
    if (!(cond)) {               \
      vdb::set_error("%s", msg); \
      return VDB_ERR_ARG;        \
    }                            \
  } while (0)

#define VDB_LAUNCH_CHECK() VDB_HIP(hipGetLastError())

// A column that has not been materialised: rows [0, len) are the contiguous cells src[0 .. len) of a witness stream,
// the last n_blind rows come from `blind` (when given), everything else is zero.  Same layout as the C ABI's vdb_colsrc.
struct ColSrc {
  const u256* src;
  uint64_t len;
  const u256* blind;
};
__device__ __forceinline__ u256 colsrc_fetch(const ColSrc& cs, uint64_t i, uint64_t n, uint32_t n_blind) {
  if (i < cs.len) return ld256(cs.src + i);
  if (cs.blind && i >= n - n_blind) return ld256(cs.blind + (i - (n - n_blind)));
  return u256_zero();
}
static inline const u256* as_u256(const vdb_fr* p) { return reinterpret_cast<const u256*>(p); }
static inline u256* as_u256(vdb_fr* p) { return reinterpret_cast<u256*>(p); }

// Optional per-kernel timing with HIP events (vdb_profile_begin / _begin_deferred / _end).  Two modes:
//   1 "synchronous": every launch is waited for (kernels run one at a time; the deferred MSM tail runs on the main stream);
//   2 "deferred":    events are recorded around every launch on the stream it goes to and only read in vdb_profile_end,
//                    after the work has drained — the kernels run exactly as in the untimed path (overlap included), which is
//                    what bench.py uses inside its timed region.
struct ProfEntry {
  double ms = 0;
  uint64_t launches = 0;
};
struct ProfPending {
  const char* name;
  hipEvent_t e0, e1;
};
extern int g_prof_mode;
extern std::map<std::string, ProfEntry> g_prof;
extern std::vector<ProfPending> g_prof_pending;
struct ProfScope {
  const char* name;
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  explicit ProfScope(const char* n, hipStream_t s = nullptr) : name(n), st(s ? s : ctx().stream) {
    if (!g_prof_mode) return;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
  }
  ~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, st);
    if (g_prof_mode == 2) {
      g_prof_pending.push_back({name, e0, e1});
      return;
    }
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ProfEntry& pe = g_prof[name];
    pe.ms += ms;
    pe.launches += 1;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
};
#define VDB_PROF(name) vdb::ProfScope _prof_scope_(name)
#define VDB_PROF_ON(name, stream) vdb::ProfScope _prof_scope_(name, stream)

// Fr domain constants computed on the host with the same field code
u256 host_root_of_unity(uint32_t k);  // ROOT_OF_UNITY^(2^(28-k)), Montgomery
u256 host_zeta();                     // halo2curves bn256 Fr::ZETA, Montgomery
u256 host_fr_from_u64(uint64_t v);
u256 host_coset_shift(uint32_t k, uint32_t t);  // g_t = zeta w_{4n}^(bitrev2(t)): the shift of slot t of the extended domain taken coset by coset (ntt.hip)

// internal device-level entry points shared between translation units (all on ctx().stream)
int ntt_dev(u256* data, u256* out_or_null, size_t n_cols, uint32_t log_n, const u256& omega, bool scale_ninv,
            bool coset_in, size_t in_len, const ColSrc* srcs, uint32_t n_blind, bool coset_out, const u256* in_scale = nullptr,
            const u256* const* in_tabs = nullptr, uint32_t vslots = 0);

}  // namespace vdb
