// Multi-GPU entry points of the C ABI: for a caller that owns one process and several GPUs (a
// patched halo2-axiom calling into the library from its prover thread).  The columns of one call are cut into contiguous
// blocks, one per bound device (vdb_init_devices); one host thread per device drives the ordinary single-device entry
// point on its block, and every device returns its results D2H (64 B per commitment) — there is no exchange between the
// devices, the path shards by column (the transcript that consumes the commitments lives on the host anyway).
// One process per GPU (bench.py, torch.distributed) remains the other supported arrangement; see INTEGRATION.md.
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

namespace vdb {

// runs fn(device, lo, hi) on one thread per device over contiguous blocks of [0, n_items); first failure wins
template <class Fn>
static int for_each_device(const std::vector<int>& devices, size_t n_items, Fn fn) {
  const size_t nd = devices.size();
  std::vector<int> rcs(nd, VDB_OK);
  std::vector<std::string> errs(nd);
  std::vector<std::thread> threads;
  try {
    for (size_t i = 0; i < nd; i++) {
      const size_t lo = n_items * i / nd, hi = n_items * (i + 1) / nd;
      threads.emplace_back([&, i, lo, hi] {
        int rc = vdb_set_device(devices[i]);
        if (rc == VDB_OK && hi > lo) rc = fn(devices[i], lo, hi);
        rcs[i] = rc;
        if (rc != VDB_OK) errs[i] = vdb_last_error();
      });
    }
  } catch (...) {
    for (auto& t : threads) t.join();
    set_error("could not start a host thread per device");
    return VDB_ERR_OOM;
  }
  for (auto& t : threads) t.join();
  for (size_t i = 0; i < nd; i++)
    if (rcs[i] != VDB_OK) {
      set_error("device %d: %s", devices[i], errs[i].c_str());
      return rcs[i];
    }
  return VDB_OK;
}

static std::vector<int> bound_devices() {
  std::vector<int> d;
  for (int i = 0; i < VDB_MAX_DEVICES; i++)
    if (context_ready(i)) d.push_back(i);
  return d;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_srs_load_all(uint32_t k, const vdb_g1* g, const vdb_g1* g_lagrange, uint32_t window_bits, vdb_srs** out, int n_out) {
  VDB_REQUIRE_INIT();
  const std::vector<int> devs = bound_devices();
  VDB_ARG(out && n_out == (int)devs.size(), "out must have one slot per bound device (vdb_devices_bound)");
  for (int i = 0; i < n_out; i++) out[i] = nullptr;
  // the bases are replicated: every device builds its own window tables from the same host arrays
  int rc = for_each_device(devs, devs.size(), [&](int, size_t lo, size_t) { return vdb_srs_load_window(k, g, g_lagrange, window_bits, &out[lo]); });
  if (rc != VDB_OK)
    for (int i = 0; i < n_out; i++) {
      vdb_srs_free(out[i]);
      out[i] = nullptr;
    }
  return rc;
}

int vdb_msm_batch_multi(vdb_srs* const* srs, int n_srs, int basis, const vdb_fr* const* cols, size_t n_cols, size_t n, vdb_g1* out) {
  VDB_REQUIRE_INIT();
  VDB_ARG(srs && n_srs >= 1 && n_srs <= VDB_MAX_DEVICES && cols && out, "bad argument");
  std::vector<int> devs;
  int dev_of[VDB_MAX_DEVICES];
  for (int i = 0; i < n_srs; i++) {
    int d = -1;
    VDB_ARG(srs[i] && vdb_srs_device(srs[i], &d) == VDB_OK && context_ready(d), "srs handle of a device that is not bound");
    dev_of[i] = d;
    devs.push_back(d);
  }
  return for_each_device(devs, n_cols, [&](int d, size_t lo, size_t hi) {
    int i = 0;
    while (dev_of[i] != d) i++;
    return vdb_msm_batch(srs[i], basis, cols + lo, hi - lo, n, out + lo);
  });
}

int vdb_ntt_batch_multi(vdb_fr* const* cols, size_t n_cols, uint32_t log_n, const vdb_fr* omega, int flags) {
  VDB_REQUIRE_INIT();
  VDB_ARG(cols && omega, "null pointer");
  return for_each_device(bound_devices(), n_cols, [&](int, size_t lo, size_t hi) { return vdb_ntt_batch(cols + lo, hi - lo, log_n, omega, flags); });
}

}  // extern "C"
