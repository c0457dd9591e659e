// Fiat–Shamir transcript and proof byte stream — host code only, sequential and tiny by nature.
// Stands in for snark_verifier's PoseidonTranscript<NativeLoader, _> as the reference constructs it
// (/root/reference/src/scaffold/mod.rs:309-310: PoseidonTranscript::<NativeLoader, _>::new::<0>), [UPSTREAM-RECALL]:
//   * Poseidon sponge over BN254 Fr with T = 5, RATE = 4, R_F = 8, R_P = 60 (snark-verifier-sdk's constants), parameters
//     from the Grain LFSR as for the chip; state starts [2^64, 0, 0, 0, 0] and is never reset;
//   * absorbed values are buffered; a squeeze feeds the buffer RATE values at a time — a short chunk gets +1 after its last
//     value, and when the buffer length is a multiple of RATE (empty included) one more permutation absorbs only that +1 —
//     and returns state[1] (the same sponge rule as PoseidonChip);
//   * a G1 point is absorbed as its affine x and y, each reduced into Fr; the identity as (0, 0);
//   * written to the proof: points compressed to 32 bytes (x little-endian, one of the two spare top bits of the last byte =
//     y is odd, the identity all zero), scalars as 32 little-endian bytes of the canonical value.  Which spare bit depends on
//     the halo2curves version as recalled — bit 6 (0x40) from 0.4 on, bit 7 (0x80) in the 0.3.x releases of the reference's
//     dependency era; the default is bit 6, vdb_transcript_set_sign_bit selects the other.
// Parity unpinned: the parameters and encodings above are recalled, the reference holds no proof bytes.
// The sponge is cross-checked against an independent Python restatement (tests/test_transcript_cpu.py), and at T = 3
// against the chip's optimised schedule.
#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "hostperm.hpp"
#include "poseidon.hpp"

using namespace vdb;

struct vdb_transcript {
  int t, rate, r_f, r_p;
  std::vector<u256> state, buf;
  PoseidonOpt opt;  // the permutation's optimised schedule (sparse partial rounds), poseidon.hip
  uint8_t sign_mask = 0x40;  // where a compressed point carries "y is odd" (vdb_transcript_set_sign_bit)
  std::vector<uint8_t> bytes;
  void* ifma = nullptr;      // lane tables of the AVX-512 IFMA permutation (hostperm_ifma.cpp), when the CPU and the width allow
  ~vdb_transcript() { host_ifma_free(ifma); }
};

namespace {

// The sponge's arithmetic lives in hostperm.cpp (four 64-bit limbs; built twice, the BMI2 + ADX build is picked when the CPU
// has both; VDB_HOST_GENERIC=1 forces the portable build, which the tests use to hold the two against each other).
struct HostKernels {
  host_permute_fn permute;
  host_horner_fn horner;
};
const HostKernels& host_kernels() {
  static const HostKernels k = [] {
    const char* force = getenv("VDB_HOST_GENERIC");
    bool fast = !(force && force[0] == '1');
#if defined(__x86_64__)
    fast = fast && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
#else
    fast = false;
#endif
    return fast ? HostKernels{host_permute_mulx, host_horner_mulx} : HostKernels{host_permute_generic, host_horner_generic};
  }();
  return k;
}

HostPermView perm_view(const vdb_transcript* tr) {
  const PoseidonOpt& o = tr->opt;
  static_assert(sizeof(u256) == 32, "field elements are 32 bytes");
  auto w = [](const std::vector<u256>& v) { return reinterpret_cast<const uint64_t*>(v.data()); };
  return HostPermView{o.t, o.half, o.rp, w(o.start), w(o.partial), w(o.end), w(o.mds), w(o.pre_sparse), w(o.sparse_row), w(o.sparse_col)};
}
// VDB_HOST_GENERIC: 1 = the portable build, 2 = at most the mulx / adx build (no AVX-512 IFMA), 3 = the AVX-512 IFMA build whenever
// the CPU has it (no measurement: the tests use it to make sure that build is the one exercised); unset: the best the CPU has
bool ifma_allowed() {
  static const bool ok = [] {
    const char* force = getenv("VDB_HOST_GENERIC");
    if (force && (force[0] == '1' || force[0] == '2')) return false;
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") != 0;
#else
    return false;
#endif
  }();
  return ok;
}
// Which build of the permutation is faster on this CPU is measured once per process and width (a few hundred microseconds): the
// vector build wins on cores with full-width 512-bit units (EPYC 9005, Xeon), the mulx build where they are double-pumped.  Both
// compute the same function, so the choice never shows in a result.
bool ifma_is_faster(vdb_transcript* tr) {
  static std::map<int, bool> decided;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (const char* force = getenv("VDB_HOST_GENERIC"))
    if (force[0] == '3') return true;
  auto it = decided.find(tr->t);
  if (it != decided.end()) return it->second;
  const HostPermView view = perm_view(tr);
  std::vector<u256> scratch(tr->state);
  uint64_t* st = reinterpret_cast<uint64_t*>(scratch.data());
  auto time_of = [&](bool ifma) {
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 24; i++) {
        if (ifma) host_permute_ifma(tr->ifma, st);
        else host_kernels().permute(view, st);
      }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt < best) best = dt;
    }
    return best;
  };
  time_of(true);   // warm both
  time_of(false);
  const bool faster = time_of(true) < time_of(false);
  decided[tr->t] = faster;
  return faster;
}
void permute(vdb_transcript* tr) {
  if (tr->ifma) {
    host_permute_ifma(tr->ifma, reinterpret_cast<uint64_t*>(tr->state.data()));
    return;
  }
  host_kernels().permute(perm_view(tr), reinterpret_cast<uint64_t*>(tr->state.data()));
}

void absorb_chunk(vdb_transcript* tr, const u256* in, int n_in) {
  for (int i = 0; i < n_in; i++) tr->state[1 + i] = fr_add(tr->state[1 + i], in[i]);
  if (n_in < tr->rate) tr->state[1 + n_in] = fr_add(tr->state[1 + n_in], mont_one<Fr>());
  permute(tr);
}

// canonical integer of an Fq element (Montgomery in memory) reduced into Fr, Montgomery
u256 fq_to_fr(const u256& a_mont) {
  u256 c = from_mont<Fq>(a_mont);
  const u256 r = mod_p<Fr>();
  while (u256_geq(c, r)) {
    u256 d;
    u256_sub(d, c, r);
    c = d;
  }
  return to_mont<Fr>(c);
}

void put_le(std::vector<uint8_t>& out, const u256& canonical, uint8_t top_flags) {
  for (int w = 0; w < 8; w++)
    for (int b = 0; b < 4; b++) out.push_back((uint8_t)(canonical.w[w] >> (8 * b)));
  out.back() |= top_flags;
}

}  // namespace

extern "C" {

// acc <- acc x + v_i for i = 0 .. n - 1 on the host (the multi-open combines tens of thousands of evaluations with powers of a
// challenge before anything goes back to the device; Python integers cost microseconds apiece)
int vdb_fr_horner(const vdb_fr* values, size_t n, const vdb_fr* x, vdb_fr* acc) {
  VDB_ARG((values || n == 0) && x && acc, "null pointer");
  host_kernels().horner(reinterpret_cast<const uint64_t*>(values), n, reinterpret_cast<const uint64_t*>(x), reinterpret_cast<uint64_t*>(acc));
  return VDB_OK;
}

int vdb_transcript_new(uint32_t t, uint32_t r_f, uint32_t r_p, vdb_transcript** out) {
  VDB_ARG(out && t >= 2 && t <= 16 && r_f >= 2 && r_f % 2 == 0 && r_f <= 64 && r_p <= 256, "bad argument");
  vdb_transcript* tr = new (std::nothrow) vdb_transcript();
  if (!tr) return VDB_ERR_OOM;
  tr->t = (int)t;
  tr->rate = (int)t - 1;
  tr->r_f = (int)r_f;
  tr->r_p = (int)r_p;
  try {
    poseidon_build_opt(tr->t, tr->r_f, tr->r_p, &tr->opt);
    tr->state.assign(t, u256_zero());
    if (ifma_allowed()) {
      tr->ifma = host_ifma_prepare(perm_view(tr));     // null (width above 8, no memory): the scalar builds serve
      if (tr->ifma && !ifma_is_faster(tr)) {            // e.g. a core that double-pumps 512-bit operations
        host_ifma_free(tr->ifma);
        tr->ifma = nullptr;
      }
    }
  } catch (...) {
    delete tr;
    return VDB_ERR_OOM;
  }
  u256 cap = u256_zero();
  cap.w[2] = 1;  // 2^64
  tr->state[0] = to_mont<Fr>(cap);
  *out = tr;
  return VDB_OK;
}

void vdb_transcript_free(vdb_transcript* tr) { delete tr; }

int vdb_transcript_set_sign_bit(vdb_transcript* tr, uint32_t bit) {
  VDB_ARG(tr && (bit == 6 || bit == 7), "the y-parity flag of a compressed point sits in bit 6 or bit 7 of the last byte");
  tr->sign_mask = (uint8_t)(1u << bit);
  return VDB_OK;
}

int vdb_transcript_common_scalar(vdb_transcript* tr, const vdb_fr* s) {
  VDB_ARG(tr && s, "null pointer");
  u256 v;
  memcpy(&v, s, 32);
  try {
    tr->buf.push_back(v);
  } catch (...) {  // nothing may propagate across the C ABI
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_common_point(vdb_transcript* tr, const vdb_g1* p) {
  VDB_ARG(tr && p, "null pointer");
  u256 xy[2];
  memcpy(xy, p, 64);
  try {
    tr->buf.push_back(fq_to_fr(xy[0]));
    tr->buf.push_back(fq_to_fr(xy[1]));
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_scalar(vdb_transcript* tr, const vdb_fr* s) {
  int rc = vdb_transcript_common_scalar(tr, s);
  if (rc) return rc;
  u256 v;
  memcpy(&v, s, 32);
  try {
    put_le(tr->bytes, from_mont<Fr>(v), 0);
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_point(vdb_transcript* tr, const vdb_g1* p) {
  int rc = vdb_transcript_common_point(tr, p);
  if (rc) return rc;
  u256 xy[2];
  memcpy(xy, p, 64);
  const u256 x = from_mont<Fq>(xy[0]), y = from_mont<Fq>(xy[1]);
  const bool identity = u256_is_zero(x) && u256_is_zero(y);
  try {
    put_le(tr->bytes, x, (!identity && (y.w[0] & 1)) ? tr->sign_mask : 0);
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_points(vdb_transcript* tr, const vdb_g1* p, size_t n) {
  VDB_ARG(tr && (p || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_write_point(tr, p + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_write_scalars(vdb_transcript* tr, const vdb_fr* s, size_t n) {
  VDB_ARG(tr && (s || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_write_scalar(tr, s + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_common_points(vdb_transcript* tr, const vdb_g1* p, size_t n) {
  VDB_ARG(tr && (p || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_common_point(tr, p + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_common_scalars(vdb_transcript* tr, const vdb_fr* s, size_t n) {
  VDB_ARG(tr && (s || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_common_scalar(tr, s + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_squeeze(vdb_transcript* tr, vdb_fr* out) {
  VDB_ARG(tr && out, "null pointer");
  const size_t n = tr->buf.size();
  for (size_t i = 0; i < n; i += tr->rate) absorb_chunk(tr, tr->buf.data() + i, (int)(n - i < (size_t)tr->rate ? n - i : (size_t)tr->rate));
  if (n % tr->rate == 0) absorb_chunk(tr, nullptr, 0);
  tr->buf.clear();
  memcpy(out, &tr->state[1], 32);
  return VDB_OK;
}

// Absorbs the complete RATE-sized chunks of what has been written so far (the sponge takes its input in order, so absorbing them
// now or at the next squeeze gives the same state: what is left, fewer than RATE values or nothing, is framed by the squeeze as
// before).  A caller interleaves host absorption with device work this way.
int vdb_transcript_flush(vdb_transcript* tr) {
  VDB_ARG(tr, "null pointer");
  const size_t n = tr->buf.size(), full = n - n % (size_t)tr->rate;
  for (size_t i = 0; i < full; i += tr->rate) absorb_chunk(tr, tr->buf.data() + i, tr->rate);
  tr->buf.erase(tr->buf.begin(), tr->buf.begin() + (std::ptrdiff_t)full);
  return VDB_OK;
}

int vdb_transcript_proof_len(const vdb_transcript* tr, size_t* len) {
  VDB_ARG(tr && len, "null pointer");
  *len = tr->bytes.size();
  return VDB_OK;
}

int vdb_transcript_proof_bytes(const vdb_transcript* tr, uint8_t* out, size_t cap) {
  VDB_ARG(tr && (out || tr->bytes.empty()), "null pointer");
  VDB_ARG(cap >= tr->bytes.size(), "buffer too small");
  if (!tr->bytes.empty()) memcpy(out, tr->bytes.data(), tr->bytes.size());
  return VDB_OK;
}

}  // extern "C"
