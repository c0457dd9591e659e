// C++ host-side mirror of the reference's Rust gadget interface, above the C ABI (include/vdb.h).
//
// The reference is compiled Rust and no Rust toolchain exists in this image, so the host layer that
// a Rust user would call is restated in C++ with the same names, argument meaning and error behaviour:
//   FixedPointChip<PRECISION_BITS>  /root/reference/src/gadget/fixed_point.rs:42-136, fixed_point_vec.rs:29-51
//   DistanceChip                    /root/reference/src/gadget/distance.rs:16-196
//   VectorDBChip                    /root/reference/src/gadget/vectordb.rs:13-362
//   Context / AssignedValue         halo2-base (witness_gen_only mode: values + flat advice stream)
// The reference takes the distance as a closure `&dyn Fn(ctx, a, b) -> AssignedValue`; every closure it
// ever passes is one of DistanceChip's methods (examples/*.rs, tests/vectordb/mod.rs:109), so the mirror
// takes a `Metric`.  Reference panics become `vdbhost::Error` (thrown on any non-zero ABI return).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vdb.h"

namespace vdbhost {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* msg) : std::runtime_error(std::string("vdb error ") + std::to_string(c) + ": " + msg), code(c) {}
};
inline void check(int rc) {
  if (rc != 0) throw Error(rc, vdb_last_error());
}

using F = vdb_fr;
enum class Metric : int { Euclidean = 0, Cosine = 1, Manhattan = 2, Hamming = 3 };

// halo2-base AssignedValue in witness_gen_only mode: the value; `cell` is the stream index when known
struct AssignedValue {
  F value{};
  uint64_t cell = UINT64_MAX;
};

// halo2-base Context restated: flat advice stream + cells_to_lookup queue
class Context {
 public:
  std::vector<F> advice;
  std::vector<F> cells_to_lookup;
  bool zero_cached = false;  // Context::load_zero caches one cell

  std::vector<AssignedValue> assign_witnesses(const std::vector<F>& witnesses) {
    std::vector<AssignedValue> out(witnesses.size());
    for (size_t i = 0; i < witnesses.size(); i++) {
      out[i].value = witnesses[i];
      out[i].cell = advice.size();
      advice.push_back(witnesses[i]);
    }
    return out;
  }
  AssignedValue load_constant(const F& c) {
    AssignedValue v{c, advice.size()};
    advice.push_back(c);
    return v;
  }
  // reserve room for a gadget's cells and return pointers to fill
  std::pair<F*, F*> grow(uint64_t cells, uint64_t lookups) {
    size_t a0 = advice.size(), l0 = cells_to_lookup.size();
    advice.resize(a0 + cells);
    cells_to_lookup.resize(l0 + lookups);
    return {advice.data() + a0, cells_to_lookup.data() + l0};
  }
};

inline std::vector<F> values_of(const std::vector<AssignedValue>& v) {
  std::vector<F> out(v.size());
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i].value;
  return out;
}
inline std::vector<F> flatten(const std::vector<std::vector<AssignedValue>>& vs) {
  std::vector<F> out;
  for (auto& v : vs)
    for (auto& c : v) out.push_back(c.value);
  return out;
}

template <uint32_t PRECISION_BITS>
class FixedPointChip {
 public:
  size_t lookup_bits;
  explicit FixedPointChip(size_t lookup_bits_) : lookup_bits(lookup_bits_) {
    static_assert(PRECISION_BITS >= 32 && PRECISION_BITS <= 63, "support only 32 <= precision bits <= 63");
  }
  static FixedPointChip default_(size_t lookup_bits) { return FixedPointChip(lookup_bits); }

  F quantization(double x) const {
    F out;
    check(vdb_fp_quantize(PRECISION_BITS, &x, &out, 1));
    return out;
  }
  double dequantization(const F& x) const {
    double out;
    check(vdb_fp_dequantize(PRECISION_BITS, &x, &out, 1));
    return out;
  }
  // FixedPointInstructions (src/gadget/fixed_point.rs:213-460), one call each: the cells go into the context in the order the Rust
  // gadget pushes them (vdb_wit_fp_op; op numbers: include/vdb.h)
  AssignedValue op(Context& ctx, int code, const AssignedValue& a, const AssignedValue* b = nullptr) const {
    uint64_t cells = 0, lookups = 0;
    const uint32_t L = (uint32_t)lookup_bits;
    check(vdb_wit_fp_op_size(code, PRECISION_BITS, L, 1, &cells, &lookups));
    auto [stream, lookup] = ctx.grow(cells, lookups);
    AssignedValue out;
    check(vdb_wit_fp_op(code, PRECISION_BITS, L, &a.value, b ? &b->value : nullptr, 1, stream, lookup, nullptr, &out.value));
    return out;
  }
  AssignedValue qadd(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 0, a, &b); }
  AssignedValue qsub(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 1, a, &b); }
  AssignedValue qmul(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 2, a, &b); }
  AssignedValue qdiv(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 3, a, &b); }
  AssignedValue neg(Context& ctx, const AssignedValue& a) const { return op(ctx, 4, a); }
  AssignedValue qabs(Context& ctx, const AssignedValue& a) const { return op(ctx, 5, a); }
  AssignedValue is_neg(Context& ctx, const AssignedValue& a) const { return op(ctx, 6, a); }
  AssignedValue qmin(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 7, a, &b); }
  AssignedValue qsqrt(Context& ctx, const AssignedValue& a) const { return op(ctx, 8, a); }
  AssignedValue qlog2(Context& ctx, const AssignedValue& a) const { return op(ctx, 9, a); }
  AssignedValue qexp2(Context& ctx, const AssignedValue& a) const { return op(ctx, 10, a); }
  AssignedValue qlog(Context& ctx, const AssignedValue& a) const { return op(ctx, 11, a); }
  AssignedValue qexp(Context& ctx, const AssignedValue& a) const { return op(ctx, 12, a); }
  AssignedValue qpow(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 13, a, &b); }
  AssignedValue bit_xor(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 14, a, &b); }
  AssignedValue cond_neg(Context& ctx, const AssignedValue& a, const AssignedValue& flag) const { return op(ctx, 15, a, &flag); }
  AssignedValue signed_div_scale(Context& ctx, const AssignedValue& a) const { return op(ctx, 16, a); }
  AssignedValue qmax(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 17, a, &b); }
  AssignedValue sign(Context& ctx, const AssignedValue& a) const { return op(ctx, 18, a); }
  AssignedValue clip(Context& ctx, const AssignedValue& a) const { return op(ctx, 19, a); }
  AssignedValue qmod(Context& ctx, const AssignedValue& a, const AssignedValue& b) const { return op(ctx, 20, a, &b); }
  AssignedValue qsin(Context& ctx, const AssignedValue& a) const { return op(ctx, 21, a); }
  AssignedValue qcos(Context& ctx, const AssignedValue& a) const { return op(ctx, 22, a); }
  AssignedValue qtan(Context& ctx, const AssignedValue& a) const { return op(ctx, 23, a); }
  AssignedValue qsinh(Context& ctx, const AssignedValue& a) const { return op(ctx, 24, a); }
  AssignedValue qcosh(Context& ctx, const AssignedValue& a) const { return op(ctx, 25, a); }
  AssignedValue qtanh(Context& ctx, const AssignedValue& a) const { return op(ctx, 26, a); }
  // FixedPointVectorInstructions
  std::vector<F> quantize_vector(const std::vector<double>& v) const {
    std::vector<F> out(v.size());
    if (!v.empty()) check(vdb_fp_quantize(PRECISION_BITS, v.data(), out.data(), v.size()));
    return out;
  }
  std::vector<double> dequantize_vector(const std::vector<AssignedValue>& v) const {
    std::vector<F> vals = values_of(v);
    std::vector<double> out(v.size());
    if (!v.empty()) check(vdb_fp_dequantize(PRECISION_BITS, vals.data(), out.data(), v.size()));
    return out;
  }
  std::vector<AssignedValue> quantize_and_assign_vector(Context& ctx, const std::vector<double>& v) const {
    return ctx.assign_witnesses(quantize_vector(v));
  }
};

template <uint32_t PRECISION_BITS>
class DistanceChip {
 public:
  const FixedPointChip<PRECISION_BITS>& fixed_point_gate;
  explicit DistanceChip(const FixedPointChip<PRECISION_BITS>& fp) : fixed_point_gate(fp) {}

  AssignedValue distance(Context& ctx, Metric m, const std::vector<AssignedValue>& a, const std::vector<AssignedValue>& b) const {
    if (a.size() != b.size()) throw Error(VDB_ERR_DOMAIN, "assert_eq!(a.len(), b.len()) failed (distance.rs:106)");
    uint64_t cells = 0, lookups = 0;
    const uint32_t L = (uint32_t)fixed_point_gate.lookup_bits;
    check(vdb_wit_distance_size((int)m, PRECISION_BITS, L, 1, a.size(), &cells, &lookups));
    std::vector<F> av = values_of(a), bv = values_of(b);
    auto [stream, lookup] = ctx.grow(cells, lookups);
    AssignedValue out;
    check(vdb_wit_distance((int)m, PRECISION_BITS, L, av.data(), bv.data(), 1, a.size(), stream, lookup, nullptr, &out.value));
    return out;
  }
  AssignedValue euclidean_distance(Context& ctx, const std::vector<AssignedValue>& a, const std::vector<AssignedValue>& b) const {
    return distance(ctx, Metric::Euclidean, a, b);
  }
  AssignedValue cosine_distance(Context& ctx, const std::vector<AssignedValue>& a, const std::vector<AssignedValue>& b) const {
    return distance(ctx, Metric::Cosine, a, b);
  }
  AssignedValue manhattan_distance(Context& ctx, const std::vector<AssignedValue>& a, const std::vector<AssignedValue>& b) const {
    return distance(ctx, Metric::Manhattan, a, b);
  }
  // one minus the share of equal elements (distance.rs:146-175)
  AssignedValue hamming_distance(Context& ctx, const std::vector<AssignedValue>& a, const std::vector<AssignedValue>& b) const {
    return distance(ctx, Metric::Hamming, a, b);
  }
};

// poseidon::PoseidonChip<F, 3, 2>::new(ctx, 8, 57): three load_constant cells [2^64, 0, 0]
class PoseidonChip {
 public:
  static constexpr size_t T = 3, RATE = 2, R_F = 8, R_P = 57;
  static PoseidonChip create(Context& ctx) {
    vdb_fr cap_c{{0, 1, 0, 0}}, cap_m, zero{{0, 0, 0, 0}};
    check(vdb_fr_from_canonical(&cap_c, &cap_m, 1));
    ctx.load_constant(cap_m);
    ctx.load_constant(zero);
    ctx.load_constant(zero);
    return PoseidonChip();
  }
};

template <uint32_t PRECISION_BITS>
class VectorDBChip {
 public:
  const FixedPointChip<PRECISION_BITS>& fixed_point_gate;
  explicit VectorDBChip(const FixedPointChip<PRECISION_BITS>& fp) : fixed_point_gate(fp) {}

  // (indicator, vector): indicator holds raw 0/1 field bits (vectordb.rs:146-149)
  std::pair<std::vector<AssignedValue>, std::vector<AssignedValue>> nearest_vector(Context& ctx, const std::vector<AssignedValue>& query,
                                                                                   const std::vector<std::vector<AssignedValue>>& vectors,
                                                                                   Metric m) const {
    const size_t n = vectors.size(), dim = query.size();
    const uint32_t L = (uint32_t)fixed_point_gate.lookup_bits;
    uint64_t cells = 0, lookups = 0;
    check(vdb_wit_nearest_size((int)m, PRECISION_BITS, L, n, dim, &cells, &lookups));
    std::vector<F> q = values_of(query), db = flatten(vectors), ind(n), res(dim);
    auto [stream, lookup] = ctx.grow(cells, lookups);
    check(vdb_wit_nearest((int)m, PRECISION_BITS, L, q.data(), db.data(), n, dim, stream, lookup, nullptr, ind.data(), res.data()));
    std::vector<AssignedValue> indicator(n), result(dim);
    for (size_t i = 0; i < n; i++) indicator[i].value = ind[i];
    for (size_t i = 0; i < dim; i++) result[i].value = res[i];
    return {indicator, result};
  }

  AssignedValue merkle_commitment(Context& ctx, PoseidonChip&, const std::vector<std::vector<AssignedValue>>& vectors) const {
    const size_t n = vectors.size(), dim = vectors.empty() ? 0 : vectors[0].size();
    uint64_t cells = 0;
    check(vdb_wit_merkle_size(n, dim, ctx.zero_cached ? 1 : 0, &cells));
    std::vector<F> db = flatten(vectors);
    auto [stream, lookup] = ctx.grow(cells, 0);
    (void)lookup;
    AssignedValue root;
    check(vdb_wit_merkle(db.data(), n, dim, ctx.zero_cached ? 1 : 0, stream, nullptr, &root.value));
    if ((n & (n - 1)) != 0) ctx.zero_cached = true;  // padding leaves called load_zero
    return root;
  }

  // (centroids, cluster_indicators): indicators are quantized 1.0 / 0 (vectordb.rs:277-283)
  template <size_t K, size_t I>
  std::pair<std::array<std::vector<AssignedValue>, K>, std::vector<std::array<AssignedValue, K>>> kmeans(
      Context& ctx, const std::vector<std::vector<AssignedValue>>& vectors, Metric m) const {
    const size_t n = vectors.size(), dim = vectors.empty() ? 0 : vectors[0].size();
    const uint32_t L = (uint32_t)fixed_point_gate.lookup_bits;
    uint64_t cells = 0, lookups = 0;
    check(vdb_wit_kmeans_size((int)m, PRECISION_BITS, L, n, dim, K, I, ctx.zero_cached ? 1 : 0, &cells, &lookups));  // assert!(K < vectors.len())
    std::vector<F> db = flatten(vectors), cent(K * dim), ind(n * K);
    auto [stream, lookup] = ctx.grow(cells, lookups);
    check(vdb_wit_kmeans((int)m, PRECISION_BITS, L, db.data(), n, dim, K, I, ctx.zero_cached ? 1 : 0, stream, lookup, nullptr, cent.data(),
                         ind.data()));
    ctx.zero_cached = true;
    std::array<std::vector<AssignedValue>, K> centroids;
    for (size_t k = 0; k < K; k++) {
      centroids[k].resize(dim);
      for (size_t j = 0; j < dim; j++) centroids[k][j].value = cent[k * dim + j];
    }
    std::vector<std::array<AssignedValue, K>> indicators(n);
    for (size_t v = 0; v < n; v++)
      for (size_t k = 0; k < K; k++) indicators[v][k].value = ind[v * K + k];
    return {centroids, indicators};
  }
};

}  // namespace vdbhost
