// Host-side field arithmetic of the Fiat–Shamir transcript (transcript.hip): the sponge's permutation and a Horner loop over
// BN254 Fr in Montgomery form (R = 2^256) on four 64-bit limbs.  hostperm.cpp is compiled twice (Makefile): once for any
// x86-64 (128-bit products the compiler lowers as it can), once with BMI2 + ADX (mulx and carry chains: 1.7x); transcript.hip
// picks by __builtin_cpu_supports at first use.  Plain C++, no HIP: the sponge is sequential and stays on the host.
#pragma once
#include <cstddef>
#include <cstdint>

namespace vdb {

// the permutation's optimised schedule (PoseidonOpt of poseidon.hpp) as plain pointers to 4 x u64 field elements
struct HostPermView {
  int t, half, rp;
  const uint64_t *start, *partial, *end, *mds, *pre_sparse, *sparse_row, *sparse_col;
};
typedef void (*host_permute_fn)(const HostPermView&, uint64_t* state /* t x 4 words */);
typedef void (*host_horner_fn)(const uint64_t* values, size_t n, const uint64_t* x, uint64_t* acc);

void host_permute_generic(const HostPermView& o, uint64_t* state);
void host_horner_generic(const uint64_t* values, size_t n, const uint64_t* x, uint64_t* acc);
void host_permute_mulx(const HostPermView& o, uint64_t* state);
void host_horner_mulx(const uint64_t* values, size_t n, const uint64_t* x, uint64_t* acc);

}  // namespace vdb
