// Host-side field arithmetic of the Fiat–Shamir transcript (transcript.hip): the sponge's permutation and a Horner loop over
// BN254 Fr in Montgomery form (R = 2^256) on four 64-bit limbs.  hostperm.cpp is compiled twice (Makefile): once for any
// x86-64 (128-bit products the compiler lowers as it can), once with BMI2 + ADX (mulx and carry chains: 1.7x); hostperm_ifma.cpp is a third
// build of the permutation for AVX-512 IFMA; transcript.hip picks by __builtin_cpu_supports at first use.  Plain C++, no HIP: the sponge is sequential and stays on the host.
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
// AVX-512 IFMA build of the permutation (hostperm_ifma.cpp; widths up to 8): the schedule's constants are converted once into lane
// tables (prepare: null when the width does not fit or memory is short), a permutation then reads only the tables
void* host_ifma_prepare(const HostPermView& o);
void host_ifma_free(void* tables);
void host_permute_ifma(const void* tables, uint64_t* state);

}  // namespace vdb
