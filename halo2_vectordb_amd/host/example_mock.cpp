// Counterpart in compiled code of the reference's Mock arm (/root/reference/src/scaffold/mod.rs:263-266:
// create_circuit(Mock) -> MockProver::run(k, &circuit, instances).assert_satisfied()) for the distance examples
// (examples/euclid.rs, examples/distances.rs): quantize the two vectors, generate the gadget's witness on the GPU with the
// keygen flags (gate starts, constants, lookup sources), then check the witness where it lies in HBM — every gate row
// a + b c = d, every lookup cell against the range table 0 .. 2^LOOKUP_BITS - 1, every lookup cell against the advice cell it
// copies, every constant cell against the keygen-time stream — through the C ABI alone (vdb_mock_check_dev).
// The GPUs are bound through the multi-device lifecycle (vdb_init_devices), the thread works on device 0.
// Usage: example_mock <metric: 0 euclidean | 1 cosine | 2 manhattan | 3 hamming> <lookup_bits> [tamper_cell]  < "dim a_0 .. a_{dim-1} b_0 .. b_{dim-1}"
// Prints the cell counts and the report; exit code 0 when the circuit is satisfied, 4 when it is not.
#include <cstdio>
#include <iostream>

#include "vectordb.hpp"

using namespace vdbhost;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int metric = std::atoi(argv[1]);
  const uint32_t L = (uint32_t)std::atoi(argv[2]), P = 48;
  const long tamper = argc > 3 ? std::atol(argv[3]) : -1;
  size_t dim;
  if (!(std::cin >> dim)) return 2;
  std::vector<double> a(dim), b(dim);
  for (auto& x : a) std::cin >> x;
  for (auto& x : b) std::cin >> x;
  try {
    check(vdb_init_devices(1));
    check(vdb_set_device(0));
    std::vector<F> qa(dim), qb(dim);
    check(vdb_fp_quantize(P, a.data(), qa.data(), dim));
    check(vdb_fp_quantize(P, b.data(), qb.data(), dim));
    uint64_t cells = 0, lookups = 0;
    check(vdb_wit_distance_size(metric, P, L, 1, dim, &cells, &lookups));
    std::vector<F> stream(cells), lookup(lookups ? lookups : 1);
    std::vector<uint8_t> flags(cells);
    F result;
    check(vdb_wit_distance(metric, P, L, qa.data(), qb.data(), 1, dim, stream.data(), lookup.data(), flags.data(), &result));
    // cells_to_lookup are copies of advice cells: the kernels mark the sources (flag bit 2) in queue order
    std::vector<int64_t> lookup_src;
    for (uint64_t i = 0; i < cells; i++)
      if (flags[i] & 4) lookup_src.push_back((int64_t)i);
    if (lookup_src.size() != lookups) {
      std::fprintf(stderr, "lookup sources %zu != lookup cells %llu\n", lookup_src.size(), (unsigned long long)lookups);
      return 3;
    }
    std::vector<F> keygen_stream = stream;              // the constants the fixed column would hold
    if (tamper >= 0 && (uint64_t)tamper < cells) stream[tamper].l[0] ^= 1;   // a cheating prover's cell
    void *d_stream, *d_flags, *d_lookup, *d_src, *d_const;
    check(vdb_malloc(&d_stream, cells * 32));
    check(vdb_malloc(&d_flags, cells));
    check(vdb_malloc(&d_lookup, lookup.size() * 32));
    check(vdb_malloc(&d_src, (lookups ? lookups : 1) * 8));
    check(vdb_malloc(&d_const, cells * 32));
    check(vdb_memcpy_h2d(d_stream, stream.data(), cells * 32));
    check(vdb_memcpy_h2d(d_flags, flags.data(), cells));
    check(vdb_memcpy_h2d(d_lookup, lookup.data(), lookup.size() * 32));
    if (lookups) check(vdb_memcpy_h2d(d_src, lookup_src.data(), lookups * 8));
    check(vdb_memcpy_h2d(d_const, keygen_stream.data(), cells * 32));
    vdb_mock_report rep;
    check(vdb_mock_check_dev((const vdb_fr*)d_stream, cells, (const uint8_t*)d_flags, (const vdb_fr*)d_lookup, lookups, L, nullptr,
                             lookups ? (const int64_t*)d_src : nullptr, (const vdb_fr*)d_const, nullptr, nullptr, 0, &rep));
    double dist = 0;
    check(vdb_fp_dequantize(P, &result, &dist, 1));
    std::printf("cells %llu lookups %llu distance %.12f\n", (unsigned long long)cells, (unsigned long long)lookups, dist);
    std::printf("gate_rows_violated %llu lookup_cells_out_of_table %llu lookup_copies_unequal %llu constants_changed %llu first_gate_row %lld\n",
                (unsigned long long)rep.gate_rows_violated, (unsigned long long)rep.lookup_cells_out_of_table, (unsigned long long)rep.lookup_copies_unequal,
                (unsigned long long)rep.constants_changed, rep.gate_rows_violated ? (long long)rep.first_gate_row : -1ll);
    for (void* p : {d_stream, d_flags, d_lookup, d_src, d_const}) vdb_free(p);
    vdb_shutdown();
    return (rep.gate_rows_violated || rep.lookup_cells_out_of_table || rep.lookup_copies_unequal || rep.constants_changed) ? 4 : 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
