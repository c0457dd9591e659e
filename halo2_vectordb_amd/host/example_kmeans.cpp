// Counterpart of /root/reference/examples/kmeans.rs (and of tests/vectordb/mod.rs:93-135 chip_kmeans):
// quantize vectors, assign them, run VectorDBChip::kmeans::<K, I>, print the dequantized centroids.
// Usage: example_kmeans [euclidean|cosine|manhattan] < vectors.txt   (first line: n dim; then n*dim floats)
#include <cstdio>
#include <iostream>

#include "vectordb.hpp"

using namespace vdbhost;

int main(int argc, char** argv) {
  constexpr uint32_t PRECISION_BITS = 48;
  constexpr size_t K = 2, I = 4;
  Metric metric = Metric::Euclidean;
  if (argc > 1 && std::string(argv[1]) == "cosine") metric = Metric::Cosine;
  if (argc > 1 && std::string(argv[1]) == "manhattan") metric = Metric::Manhattan;
  size_t n, dim;
  if (!(std::cin >> n >> dim)) return 2;
  std::vector<std::vector<double>> input(n, std::vector<double>(dim));
  for (auto& v : input)
    for (auto& x : v) std::cin >> x;
  try {
    check(vdb_init(0));
    const size_t lookup_bits = 13;  // LOOKUP_BITS
    auto fixed_point_chip = FixedPointChip<PRECISION_BITS>::default_(lookup_bits);
    VectorDBChip<PRECISION_BITS> vectordb_chip(fixed_point_chip);
    Context ctx;
    std::vector<std::vector<AssignedValue>> vectors;
    for (auto& v : input) vectors.push_back(ctx.assign_witnesses(fixed_point_chip.quantize_vector(v)));
    auto [centroids, indicators] = vectordb_chip.kmeans<K, I>(ctx, vectors, metric);
    std::printf("cells %zu lookups %zu\n", ctx.advice.size(), ctx.cells_to_lookup.size());
    for (auto& c : centroids) {
      for (double x : fixed_point_chip.dequantize_vector(c)) std::printf("%.17g ", x);
      std::printf("\n");
    }
    for (auto& ind : indicators) {
      size_t id = K;
      for (size_t k = 0; k < K; k++)
        if (fixed_point_chip.dequantization(ind[k].value) == 1.0 && id == K) id = k;  // first index that has 1 (tests/vectordb/mod.rs:121-132)
      std::printf("%zu ", id);
    }
    std::printf("\n");
    vdb_shutdown();
  } catch (const Error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
