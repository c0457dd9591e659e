// examples/fixed_point.rs (/root/reference/examples/fixed_point.rs:38-112) in compiled code over the C ABI: FixedPointChip<32> on one
// value x — load_witness(x), qexp2(x), qlog2(x) when x > 0, qsin(x), each made public — with the native f64 values and the errors the
// Rust example prints.  The chip's calls go through host/vectordb.hpp's mirror of FixedPointInstructions (vdb_wit_fp_op: the GPU path).
// Usage: example_fixed_point <lookup_bits> <x>
// Prints: "cells <advice> lookups <lookup> public <n>", then one line per result: name, zk value, native value, absolute error.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "vectordb.hpp"

using namespace vdbhost;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const size_t lookup_bits = (size_t)std::atoi(argv[1]);
  const double x_decimal = std::atof(argv[2]);
  constexpr uint32_t PRECISION_BITS = 32;
  try {
    check(vdb_init(0));
    FixedPointChip<PRECISION_BITS> fixed_point_chip = FixedPointChip<PRECISION_BITS>::default_(lookup_bits);
    Context ctx;
    std::vector<AssignedValue> make_public;
    AssignedValue x = ctx.assign_witnesses({fixed_point_chip.quantization(x_decimal)})[0];       // ctx.load_witness(x)
    make_public.push_back(x);
    struct Line {
      const char* name;
      double zk, native;
    };
    std::vector<Line> lines;
    AssignedValue exp_1 = fixed_point_chip.qexp2(ctx, x);
    lines.push_back({"exp2", fixed_point_chip.dequantization(exp_1.value), std::exp2(x_decimal)});
    make_public.push_back(exp_1);
    if (x_decimal > 0.0) {
      AssignedValue log_2 = fixed_point_chip.qlog2(ctx, x);
      lines.push_back({"log2", fixed_point_chip.dequantization(log_2.value), std::log2(x_decimal)});
      make_public.push_back(log_2);
    }
    AssignedValue sin_x = fixed_point_chip.qsin(ctx, x);
    lines.push_back({"sin", fixed_point_chip.dequantization(sin_x.value), std::sin(x_decimal)});
    make_public.push_back(sin_x);
    std::printf("cells %zu lookups %zu public %zu\n", ctx.advice.size(), ctx.cells_to_lookup.size(), make_public.size());
    for (auto& l : lines) std::printf("%s %.12f %.12f %.3e\n", l.name, l.zk, l.native, std::fabs(l.zk - l.native));
    vdb_shutdown();
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 3;
  }
  return 0;
}
