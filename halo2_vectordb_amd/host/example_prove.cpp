// Counterpart in compiled code of the reference's Prove arm for the k-means example (/root/reference/src/scaffold/mod.rs:284-298
// run_cli -> create_circuit -> gen_snark_shplonk, up to and including the advice round): quantize, generate the witness on the
// GPU, derive the break points (the Keygen arm pins them, :272), commit every advice / lookup column and transform it
// (lagrange_to_coeff, coeff_to_extended) — the hot path of bench.py, driven through the C ABI alone, device resident, with the
// commit reading the columns straight from the witness stream.
// Usage: example_prove <tau (decimal, < 2^63)> <dump.bin> < vectors.txt   (first line: n dim K I k L; then n*dim floats)
// Writes to dump.bin: u64 n_adv, n_lk, rows; the commitments (n_cols x 64 B); then the Lagrange images of the first advice
// column, the last advice column and the first lookup column (rows x 32 B each) so that a checker can recompute their
// commitments independently (tests/test_gpu_host_cpp.py).
#include <cmath>
#include <cstdio>
#include <iostream>

#include "vectordb.hpp"

using namespace vdbhost;

namespace {
constexpr uint32_t P = 48, MINIMUM_ROWS = 9, N_BLIND = 7;  // src/scaffold/mod.rs:383 (= blinding_factors + 3); halo2: blinding_factors + 1 rows of blinds

struct Dev {  // RAII device allocation
  void* p = nullptr;
  size_t bytes = 0;
  explicit Dev(size_t n) : bytes(n ? n : 32) { check(vdb_malloc(&p, bytes)); }
  ~Dev() {
    if (p) vdb_free(p);
  }
  Dev(const Dev&) = delete;
  template <class T>
  T* as(size_t byte_off = 0) const {
    return reinterpret_cast<T*>(static_cast<char*>(p) + byte_off);
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  size_t n, dim, K, I;
  uint32_t k, L;
  if (!(std::cin >> n >> dim >> K >> I >> k >> L)) return 2;
  std::vector<double> input(n * dim);
  for (auto& x : input) std::cin >> x;
  try {
    check(vdb_init(0));
    const uint64_t rows = 1ull << k;
    // quantize_vector + assign_witnesses: the vectors are the first cells of the stream (examples/kmeans.rs)
    std::vector<F> q(n * dim);
    check(vdb_fp_quantize(P, input.data(), q.data(), q.size()));
    uint64_t cells = 0, lookups = 0;
    check(vdb_wit_kmeans_size((int)Metric::Euclidean, P, L, n, dim, K, I, 0, &cells, &lookups));
    const uint64_t n_in = n * dim, n_cells = n_in + cells;
    Dev d_vec(q.size() * 32), d_stream(n_cells * 32), d_lookup(lookups * 32), d_sel(n_cells), d_cent(K * dim * 32), d_ind(n * K * 32);
    check(vdb_memcpy_h2d(d_vec.p, q.data(), q.size() * 32));
    check(vdb_memset_dev(d_sel.p, 0, n_cells));
    auto witness = [&](bool record_flags) {
      check(vdb_memcpy_d2d(d_stream.p, d_vec.p, n_in * 32));
      check(vdb_wit_kmeans_dev((int)Metric::Euclidean, P, L, d_vec.as<F>(), n, dim, K, I, 0, d_stream.as<F>(n_in * 32), d_lookup.as<F>(),
                               record_flags ? d_sel.as<uint8_t>(n_in) : nullptr, d_cent.as<F>(), d_ind.as<F>()));
    };
    // Keygen arm: a flag-recording run gives the gate starts, from which the break points follow
    witness(true);
    uint64_t n_bp = 0;
    check(vdb_layout_plan_dev(d_sel.as<uint8_t>(), n_cells, k, MINIMUM_ROWS, nullptr, 0, &n_bp));
    std::vector<uint64_t> bp(n_bp ? n_bp : 1);
    check(vdb_layout_plan_dev(d_sel.as<uint8_t>(), n_cells, k, MINIMUM_ROWS, bp.data(), bp.size(), &n_bp));
    const uint64_t n_adv = n_bp + 1, max_rows = rows - MINIMUM_ROWS, n_lk = (lookups + max_rows - 1) / max_rows, n_cols = n_adv + n_lk;
    // gen_srs(k) for a known tau ("unsafe" setup, src/scaffold/mod.rs:260-261)
    F tau_c{}, tau;
    tau_c.l[0] = std::strtoull(argv[1], nullptr, 10);
    check(vdb_fr_from_canonical(&tau_c, &tau, 1));
    std::vector<vdb_g1> g(rows), gl(rows);
    check(vdb_srs_setup_unsafe(k, &tau, g.data(), gl.data()));
    vdb_srs* srs = nullptr;
    check(vdb_srs_load(k, nullptr, gl.data(), &srs));
    // blinding rows (the reference draws them from OsRng; any field elements do): a fixed xorshift stream below 2^253
    std::vector<F> blind(n_cols * N_BLIND);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& b : blind)
      for (int j = 0; j < 4; j++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        b.l[j] = j == 3 ? (x >> 3) : x;
      }
    Dev d_blind(blind.size() * 32), d_src(n_cols * sizeof(vdb_colsrc)), d_cols(n_cols * rows * 32), d_ext(n_cols * rows * 4 * 32);
    check(vdb_memcpy_h2d(d_blind.p, blind.data(), blind.size() * 32));
    check(vdb_colsrc_build_dev(d_stream.as<F>(), n_cells, bp.data(), n_bp, k, 0, n_adv, d_blind.as<F>(), N_BLIND, d_src.as<vdb_colsrc>()));
    if (n_lk)
      check(vdb_colsrc_build_lookup_dev(d_lookup.as<F>(), lookups, k, MINIMUM_ROWS, 0, n_lk, d_blind.as<F>(n_adv * N_BLIND * 32), N_BLIND,
                                        d_src.as<vdb_colsrc>(n_adv * sizeof(vdb_colsrc))));
    // Prove arm: witness -> commit (queued; the bucket folding of the last batch runs beside the transforms) -> NTTs -> join
    std::vector<vdb_g1> commitments(n_cols);
    float ms = 0;
    check(vdb_timer_start());
    witness(false);
    check(vdb_msm_batch_src_dev_begin(srs, 1, d_src.as<vdb_colsrc>(), n_cols, rows, N_BLIND, nullptr, nullptr));
    check(vdb_lagrange_to_coeff_src_dev(d_src.as<vdb_colsrc>(), d_cols.as<F>(), n_cols, k, N_BLIND));
    check(vdb_coeff_to_extended_dev(d_cols.as<F>(), d_ext.as<F>(), n_cols, k, 2));
    check(vdb_msm_batch_end(commitments.data(), n_cols));
    check(vdb_timer_stop(&ms));
    std::printf("cells %llu lookups %llu advice_columns %llu lookup_columns %llu step_ms %.2f\n", (unsigned long long)n_cells, (unsigned long long)lookups,
                (unsigned long long)n_adv, (unsigned long long)n_lk, ms);
    // dump for the checker: the Lagrange image of three columns, laid out the ordinary way
    std::FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    const uint64_t hdr[3] = {n_adv, n_lk, rows};
    std::fwrite(hdr, 8, 3, f);
    std::fwrite(commitments.data(), sizeof(vdb_g1), n_cols, f);
    std::vector<F> col(rows);
    Dev d_one(rows * 32);
    for (uint64_t c : {(uint64_t)0, n_adv - 1}) {
      check(vdb_layout_columns_range_dev(d_stream.as<F>(), n_cells, bp.data(), n_bp, k, c, c + 1, d_one.as<F>(), d_blind.as<F>(), N_BLIND));
      check(vdb_memcpy_d2h(col.data(), d_one.p, rows * 32));
      std::fwrite(col.data(), 32, rows, f);
    }
    if (n_lk) {
      check(vdb_layout_lookup_range_dev(d_lookup.as<F>(), lookups, k, MINIMUM_ROWS, 0, 1, d_one.as<F>(), d_blind.as<F>(n_adv * N_BLIND * 32), N_BLIND));
      check(vdb_memcpy_d2h(col.data(), d_one.p, rows * 32));
      std::fwrite(col.data(), 32, rows, f);
    }
    std::fclose(f);
    vdb_srs_free(srs);
    vdb_shutdown();
  } catch (const Error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
