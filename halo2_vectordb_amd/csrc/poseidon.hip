// Poseidon: host-side parameter generation (Grain LFSR + optimized schedule) and the hash-only
// device kernels behind vdb_poseidon_* (SURVEY §8 b6).  The trace-emitting variant used for witness
// generation lives in witness.hip and shares poseidon.hpp.
#include "poseidon.hpp"

#include "common.hpp"

namespace vdb {

// ------------------------------------------------------------------ host: parameters
namespace {

// Grain LFSR of the Poseidon paper's parameter script (80-bit register, taps 62/51/38/23/13/0,
// 160 warm-up clocks, self-shrinking output)
struct GrainLfsr {
  bool s[80];
  bool clock() {
    bool nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0];
    for (int i = 0; i < 79; i++) s[i] = s[i + 1];
    s[79] = nb;
    return nb;
  }
  GrainLfsr(unsigned field_bits, unsigned t, unsigned r_f, unsigned r_p) {
    int pos = 0;
    auto put = [&](unsigned v, int bits) {
      for (int i = bits - 1; i >= 0; i--) s[pos++] = (v >> i) & 1u;
    };
    put(1, 2);  // prime field
    put(0, 4);  // x^alpha S-box
    put(field_bits, 12);
    put(t, 12);
    put(r_f, 10);
    put(r_p, 10);
    for (int i = 0; i < 30; i++) s[pos++] = true;
    for (int i = 0; i < 160; i++) clock();
  }
  bool next_bit() {
    for (;;) {
      bool keep = clock();
      bool v = clock();
      if (keep) return v;
    }
  }
  // 254 bits, most significant first
  u256 next_integer() {
    u256 v = u256_zero();
    for (int i = 253; i >= 0; i--)
      if (next_bit()) v.w[i >> 5] |= 1u << (i & 31);
    return v;
  }
  u256 next_field(bool reject) {
    const u256 p = mod_p<Fr>();
    for (;;) {
      u256 v = next_integer();
      if (!u256_geq(v, p)) return to_mont<Fr>(v);
      if (!reject) {  // reduce: v < 2^254 < 2p
        u256 r;
        u256_sub(r, v, p);
        return to_mont<Fr>(r);
      }
    }
  }
};

typedef std::vector<std::vector<u256>> Mat;
Mat mat_identity(int n) {
  Mat m(n, std::vector<u256>(n, u256_zero()));
  for (int i = 0; i < n; i++) m[i][i] = mont_one<Fr>();
  return m;
}
Mat mat_transpose(const Mat& a) {
  int n = (int)a.size();
  Mat t(n, std::vector<u256>(n));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) t[i][j] = a[j][i];
  return t;
}
Mat mat_mul(const Mat& a, const Mat& b) {
  int n = (int)a.size();
  Mat r(n, std::vector<u256>(n, u256_zero()));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      for (int k = 0; k < n; k++) r[i][j] = fr_add(r[i][j], fr_mul(a[i][k], b[k][j]));
  return r;
}
std::vector<u256> mat_vec(const Mat& a, const std::vector<u256>& v) {
  int n = (int)a.size();
  std::vector<u256> r(n, u256_zero());
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) r[i] = fr_add(r[i], fr_mul(a[i][j], v[j]));
  return r;
}
Mat mat_inverse(Mat w) {
  int n = (int)w.size();
  Mat inv = mat_identity(n);
  for (int c = 0; c < n; c++) {
    int piv = c;
    while (u256_is_zero(w[piv][c])) piv++;
    std::swap(w[c], w[piv]);
    std::swap(inv[c], inv[piv]);
    u256 pi = mont_inv<Fr>(w[c][c]);
    for (int j = 0; j < n; j++) {
      w[c][j] = fr_mul(w[c][j], pi);
      inv[c][j] = fr_mul(inv[c][j], pi);
    }
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      u256 f = w[r][c];
      for (int j = 0; j < n; j++) {
        w[r][j] = fr_sub(w[r][j], fr_mul(f, w[c][j]));
        inv[r][j] = fr_sub(inv[r][j], fr_mul(f, inv[c][j]));
      }
    }
  }
  return inv;
}

}  // namespace

// Plain (unoptimised) parameters for any width: round constants row by row and the Cauchy MDS matrix, Montgomery form.
// Used by the transcript sponge (transcript.hip), whose width differs from the chip's.
void poseidon_plain_params(int t, int r_f, int r_p, std::vector<u256>& rc, std::vector<u256>& mds) {
  GrainLfsr g(254, t, r_f, r_p);
  rc.resize((size_t)(r_f + r_p) * t);
  for (auto& c : rc) c = g.next_field(true);
  std::vector<u256> xs(t), ys(t);
  for (auto& x : xs) x = g.next_field(false);
  for (auto& y : ys) y = g.next_field(false);
  mds.resize((size_t)t * t);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) mds[(size_t)i * t + j] = mont_inv<Fr>(fr_add(xs[i], ys[j]));
}

// Builds the optimized schedule used by the halo2-lib Poseidon chip (PSE poseidon `Spec`) for any width:
// round constants folded through M^-1, partial rounds as sparse matrices M = M' * M''.
void poseidon_build_opt(int t, int r_f, int r_p, PoseidonOpt* out) {
  const int half = r_f / 2, rp = r_p, rounds = r_f + r_p;
  GrainLfsr g(254, t, r_f, r_p);
  std::vector<std::vector<u256>> rc(rounds, std::vector<u256>(t));
  for (auto& row : rc)
    for (auto& c : row) c = g.next_field(true);
  std::vector<u256> xs(t), ys(t);
  for (auto& x : xs) x = g.next_field(false);
  for (auto& y : ys) y = g.next_field(false);
  Mat mds(t, std::vector<u256>(t));
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) mds[i][j] = mont_inv<Fr>(fr_add(xs[i], ys[j]));  // Cauchy matrix
  Mat minv = mat_inverse(mds);
  out->t = t, out->half = half, out->rp = rp;
  out->start.assign((size_t)(half + 1) * t, u256_zero());
  out->partial.assign(rp, u256_zero());
  out->end.assign((size_t)(half > 1 ? half - 1 : 0) * t, u256_zero());
  out->mds.assign((size_t)t * t, u256_zero());
  out->pre_sparse.assign((size_t)t * t, u256_zero());
  out->sparse_row.assign((size_t)rp * t, u256_zero());
  out->sparse_col.assign((size_t)rp * (t - 1), u256_zero());

  for (int i = 0; i < t; i++) out->start[i] = rc[0][i];
  for (int r = 1; r < half; r++) {
    auto v = mat_vec(minv, rc[r]);
    for (int i = 0; i < t; i++) out->start[(size_t)r * t + i] = v[i];
  }
  std::vector<u256> acc = rc[half + rp];
  for (int p = rp - 1; p >= 0; p--) {
    auto tmp = mat_vec(minv, acc);
    out->partial[p] = tmp[0];
    tmp[0] = u256_zero();
    for (int i = 0; i < t; i++) acc[i] = fr_add(tmp[i], rc[half + p][i]);
  }
  {
    auto v = mat_vec(minv, acc);
    for (int i = 0; i < t; i++) out->start[(size_t)half * t + i] = v[i];
  }
  for (int r = 0; r < half - 1; r++) {
    auto v = mat_vec(minv, rc[half + rp + 1 + r]);
    for (int i = 0; i < t; i++) out->end[(size_t)r * t + i] = v[i];
  }
  // sparse factorisation, walking from the last partial round back to the first
  Mat mT = mat_transpose(mds), cur = mT;
  for (int p = 0; p < rp; p++) {
    // cur = [[c00, v],[w, Mhat]]  ->  M' = diag(1, Mhat),  M'' = [[c00, v],[Mhat^-1 w, I]]
    Mat hat(t - 1, std::vector<u256>(t - 1));
    std::vector<u256> w(t - 1);
    for (int i = 1; i < t; i++) {
      w[i - 1] = cur[i][0];
      for (int j = 1; j < t; j++) hat[i - 1][j - 1] = cur[i][j];
    }
    auto what = mat_vec(mat_inverse(hat), w);
    Mat mprime = mat_identity(t);
    for (int i = 1; i < t; i++)
      for (int j = 1; j < t; j++) mprime[i][j] = hat[i - 1][j - 1];
    int dst = rp - 1 - p;
    // the chip applies transpose(M''): first row = (c00, w_hat...), first column below = v
    out->sparse_row[(size_t)dst * t] = cur[0][0];
    for (int j = 1; j < t; j++) out->sparse_row[(size_t)dst * t + j] = what[j - 1];
    for (int i = 1; i < t; i++) out->sparse_col[(size_t)dst * (t - 1) + i - 1] = cur[0][i];
    cur = mat_mul(mT, mprime);
  }
  Mat pre = mat_transpose(cur);
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) {
      out->mds[(size_t)i * t + j] = mds[i][j];
      out->pre_sparse[(size_t)i * t + j] = pre[i][j];
    }
}
// the chip's width, in the fixed-size layout the kernels read
void poseidon_build_spec(PoseidonSpec* out) {
  PoseidonOpt o;
  poseidon_build_opt(PSD_T, PSD_RF, PSD_RP, &o);
  const int t = PSD_T;
  for (int r = 0; r <= PSD_HALF; r++)
    for (int i = 0; i < t; i++) out->start[r][i] = o.start[(size_t)r * t + i];
  for (int p = 0; p < PSD_RP; p++) {
    out->partial[p] = o.partial[p];
    for (int j = 0; j < t; j++) out->sparse_row[p][j] = o.sparse_row[(size_t)p * t + j];
    for (int j = 0; j < t - 1; j++) out->sparse_col[p][j] = o.sparse_col[(size_t)p * (t - 1) + j];
  }
  for (int r = 0; r < PSD_HALF - 1; r++)
    for (int i = 0; i < t; i++) out->end[r][i] = o.end[(size_t)r * t + i];
  for (int i = 0; i < t; i++)
    for (int j = 0; j < t; j++) {
      out->mds[i][j] = o.mds[(size_t)i * t + j];
      out->pre_sparse[i][j] = o.pre_sparse[(size_t)i * t + j];
    }
  u256 cap = u256_zero();
  cap.w[2] = 1;  // 2^64
  out->cap = to_mont<Fr>(cap);
  out->one = mont_one<Fr>();
}

static PoseidonSpec g_spec_host;
static std::once_flag g_spec_once;
void poseidon_release(Context& c) {
  if (c.poseidon_spec) (void)hipFree(c.poseidon_spec);
  c.poseidon_spec = nullptr;
}
// the host copy is built once per process, the device copy once per bound device (Context::poseidon_spec)
int poseidon_spec_dev(const PoseidonSpec** dev_out, const PoseidonSpec** host_out) {
  std::call_once(g_spec_once, [] { poseidon_build_spec(&g_spec_host); });
  Context& c = ctx();
  if (dev_out && !c.poseidon_spec) {
    VDB_HIP(hipMalloc(&c.poseidon_spec, sizeof(PoseidonSpec)));
    VDB_HIP(hipMemcpy(c.poseidon_spec, &g_spec_host, sizeof(PoseidonSpec), hipMemcpyHostToDevice));
  }
  if (dev_out) *dev_out = static_cast<const PoseidonSpec*>(c.poseidon_spec);
  if (host_out) *host_out = &g_spec_host;
  return VDB_OK;
}

// ------------------------------------------------------------------ device kernels (hash only)
__global__ __launch_bounds__(256) void k_poseidon_hash_many(const PoseidonSpec* __restrict__ sp, const u256* __restrict__ in,
                                                           size_t n_msgs, size_t msg_len, u256* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_msgs) return;
  st256(out + t, psd_hash(sp, in + t * msg_len, msg_len, 1));
}
// one Merkle level: out[i] = H(in[2i], in[2i+1])
__global__ __launch_bounds__(256) void k_poseidon_level(const PoseidonSpec* __restrict__ sp, const u256* __restrict__ in,
                                                       size_t n_out, u256* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  st256(out + t, psd_hash(sp, in + 2 * t, 2, 1));
}
__global__ __launch_bounds__(256) void k_poseidon_permute(const PoseidonSpec* __restrict__ sp, u256* __restrict__ states, size_t n) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u256 st[PSD_T], zero[PSD_RATE];
  for (int i = 0; i < PSD_T; i++) st[i] = ld256(states + t * PSD_T + i);
  for (int i = 0; i < PSD_RATE; i++) zero[i] = u256_zero();
  psd_permute_absorb(sp, st, zero, PSD_RATE);  // absorbing RATE zeros adds only the round-0 constants
  for (int i = 0; i < PSD_T; i++) st256(states + t * PSD_T + i, st[i]);
}

// device-level merkle root over already-resident leaves buffer (size: next pow2), result in lv[0]
int poseidon_merkle_dev(const u256* vectors_dev, size_t n, size_t dim, u256* lv /* leaves pow2 */, u256* tmp) {
  Context& c = ctx();
  const PoseidonSpec* sp;
  int rc = poseidon_spec_dev(&sp, nullptr);
  if (rc) return rc;
  size_t leaves = 1;
  while (leaves < n) leaves <<= 1;
  VDB_HIP(hipMemsetAsync(lv, 0, leaves * sizeof(u256), c.stream));
  {
    VDB_PROF("k_poseidon_hash_many");
    hipLaunchKernelGGL(k_poseidon_hash_many, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, sp, vectors_dev, n, dim, lv);
  }
  VDB_LAUNCH_CHECK();
  u256 *a = lv, *b = tmp;
  while (leaves > 1) {
    size_t no = leaves / 2;
    {
      VDB_PROF("k_poseidon_level");
      hipLaunchKernelGGL(k_poseidon_level, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, c.stream, sp, a, no, b);
    }
    VDB_LAUNCH_CHECK();
    std::swap(a, b);
    leaves = no;
  }
  if (a != lv) VDB_HIP(hipMemcpyAsync(lv, a, sizeof(u256), hipMemcpyDeviceToDevice, c.stream));
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int vdb_poseidon_hash_many(const vdb_fr* inputs, size_t n_msgs, size_t msg_len, vdb_fr* digests) {
  VDB_REQUIRE_INIT();
  VDB_ARG(digests && (inputs || n_msgs * msg_len == 0), "null pointer");
  if (n_msgs == 0) return VDB_OK;
  Context& c = ctx();
  const PoseidonSpec* sp;
  int rc = poseidon_spec_dev(&sp, nullptr);
  if (rc) return rc;
  size_t in_bytes = n_msgs * msg_len * sizeof(u256);
  u256* din = (u256*)scratch_get(0, in_bytes ? in_bytes : 32);
  u256* dout = (u256*)scratch_get(1, n_msgs * sizeof(u256));
  if (!din || !dout) return VDB_ERR_OOM;
  if (in_bytes) VDB_HIP(hipMemcpyAsync(din, inputs, in_bytes, hipMemcpyHostToDevice, c.stream));
  {
    VDB_PROF("k_poseidon_hash_many");
    hipLaunchKernelGGL(k_poseidon_hash_many, dim3((unsigned)((n_msgs + 255) / 256)), dim3(256), 0, c.stream, sp, din, n_msgs, msg_len, dout);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(digests, dout, n_msgs * sizeof(u256), hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

int vdb_poseidon_merkle_root(const vdb_fr* vectors, size_t n, size_t dim, vdb_fr* root) {
  VDB_REQUIRE_INIT();
  VDB_ARG(vectors && root && n > 0, "null pointer or empty database");
  Context& c = ctx();
  size_t leaves = 1;
  while (leaves < n) leaves <<= 1;
  u256* din = (u256*)scratch_get(0, n * dim * sizeof(u256) + 32);
  u256* lv = (u256*)scratch_get(1, leaves * sizeof(u256));
  u256* tmp = (u256*)scratch_get(2, leaves * sizeof(u256));
  if (!din || !lv || !tmp) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(din, vectors, n * dim * sizeof(u256), hipMemcpyHostToDevice, c.stream));
  int rc = poseidon_merkle_dev(din, n, dim, lv, tmp);
  if (rc) return rc;
  VDB_HIP(hipMemcpyAsync(root, lv, sizeof(u256), hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

int vdb_poseidon_permute(vdb_fr* states, size_t n) {
  VDB_REQUIRE_INIT();
  VDB_ARG(states, "null pointer");
  if (n == 0) return VDB_OK;
  Context& c = ctx();
  const PoseidonSpec* sp;
  int rc = poseidon_spec_dev(&sp, nullptr);
  if (rc) return rc;
  size_t bytes = n * PSD_T * sizeof(u256);
  u256* d = (u256*)scratch_get(0, bytes);
  if (!d) return VDB_ERR_OOM;
  VDB_HIP(hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, c.stream));
  {
    VDB_PROF("k_poseidon_permute");
    hipLaunchKernelGGL(k_poseidon_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, sp, d, n);
  }
  VDB_LAUNCH_CHECK();
  VDB_HIP(hipMemcpyAsync(states, d, bytes, hipMemcpyDeviceToHost, c.stream));
  VDB_HIP(hipStreamSynchronize(c.stream));
  return VDB_OK;
}

}  // extern "C"
