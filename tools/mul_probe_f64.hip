// Throughput + correctness probe: the nine-limb integer Montgomery product (mont_core29, v_mad_u64_u32 on 29-bit limbs)
// against a Montgomery product on FIVE 52-BIT LIMBS HELD IN DOUBLES, multiplied with v_fma_f64 (the split of a 104-bit
// limb product into two 52-bit halves by two fused multiply-adds under round-toward-zero, N. Emmart's construction):
//     hi = fma_rz(a, b, 2^104)                    -> bits: 0x467 exponent | floor(a b / 2^52)
//     lo = fma_rz(a, b, (2^104 + 2^52) - hi)      -> bits: 0x433 exponent | a b mod 2^52
// The halves are accumulated as 64-bit integers (the raw bit patterns; the exponent patterns are pre-subtracted from the
// column accumulators), the reduction is word-serial with R' = 2^260.  Per 52 x 52 limb product: 2 FMA + 1 f64 subtract +
// 2 64-bit integer adds = 5 instructions for 2,704 bit^2, against ONE v_mad_u64_u32 for 841 bit^2 (which also accumulates).
// The round-toward-zero mode of the f64 pipe is set once per kernel (s_setreg MODE.FP_ROUND[3:2] = 3).
// Prints G products/s of both and checks the f64 product against exact integers on the host.
// Build: hipcc -O3 --offload-arch=gfx950 -I../halo2_vectordb_amd/csrc mul_probe_f64.hip -o mul_probe_f64
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "field.hpp"
using namespace vdb;

typedef unsigned __int128 u128;
static constexpr uint64_t M52 = (1ull << 52) - 1;
static constexpr uint64_t PAT_LO = 0x4330000000000000ull;  // exponent of [2^52, 2^53)
static constexpr uint64_t PAT_HI = 0x4670000000000000ull;  // exponent of [2^104, 2^105)

struct F5Consts {
  double p[5];       // r in 52-bit limbs
  double pinv;       // -r^-1 mod 2^52
  uint64_t init[11];  // column accumulators start at minus the exponent patterns their terms will carry
};

// The compiler does not model the rounding mode as an input of floating-point instructions and moves them across
// s_setreg; volatile asm statements keep their order among themselves, so every value that enters (leaves) the region
// computed under the changed mode is passed through one placed after (before) the mode switch.
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void round_toward_zero_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); }
__device__ __forceinline__ void round_toward_zero_off() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"); }
__device__ __forceinline__ double bits_to_double(uint64_t b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ uint64_t double_bits(double d) { return (uint64_t)__double_as_longlong(d); }
// an integer below 2^52 as an exact double
__device__ __forceinline__ double int52_to_double(uint64_t v) { return bits_to_double(v | PAT_LO) - 4503599627370496.0; }

// out = a * b / 2^260 mod r (below 2 r), limbs below 2^52 as doubles; rounding mode of the wave must be toward zero
__device__ __forceinline__ void mont_f64(double out[5], const double a[5], const double b[5], const F5Consts& K) {
  const double C1 = 20282409603651670423947251286016.0;                 // 2^104
  const double C2 = 20282409603651670423947251286016.0 + 4503599627370496.0;  // 2^104 + 2^52 (exact: 53 significant bits)
  uint64_t c[11];
#pragma unroll
  for (int k = 0; k < 11; k++) c[k] = K.init[k];
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = __builtin_fma(a[i], b[j], C1);
      const double lo = __builtin_fma(a[i], b[j], C2 - hi);
      c[i + j] += double_bits(lo);
      c[i + j + 1] += double_bits(hi);
    }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const double q = int52_to_double(c[i] & M52);
    const double mh = __builtin_fma(q, K.pinv, C1);
    const double ml = __builtin_fma(q, K.pinv, C2 - mh);  // 2^52 + (q pinv mod 2^52)
    const double m = ml - 4503599627370496.0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const double hi = __builtin_fma(m, K.p[j], C1);
      const double lo = __builtin_fma(m, K.p[j], C2 - hi);
      c[i + j] += double_bits(lo);
      c[i + j + 1] += double_bits(hi);
    }
    c[i + 1] += c[i] >> 52;  // the low 52 bits of column i are zero now
  }
#pragma unroll
  for (int k = 5; k < 9; k++) {
    c[k + 1] += c[k] >> 52;
    c[k] &= M52;
  }
#pragma unroll
  for (int k = 0; k < 5; k++) out[k] = int52_to_double(c[5 + k]);
}

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint32_t* out, const uint32_t* w, F5Consts K, int iters) {
  uint32_t s = 0;
  if (KIND == 0) {
    uint32_t a[2][9], W[9];
    for (int j = 0; j < 9; j++) {
      W[j] = (w[j] + threadIdx.x) & 0x1fffffffu;
      a[0][j] = (w[j] * 3 + blockIdx.x) & 0x1fffffffu;
      a[1][j] = (w[j] * 5 + blockIdx.x) & 0x1fffffffu;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        uint32_t r[9];
        mont_core29<Fr>(r, a[c], W);
#pragma unroll
        for (int j = 0; j < 9; j++) a[c][j] = r[j];
      }
    }
    for (int j = 0; j < 9; j++) s += a[0][j] ^ a[1][j];
  } else {
    double a[2][5], W[5];
    for (int j = 0; j < 5; j++) {
      W[j] = (double)(((uint64_t)w[j] << 19 | threadIdx.x) & M52);
      a[0][j] = (double)((((uint64_t)w[j] * 3) << 18 | blockIdx.x) & M52);
      a[1][j] = (double)((((uint64_t)w[j] * 5) << 17 | blockIdx.x) & M52);
    }
    W[4] = a[0][4] = a[1][4] = 1234567.0;  // keep the operands below r (top limb of r is 0x30644e72e131a ~ 2^45.6)
    round_toward_zero_on();  // MODE.FP_ROUND[3:2] (f64 / f16) = round toward zero
    for (int j = 0; j < 5; j++) pin(W[j]), pin(a[0][j]), pin(a[1][j]);
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        double r[5];
        mont_f64(r, a[c], W, K);
#pragma unroll
        for (int j = 0; j < 5; j++) a[c][j] = r[j];
      }
    }
    for (int j = 0; j < 5; j++) pin(a[0][j]), pin(a[1][j]);
    round_toward_zero_off();
    for (int j = 0; j < 5; j++) s += (uint32_t)double_bits(a[0][j]) ^ (uint32_t)double_bits(a[1][j]);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one product per thread on given operands, for the check against exact integers
__global__ void check_kernel(const double* a, const double* b, double* r, F5Consts K, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double x[5], y[5], o[5];
  for (int j = 0; j < 5; j++) x[j] = a[t * 5 + j], y[j] = b[t * 5 + j];
  round_toward_zero_on();
  for (int j = 0; j < 5; j++) pin(x[j]), pin(y[j]);
  mont_f64(o, x, y, K);
  for (int j = 0; j < 5; j++) pin(o[j]);
  round_toward_zero_off();
  for (int j = 0; j < 5; j++) r[t * 5 + j] = o[j];
}

// ---- host big integers (little-endian 64-bit words), just enough for the check
struct Big {
  uint64_t w[10];
};
static Big big_from_limbs52(const double* l) {
  Big r;
  memset(&r, 0, sizeof(r));
  for (int j = 0; j < 5; j++) {
    uint64_t v = (uint64_t)l[j];
    int pos = 52 * j, wi = pos >> 6, o = pos & 63;
    r.w[wi] |= v << o;
    if (o > 12) r.w[wi + 1] |= v >> (64 - o);
  }
  return r;
}
static Big big_mul(const Big& a, const Big& b) {  // inputs below 2^320
  Big r;
  memset(&r, 0, sizeof(r));
  for (int i = 0; i < 5; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 5; j++) {
      u128 t = (u128)a.w[i] * b.w[j] + r.w[i + j] + carry;
      r.w[i + j] = (uint64_t)t;
      carry = (uint64_t)(t >> 64);
    }
    r.w[i + 5] += carry;
  }
  return r;
}
static int big_cmp(const Big& a, const Big& b) {
  for (int i = 9; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
static void big_sub(Big& a, const Big& b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 10; i++) {
    u128 t = (u128)a.w[i] - b.w[i] - borrow;
    a.w[i] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
}
static Big big_shl(const Big& a, int bits) {
  Big r;
  memset(&r, 0, sizeof(r));
  int ws = bits >> 6, o = bits & 63;
  for (int i = 9; i >= ws; i--) {
    r.w[i] = a.w[i - ws] << o;
    if (o && i - ws - 1 >= 0) r.w[i] |= a.w[i - ws - 1] >> (64 - o);
  }
  return r;
}
// x mod p by shift-and-subtract (x below 2^640)
static Big big_mod(Big x, const Big& p) {
  for (int s = 640 - 254; s >= 0; s--) {
    Big ps = big_shl(p, s);
    if (big_cmp(x, ps) >= 0) big_sub(x, ps);
  }
  return x;
}

template <int KIND>
static double run(const char* name, const uint32_t* dw, const F5Consts& K) {
  const int blocks = 256 * 16, iters = 2048;
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, dw, K, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, dw, K, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * 2;
  double rate = ops / (ms * 1e-3) / 1e9;
  printf("%-22s %8.2f G mul/s   (%.3f ms)\n", name, rate, ms);
  hipFree(out);
  return rate;
}

int main() {
  // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
  const uint64_t rw[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  F5Consts K;
  Big P;
  memset(&P, 0, sizeof(P));
  for (int i = 0; i < 4; i++) P.w[i] = rw[i];
  for (int j = 0; j < 5; j++) {
    int pos = 52 * j, wi = pos >> 6, o = pos & 63;
    uint64_t v = rw[wi] >> o;
    if (o > 12 && wi + 1 < 4) v |= rw[wi + 1] << (64 - o);
    K.p[j] = (double)(v & M52);
  }
  // -r^-1 mod 2^52 by Newton iteration on the low word
  uint64_t inv = 1;
  for (int i = 0; i < 6; i++) inv *= 2 - rw[0] * inv;
  K.pinv = (double)((0 - inv) & M52);
  // exponent patterns per column: product terms + the five reduction rounds
  int n_lo[11] = {0}, n_hi[11] = {0};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 5; j++) {
      n_lo[i + j] += 2;      // a_i b_j and m_i p_j
      n_hi[i + j + 1] += 2;
    }
  for (int k = 0; k < 11; k++) K.init[k] = 0 - ((uint64_t)n_lo[k] * PAT_LO + (uint64_t)n_hi[k] * PAT_HI);

  // ---- correctness: 4096 random products against exact integers
  const int n = 4096;
  double *ha = new double[n * 5], *hb = new double[n * 5], *hr = new double[n * 5];
  uint64_t seed = 88172645463325252ull;
  auto rnd = [&]() {
    seed ^= seed << 13;
    seed ^= seed >> 7;
    seed ^= seed << 17;
    return seed;
  };
  for (int t = 0; t < n; t++)
    for (int j = 0; j < 5; j++) {
      ha[t * 5 + j] = (double)(rnd() & (j == 4 ? ((1ull << 45) - 1) : M52));  // below 2^253 < r
      hb[t * 5 + j] = (double)(rnd() & (j == 4 ? ((1ull << 45) - 1) : M52));
    }
  for (int j = 0; j < 5; j++) ha[j] = K.p[j], hb[j] = K.p[j];   // (r, r): every limb product at its largest
  ha[4] -= 1.0, hb[4] -= 1.0;
  double *da, *db, *dr;
  hipMalloc(&da, n * 5 * 8);
  hipMalloc(&db, n * 5 * 8);
  hipMalloc(&dr, n * 5 * 8);
  hipMemcpy(da, ha, n * 5 * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, n * 5 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dr, K, n);
  hipMemcpy(hr, dr, n * 5 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < n; t++) {
    // r * 2^260 == a * b (mod p), r below 2 p, limbs below 2^52
    Big A = big_from_limbs52(ha + t * 5), Bq = big_from_limbs52(hb + t * 5), Rr = big_from_limbs52(hr + t * 5);
    Big lhs = big_mod(big_shl(Rr, 260), P), rhs = big_mod(big_mul(A, Bq), P);
    Big twoP = big_shl(P, 1);
    bool ok = big_cmp(lhs, rhs) == 0 && big_cmp(Rr, twoP) < 0;
    for (int j = 0; j < 5; j++) ok = ok && hr[t * 5 + j] >= 0 && hr[t * 5 + j] < 4503599627370496.0;
    if (!ok) bad++;
  }
  printf("f64 product: %d of %d products differ from the exact result\n", bad, n);

  uint32_t hw[18];
  for (int i = 0; i < 18; i++) hw[i] = 0x12345678u * (i + 1) + 0x9abcdefu;
  uint32_t* dw;
  hipMalloc(&dw, sizeof(hw));
  hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
  double r0 = 0, r1 = 0;
  for (int rep = 0; rep < 2; rep++) {
    r0 = run<0>("mont_core29 (9 x 29b)", dw, K);
    r1 = run<1>("mont_f64 (5 x 52b FMA)", dw, K);
  }
  printf("{\"mont_core29_gmul_s\": %.2f, \"mont_f64_gmul_s\": %.2f, \"ratio_f64_over_int\": %.3f, \"f64_wrong_products\": %d}\n", r0, r1, r1 / r0, bad);
  return bad ? 1 : 0;
}
