// Host check of Gadgets::divmod_u256 against Python integers: hipcc -O2 -std=c++17 --offload-arch=gfx950 -o divtest tools/divtest.hip && ./divtest > div.out;
// every line "a b q r" in hex must satisfy q == a // b, r == a % b (6,912 cases: every pair of bit lengths, equal operands, all-ones, powers of two, zero)
#include <cstdio>
#include <cstdint>
#include "../halo2_vectordb_amd/csrc/gadgets.hpp"
using namespace vdb;
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static u256 rnd_bits(unsigned bits) {
  u256 v = u256_zero();
  for (int i = 0; i < 8; i++) v.w[i] = rnd();
  if (bits == 0) return u256_zero();
  v = u256_low_bits(v, bits);
  // force the top bit
  v.w[(bits - 1) >> 5] |= 1u << ((bits - 1) & 31);
  return v;
}
static void pr(const u256& v) { for (int i = 7; i >= 0; i--) printf("%08x", v.w[i]); }
int main() {
  unsigned sizes[] = {1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 148, 160, 191, 192, 193, 224, 250, 255, 256};
  for (unsigned ia = 0; ia < sizeof(sizes) / 4; ia++)
    for (unsigned ib = 0; ib < sizeof(sizes) / 4; ib++)
      for (int rep = 0; rep < 12; rep++) {
        u256 a = rnd_bits(sizes[ia]), b = rnd_bits(sizes[ib]);
        if (rep == 1) a = b;                                         // equal
        if (rep == 2) for (int i = 0; i < 8; i++) a.w[i] = 0xffffffffu;  // all ones
        if (rep == 3) { b = u256_zero(); b.w[(sizes[ib] - 1) >> 5] = 1u << ((sizes[ib] - 1) & 31); }  // power of two
        if (rep == 4) { for (int i = 0; i < 8; i++) b.w[i] = 0xffffffffu; b = u256_low_bits(b, sizes[ib]); }  // all ones of that length
        if (rep == 5) a = u256_zero();
        u256 q, r;
        Gadgets::divmod_u256(a, b, q, r);
        pr(a); printf(" "); pr(b); printf(" "); pr(q); printf(" "); pr(r); printf("\n");
      }
  return 0;
}
