// Fiat–Shamir transcript and proof byte stream — host code only, sequential and tiny by nature.
// Stands in for snark_verifier's PoseidonTranscript<NativeLoader, _> as the reference constructs it
// (/root/reference/src/scaffold/mod.rs:309-310: PoseidonTranscript::<NativeLoader, _>::new::<0>), [UPSTREAM-RECALL]:
//   * Poseidon sponge over BN254 Fr with T = 5, RATE = 4, R_F = 8, R_P = 60 (snark-verifier-sdk's constants), parameters
//     from the Grain LFSR as for the chip; state starts [2^64, 0, 0, 0, 0] and is never reset;
//   * absorbed values are buffered; a squeeze feeds the buffer RATE values at a time — a short chunk gets +1 after its last
//     value, and when the buffer length is a multiple of RATE (empty included) one more permutation absorbs only that +1 —
//     and returns state[1] (the same sponge rule as PoseidonChip);
//   * a G1 point is absorbed as its affine x and y, each reduced into Fr; the identity as (0, 0);
//   * written to the proof: points compressed to 32 bytes (x little-endian, one of the two spare top bits of the last byte =
//     y is odd, the identity all zero), scalars as 32 little-endian bytes of the canonical value.  Which spare bit depends on
//     the halo2curves version as recalled — bit 6 (0x40) from 0.4 on, bit 7 (0x80) in the 0.3.x releases of the reference's
//     dependency era; the default is bit 6, vdb_transcript_set_sign_bit selects the other.
// Parity unpinned: the parameters and encodings above are recalled, the reference holds no proof bytes.
// The sponge is cross-checked against an independent Python restatement (tests/test_transcript_cpu.py), and at T = 3
// against the chip's optimised schedule.
#include <vector>

#include "common.hpp"
#include "poseidon.hpp"

using namespace vdb;

struct vdb_transcript {
  int t, rate, r_f, r_p;
  std::vector<u256> state, buf;
  PoseidonOpt opt;  // the permutation's optimised schedule (sparse partial rounds), poseidon.hip
  uint8_t sign_mask = 0x40;  // where a compressed point carries "y is odd" (vdb_transcript_set_sign_bit)
  std::vector<uint8_t> bytes;
};

namespace {

// Host arithmetic for the sponge: the same Montgomery representatives (R = 2^256) on four 64-bit limbs with 128-bit
// products — a proof of a few thousand columns absorbs tens of thousands of values, i.e. ~10^4 permutations of 68 rounds,
// and the device-oriented 29-bit-limb product is several times slower on a CPU core.
struct F4 {
  uint64_t l[4];
};
constexpr uint64_t FR_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr uint64_t FR_INV = 0xc2e1f593efffffffull;  // -r^-1 mod 2^64
inline F4 to_f4(const u256& a) {
  F4 r;
  memcpy(r.l, a.w, 32);
  return r;
}
inline u256 from_f4(const F4& a) {
  u256 r;
  memcpy(r.w, a.l, 32);
  return r;
}
inline bool geq_p(const uint64_t t[4]) {
  for (int i = 3; i >= 0; i--)
    if (t[i] != FR_P[i]) return t[i] > FR_P[i];
  return true;
}
inline void sub_p(uint64_t t[4]) {
  unsigned __int128 b = 0;
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)t[i] - FR_P[i] - (uint64_t)b;
    t[i] = (uint64_t)d;
    b = (d >> 64) & 1;
  }
}
inline F4 f4_add(const F4& a, const F4& b) {
  F4 r;
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (unsigned __int128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || geq_p(r.l)) sub_p(r.l);  // a + b < 2 r < 2^255: no carry out in fact
  return r;
}
inline F4 f4_mul(const F4& a, const F4& b) {  // CIOS
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * FR_INV;
    c = (unsigned __int128)m * FR_P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (unsigned __int128)m * FR_P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  F4 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_p(r.l)) sub_p(r.l);
  return r;
}

// sum_j a[j] b[j] with ONE Montgomery reduction: the n (<= 16) double-width products are summed in 576 bits, then reduced
// word by word.  A proof of 2 x 10^4 columns absorbs ~2 x 10^5 values, i.e. ~5 x 10^4 permutations: this dot product and the
// sparse partial rounds below are what the transcript's host time is made of.
inline F4 f4_dot(const F4* a, const F4* b, int n, size_t stride_b = 1) {
  // product scanning: column c collects every x_i y_j with i + j = c of every pair into a three-word accumulator
  uint64_t T[9];
  uint64_t a0 = 0, a1 = 0, a2 = 0;
  for (int c = 0; c < 7; c++) {
    const int ilo = c > 3 ? c - 3 : 0, ihi = c < 3 ? c : 3;
    for (int k = 0; k < n; k++) {
      const uint64_t* x = a[k].l;
      const uint64_t* y = b[(size_t)k * stride_b].l;
      for (int i = ilo; i <= ihi; i++) {
        const unsigned __int128 pr = (unsigned __int128)x[i] * y[c - i];
        const unsigned __int128 s = (unsigned __int128)a0 + (uint64_t)pr;
        a0 = (uint64_t)s;
        const unsigned __int128 s1 = (unsigned __int128)a1 + (uint64_t)(pr >> 64) + (uint64_t)(s >> 64);
        a1 = (uint64_t)s1;
        a2 += (uint64_t)(s1 >> 64);
      }
    }
    T[c] = a0;
    a0 = a1;
    a1 = a2;
    a2 = 0;
  }
  T[7] = a0;
  T[8] = a1;
  for (int i = 0; i < 4; i++) {  // T += m r 2^(64 i) clears word i
    const uint64_t m = T[i] * FR_INV;
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (unsigned __int128)m * FR_P[j] + T[i + j];
      T[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int w = i + 4; w < 9; w++) {
      c += T[w];
      T[w] = (uint64_t)c;
      c >>= 64;
    }
  }
  // below n r + r: a few conditional subtractions, the word above 256 bits included
  uint64_t hi = T[8];
  F4 r = {{T[4], T[5], T[6], T[7]}};
  while (hi || geq_p(r.l)) {
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)r.l[i] - FR_P[i] - (uint64_t)bw;
      r.l[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    hi -= (uint64_t)bw;
  }
  return r;
}
inline F4 f4_pow5(const F4& x) {
  const F4 x2 = f4_mul(x, x);
  return f4_mul(f4_mul(x2, x2), x);
}

// The permutation in the optimised schedule of the PSE `poseidon` Spec (the one the chip runs on the device): the first
// round's constants, half - 1 full rounds with folded constants and the MDS matrix, one with the pre-sparse matrix, the
// partial rounds as sparse matrices (2 t - 1 products instead of t^2), the remaining full rounds.  Same function as the
// textbook schedule (tests/test_transcript_cpu.py holds it against an independent Python restatement of that one).
void permute(vdb_transcript* tr) {
  const PoseidonOpt& o = tr->opt;
  const int t = o.t, half = o.half;
  F4 st[16], nx[16];
  const F4* start = reinterpret_cast<const F4*>(o.start.data());
  const F4* mds = reinterpret_cast<const F4*>(o.mds.data());
  const F4* pre = reinterpret_cast<const F4*>(o.pre_sparse.data());
  const F4* srow = reinterpret_cast<const F4*>(o.sparse_row.data());
  const F4* scol = reinterpret_cast<const F4*>(o.sparse_col.data());
  const F4* end = reinterpret_cast<const F4*>(o.end.data());
  const F4* partial = reinterpret_cast<const F4*>(o.partial.data());
  for (int i = 0; i < t; i++) st[i] = f4_add(to_f4(tr->state[i]), start[i]);
  auto dense = [&](const F4* m) {
    for (int i = 0; i < t; i++) nx[i] = f4_dot(m + (size_t)i * t, st, t);
    for (int i = 0; i < t; i++) st[i] = nx[i];
  };
  for (int r = 1; r <= half; r++) {
    for (int i = 0; i < t; i++) st[i] = f4_add(f4_pow5(st[i]), start[(size_t)r * t + i]);
    dense(r < half ? mds : pre);
  }
  for (int p = 0; p < o.rp; p++) {
    st[0] = f4_add(f4_pow5(st[0]), partial[p]);
    const F4 n0 = f4_dot(srow + (size_t)p * t, st, t);
    for (int i = 1; i < t; i++) st[i] = f4_add(f4_mul(st[0], scol[(size_t)p * (t - 1) + i - 1]), st[i]);
    st[0] = n0;
  }
  for (int r = 0; r < half - 1; r++) {
    for (int i = 0; i < t; i++) st[i] = f4_add(f4_pow5(st[i]), end[(size_t)r * t + i]);
    dense(mds);
  }
  for (int i = 0; i < t; i++) st[i] = f4_pow5(st[i]);
  dense(mds);
  for (int i = 0; i < t; i++) tr->state[i] = from_f4(st[i]);
}

void absorb_chunk(vdb_transcript* tr, const u256* in, int n_in) {
  for (int i = 0; i < n_in; i++) tr->state[1 + i] = fr_add(tr->state[1 + i], in[i]);
  if (n_in < tr->rate) tr->state[1 + n_in] = fr_add(tr->state[1 + n_in], mont_one<Fr>());
  permute(tr);
}

// canonical integer of an Fq element (Montgomery in memory) reduced into Fr, Montgomery
u256 fq_to_fr(const u256& a_mont) {
  u256 c = from_mont<Fq>(a_mont);
  const u256 r = mod_p<Fr>();
  while (u256_geq(c, r)) {
    u256 d;
    u256_sub(d, c, r);
    c = d;
  }
  return to_mont<Fr>(c);
}

void put_le(std::vector<uint8_t>& out, const u256& canonical, uint8_t top_flags) {
  for (int w = 0; w < 8; w++)
    for (int b = 0; b < 4; b++) out.push_back((uint8_t)(canonical.w[w] >> (8 * b)));
  out.back() |= top_flags;
}

}  // namespace

extern "C" {

// acc <- acc x + v_i for i = 0 .. n - 1 on the host (the multi-open combines tens of thousands of evaluations with powers of a
// challenge before anything goes back to the device; Python integers cost microseconds apiece)
int vdb_fr_horner(const vdb_fr* values, size_t n, const vdb_fr* x, vdb_fr* acc) {
  VDB_ARG((values || n == 0) && x && acc, "null pointer");
  F4 a, xx;
  memcpy(a.l, acc, 32);
  memcpy(xx.l, x, 32);
  for (size_t i = 0; i < n; i++) {
    F4 v;
    memcpy(v.l, values + i, 32);
    a = f4_add(f4_mul(a, xx), v);
  }
  memcpy(acc, a.l, 32);
  return VDB_OK;
}

int vdb_transcript_new(uint32_t t, uint32_t r_f, uint32_t r_p, vdb_transcript** out) {
  VDB_ARG(out && t >= 2 && t <= 16 && r_f >= 2 && r_f % 2 == 0 && r_f <= 64 && r_p <= 256, "bad argument");
  vdb_transcript* tr = new (std::nothrow) vdb_transcript();
  if (!tr) return VDB_ERR_OOM;
  tr->t = (int)t;
  tr->rate = (int)t - 1;
  tr->r_f = (int)r_f;
  tr->r_p = (int)r_p;
  try {
    poseidon_build_opt(tr->t, tr->r_f, tr->r_p, &tr->opt);
    tr->state.assign(t, u256_zero());
  } catch (...) {
    delete tr;
    return VDB_ERR_OOM;
  }
  u256 cap = u256_zero();
  cap.w[2] = 1;  // 2^64
  tr->state[0] = to_mont<Fr>(cap);
  *out = tr;
  return VDB_OK;
}

void vdb_transcript_free(vdb_transcript* tr) { delete tr; }

int vdb_transcript_set_sign_bit(vdb_transcript* tr, uint32_t bit) {
  VDB_ARG(tr && (bit == 6 || bit == 7), "the y-parity flag of a compressed point sits in bit 6 or bit 7 of the last byte");
  tr->sign_mask = (uint8_t)(1u << bit);
  return VDB_OK;
}

int vdb_transcript_common_scalar(vdb_transcript* tr, const vdb_fr* s) {
  VDB_ARG(tr && s, "null pointer");
  u256 v;
  memcpy(&v, s, 32);
  try {
    tr->buf.push_back(v);
  } catch (...) {  // nothing may propagate across the C ABI
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_common_point(vdb_transcript* tr, const vdb_g1* p) {
  VDB_ARG(tr && p, "null pointer");
  u256 xy[2];
  memcpy(xy, p, 64);
  try {
    tr->buf.push_back(fq_to_fr(xy[0]));
    tr->buf.push_back(fq_to_fr(xy[1]));
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_scalar(vdb_transcript* tr, const vdb_fr* s) {
  int rc = vdb_transcript_common_scalar(tr, s);
  if (rc) return rc;
  u256 v;
  memcpy(&v, s, 32);
  try {
    put_le(tr->bytes, from_mont<Fr>(v), 0);
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_point(vdb_transcript* tr, const vdb_g1* p) {
  int rc = vdb_transcript_common_point(tr, p);
  if (rc) return rc;
  u256 xy[2];
  memcpy(xy, p, 64);
  const u256 x = from_mont<Fq>(xy[0]), y = from_mont<Fq>(xy[1]);
  const bool identity = u256_is_zero(x) && u256_is_zero(y);
  try {
    put_le(tr->bytes, x, (!identity && (y.w[0] & 1)) ? tr->sign_mask : 0);
  } catch (...) {
    return VDB_ERR_OOM;
  }
  return VDB_OK;
}

int vdb_transcript_write_points(vdb_transcript* tr, const vdb_g1* p, size_t n) {
  VDB_ARG(tr && (p || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_write_point(tr, p + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_write_scalars(vdb_transcript* tr, const vdb_fr* s, size_t n) {
  VDB_ARG(tr && (s || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_write_scalar(tr, s + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_common_points(vdb_transcript* tr, const vdb_g1* p, size_t n) {
  VDB_ARG(tr && (p || n == 0), "null pointer");
  for (size_t i = 0; i < n; i++) {
    int rc = vdb_transcript_common_point(tr, p + i);
    if (rc) return rc;
  }
  return VDB_OK;
}

int vdb_transcript_squeeze(vdb_transcript* tr, vdb_fr* out) {
  VDB_ARG(tr && out, "null pointer");
  const size_t n = tr->buf.size();
  for (size_t i = 0; i < n; i += tr->rate) absorb_chunk(tr, tr->buf.data() + i, (int)(n - i < (size_t)tr->rate ? n - i : (size_t)tr->rate));
  if (n % tr->rate == 0) absorb_chunk(tr, nullptr, 0);
  tr->buf.clear();
  memcpy(out, &tr->state[1], 32);
  return VDB_OK;
}

int vdb_transcript_proof_len(const vdb_transcript* tr, size_t* len) {
  VDB_ARG(tr && len, "null pointer");
  *len = tr->bytes.size();
  return VDB_OK;
}

int vdb_transcript_proof_bytes(const vdb_transcript* tr, uint8_t* out, size_t cap) {
  VDB_ARG(tr && (out || tr->bytes.empty()), "null pointer");
  VDB_ARG(cap >= tr->bytes.size(), "buffer too small");
  if (!tr->bytes.empty()) memcpy(out, tr->bytes.data(), tr->bytes.size());
  return VDB_OK;
}

}  // extern "C"
