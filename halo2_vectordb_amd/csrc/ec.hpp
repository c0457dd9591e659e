// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for the MSM kernels.
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed addition costs 8M + 2S and needs no inversion.  Identity: ZZ == 0.
// All exceptional cases (equal points, opposite points, identity operands) are handled exactly —
// the result must be the same group element the reference's `best_multiexp` returns.
#pragma once
#include "field.hpp"

namespace vdb {

struct alignas(16) Affine {
  u256 x, y;  // Montgomery Fq; identity encoded as (0, 0) like halo2curves
};
struct XYZZ {
  u256 x, y, zz, zzz;
};

HD bool affine_is_identity(const Affine& p) { return u256_is_zero(p.x) && u256_is_zero(p.y); }
HD bool xyzz_is_identity(const XYZZ& p) { return u256_is_zero(p.zz); }
HD XYZZ xyzz_identity() {
  XYZZ r;
  r.x = u256_zero();
  r.y = u256_zero();
  r.zz = u256_zero();
  r.zzz = u256_zero();
  return r;
}
HD XYZZ xyzz_from_affine(const Affine& p) {
  XYZZ r;
  if (affine_is_identity(p)) return xyzz_identity();
  r.x = p.x;
  r.y = p.y;
  r.zz = mont_one<Fq>();
  r.zzz = r.zz;
  return r;
}
// 2 * (affine p)   [mdbl-2008-s-1]
HD XYZZ xyzz_double_affine(const Affine& p) {
  XYZZ r;
  if (affine_is_identity(p) || u256_is_zero(p.y)) return xyzz_identity();
  u256 u = fq_add(p.y, p.y);
  u256 v = fq_sqr(u);
  u256 w = fq_mul(u, v);
  u256 s = fq_mul(p.x, v);
  u256 xx = fq_sqr(p.x);
  u256 m = fq_add(fq_add(xx, xx), xx);
  r.x = fq_sub(fq_sqr(m), fq_add(s, s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}
// 2 * p   [dbl-2008-s-1, a = 0]
HD XYZZ xyzz_double(const XYZZ& p) {
  XYZZ r;
  if (xyzz_is_identity(p) || u256_is_zero(p.y)) return xyzz_identity();
  u256 u = fq_add(p.y, p.y);
  u256 v = fq_sqr(u);
  u256 w = fq_mul(u, v);
  u256 s = fq_mul(p.x, v);
  u256 xx = fq_sqr(p.x);
  u256 m = fq_add(fq_add(xx, xx), xx);
  r.x = fq_sub(fq_sqr(m), fq_add(s, s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
  r.zz = fq_mul(v, p.zz);
  r.zzz = fq_mul(w, p.zzz);
  return r;
}
// acc += (neg ? -q : q)   [madd-2008-s]
HD void xyzz_add_mixed(XYZZ& acc, const Affine& q, bool neg) {
  if (affine_is_identity(q)) return;
  u256 qy = neg ? fq_neg(q.y) : q.y;
  if (xyzz_is_identity(acc)) {
    acc.x = q.x;
    acc.y = qy;
    acc.zz = mont_one<Fq>();
    acc.zzz = acc.zz;
    return;
  }
  u256 u2 = fq_mul(q.x, acc.zz);
  u256 s2 = fq_mul(qy, acc.zzz);
  u256 p = fq_sub(u2, acc.x);
  u256 r = fq_sub(s2, acc.y);
  if (u256_is_zero(p)) {
    if (u256_is_zero(r)) {
      Affine t;
      t.x = q.x;
      t.y = qy;
      acc = xyzz_double_affine(t);
    } else {
      acc = xyzz_identity();
    }
    return;
  }
  u256 pp = fq_sqr(p);
  u256 ppp = fq_mul(p, pp);
  u256 qq = fq_mul(acc.x, pp);
  u256 x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_add(qq, qq));
  u256 y3 = fq_sub(fq_mul(r, fq_sub(qq, x3)), fq_mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fq_mul(acc.zz, pp);
  acc.zzz = fq_mul(acc.zzz, ppp);
}
// acc += b   [add-2008-s]
HD void xyzz_add(XYZZ& acc, const XYZZ& b) {
  if (xyzz_is_identity(b)) return;
  if (xyzz_is_identity(acc)) {
    acc = b;
    return;
  }
  u256 u1 = fq_mul(acc.x, b.zz);
  u256 u2 = fq_mul(b.x, acc.zz);
  u256 s1 = fq_mul(acc.y, b.zzz);
  u256 s2 = fq_mul(b.y, acc.zzz);
  u256 p = fq_sub(u2, u1);
  u256 r = fq_sub(s2, s1);
  if (u256_is_zero(p)) {
    if (u256_is_zero(r)) acc = xyzz_double(acc);
    else acc = xyzz_identity();
    return;
  }
  u256 pp = fq_sqr(p);
  u256 ppp = fq_mul(p, pp);
  u256 qq = fq_mul(u1, pp);
  u256 x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_add(qq, qq));
  u256 y3 = fq_sub(fq_mul(r, fq_sub(qq, x3)), fq_mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fq_mul(fq_mul(acc.zz, b.zz), pp);
  acc.zzz = fq_mul(fq_mul(acc.zzz, b.zzz), ppp);
}
// canonical affine (identity -> (0,0)); one Fermat inversion
HD Affine xyzz_to_affine(const XYZZ& p) {
  Affine a;
  if (xyzz_is_identity(p)) {
    a.x = u256_zero();
    a.y = u256_zero();
    return a;
  }
  u256 t = mont_inv<Fq>(fq_mul(p.zz, p.zzz));
  a.x = fq_mul(p.x, fq_mul(t, p.zzz));  // X / ZZ
  a.y = fq_mul(p.y, fq_mul(t, p.zz));   // Y / ZZZ
  return a;
}

__device__ __forceinline__ Affine ld_affine(const Affine* p) {
  Affine a;
  a.x = ld256(&p->x);
  a.y = ld256(&p->y);
  return a;
}
__device__ __forceinline__ void st_affine(Affine* p, const Affine& a) {
  st256(&p->x, a.x);
  st256(&p->y, a.y);
}
__device__ __forceinline__ XYZZ ld_xyzz(const XYZZ* p) {
  XYZZ a;
  a.x = ld256(&p->x);
  a.y = ld256(&p->y);
  a.zz = ld256(&p->zz);
  a.zzz = ld256(&p->zzz);
  return a;
}
__device__ __forceinline__ void st_xyzz(XYZZ* p, const XYZZ& a) {
  st256(&p->x, a.x);
  st256(&p->y, a.y);
  st256(&p->zz, a.zz);
  st256(&p->zzz, a.zzz);
}

}  // namespace vdb
