// BLS12-381 G1 arithmetic for gfx950 (replaces blst_p1_* behind src/bls.rs:362-552).
//
// Accumulators use extended Jacobian "XYZZ" coordinates (X, Y, ZZ, ZZZ) with
// x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0.  A mixed addition
// (XYZZ += affine) costs 8M + 2S and is the unit of work of every MSM here.
// All additions are COMPLETE (identity, P+P and P+(-P) handled) -- SURVEY.md
// section 7.2 "Completeness of the adder".
#pragma once
#include "../../include/kateth_amd.h"  // KZG_OK / KZG_ERR_* status codes
#include "field.cuh"

namespace kzg {

struct g1_affine {  // Montgomery coordinates; infinity encoded as inf != 0
  fp_t x, y;
};

struct g1_xyzz {
  fp_t x, y, zz, zzz;
};

KZG_HD void xyzz_set_inf(g1_xyzz& p) {
  bn_zero(p.x);
  bn_zero(p.y);
  bn_zero(p.zz);
  bn_zero(p.zzz);
}
KZG_HD bool xyzz_is_inf(const g1_xyzz& p) { return bn_is_zero(p.zz); }

KZG_HD void xyzz_from_affine(g1_xyzz& p, const fp_t& x, const fp_t& y) {
  p.x = x;
  p.y = y;
  p.zz = fp_one();
  p.zzz = fp_one();
}

// p = 2*(x, y)   (mdbl-2008-s-1, a = 0).  (x,y) must not be infinity; y == 0
// cannot happen on this curve's prime-order subgroup but is handled (-> inf).
KZG_HD void xyzz_mdbl_inl(g1_xyzz& p, const fp_t& x, const fp_t& y) {
  if (bn_is_zero(y)) {
    xyzz_set_inf(p);
    return;
  }
  fp_t u, v, w, s, m, t;
  fp_dbl(u, y);
  fp_sqr(v, u);
  fp_mul(w, u, v);
  fp_mul(s, x, v);
  fp_sqr(m, x);
  fp_dbl(t, m);
  fp_add(m, m, t);  // 3x^2
  fp_sqr(p.x, m);
  fp_sub(p.x, p.x, s);
  fp_sub(p.x, p.x, s);
  fp_sub(t, s, p.x);
  fp_mul(t, m, t);
  fp_mul(u, w, y);
  fp_sub(p.y, t, u);
  p.zz = v;
  p.zzz = w;
}
KZG_HD_NOINLINE void xyzz_mdbl(g1_xyzz& p, const fp_t& x, const fp_t& y) { xyzz_mdbl_inl(p, x, y); }

// p = 2*p  (dbl-2008-s-1)
KZG_HD_NOINLINE void xyzz_dbl(g1_xyzz& p) {
  if (xyzz_is_inf(p)) return;
  if (bn_is_zero(p.y)) {
    xyzz_set_inf(p);
    return;
  }
  fp_t u, v, w, s, m, t;
  fp_dbl(u, p.y);
  fp_sqr(v, u);
  fp_mul(w, u, v);
  fp_mul(s, p.x, v);
  fp_sqr(m, p.x);
  fp_dbl(t, m);
  fp_add(m, m, t);
  fp_t x3, y3;
  fp_sqr(x3, m);
  fp_sub(x3, x3, s);
  fp_sub(x3, x3, s);
  fp_sub(t, s, x3);
  fp_mul(t, m, t);
  fp_mul(u, w, p.y);
  fp_sub(y3, t, u);
  fp_mul(p.zz, v, p.zz);
  fp_mul(p.zzz, w, p.zzz);
  p.x = x3;
  p.y = y3;
}

// p += (x2, y2)  with (x2, y2) a finite affine point   (madd-2008-s)
KZG_HD void xyzz_madd(g1_xyzz& p, const fp_t& x2, const fp_t& y2) {
  if (xyzz_is_inf(p)) {
    xyzz_from_affine(p, x2, y2);
    return;
  }
  fp_t u2, s2, pp, ppp, q, r, t;
  fp_mul(u2, x2, p.zz);
  fp_mul(s2, y2, p.zzz);
  fp_sub(u2, u2, p.x);  // P
  fp_sub(r, s2, p.y);   // R
  if (bn_is_zero(u2)) {
    // rare (P == +-Q): handled inline -- an out-of-line call would take the address of the
    // caller's accumulator and operands and force them into scratch memory in the hot loop
    if (bn_is_zero(r))
      xyzz_mdbl_inl(p, x2, y2);  // P + P
    else
      xyzz_set_inf(p);  // P + (-P)
    return;
  }
  fp_sqr(pp, u2);
  fp_mul(ppp, u2, pp);
  fp_mul(q, p.x, pp);
  fp_sqr(t, r);
  fp_sub(t, t, ppp);
  fp_sub(t, t, q);
  fp_sub(t, t, q);  // X3
  fp_sub(q, q, t);
  fp_mul(q, r, q);
  fp_mul(s2, p.y, ppp);
  fp_sub(p.y, q, s2);
  p.x = t;
  fp_mul(p.zz, p.zz, pp);
  fp_mul(p.zzz, p.zzz, ppp);
}

// p += (x2, y2) with the accumulator kept in LAZY form (coordinates in [0, 2p)); (x2, y2) canonical.
// Same formulas as xyzz_madd; saves the final conditional subtraction of all ten multiplies.
// Used only by the fixed-base MSM hot loop; callers canonicalise the accumulator afterwards.
KZG_HD void xyzz_madd_lazy(g1_xyzz& p, const fp_t& x2, const fp_t& y2) {
  if (bn_is_zero(p.zz) && bn_is_zero(p.zzz)) {  // identity (only ever set exactly to zero)
    xyzz_from_affine(p, x2, y2);
    return;
  }
  fp_t u2, s2, pp, ppp, q, r, t;
  mont_mul_lazy<FpParams>(u2, x2, p.zz);
  mont_mul_lazy<FpParams>(s2, y2, p.zzz);
  sub_lazy<FpParams>(u2, u2, p.x);  // P
  sub_lazy<FpParams>(r, s2, p.y);   // R
  if (is_zero_lazy<FpParams>(u2)) {  // rare: P == +-Q
    if (is_zero_lazy<FpParams>(r))
      xyzz_mdbl_inl(p, x2, y2);
    else
      xyzz_set_inf(p);
    return;
  }
  mont_mul_lazy<FpParams>(pp, u2, u2);
  mont_mul_lazy<FpParams>(ppp, u2, pp);
  mont_mul_lazy<FpParams>(q, p.x, pp);
  mont_mul_lazy<FpParams>(t, r, r);
  sub_lazy<FpParams>(t, t, ppp);
  sub_lazy<FpParams>(t, t, q);
  sub_lazy<FpParams>(t, t, q);  // X3
  sub_lazy<FpParams>(q, q, t);
  mont_mul_lazy<FpParams>(q, r, q);
  mont_mul_lazy<FpParams>(s2, p.y, ppp);
  sub_lazy<FpParams>(p.y, q, s2);
  p.x = t;
  mont_mul_lazy<FpParams>(p.zz, p.zz, pp);
  mont_mul_lazy<FpParams>(p.zzz, p.zzz, ppp);
}
KZG_HD void xyzz_canonicalize(g1_xyzz& p) {
  canonicalize<FpParams>(p.x);
  canonicalize<FpParams>(p.y);
  canonicalize<FpParams>(p.zz);
  canonicalize<FpParams>(p.zzz);
}

// p += q   (add-2008-s), complete
KZG_HD_NOINLINE void xyzz_add(g1_xyzz& p, const g1_xyzz& q) {
  if (xyzz_is_inf(q)) return;
  if (xyzz_is_inf(p)) {
    p = q;
    return;
  }
  fp_t u1, u2, s1, s2, pp, ppp, qq, r, t;
  fp_mul(u1, p.x, q.zz);
  fp_mul(u2, q.x, p.zz);
  fp_mul(s1, p.y, q.zzz);
  fp_mul(s2, q.y, p.zzz);
  fp_sub(u2, u2, u1);  // P
  fp_sub(r, s2, s1);   // R
  if (bn_is_zero(u2)) {
    if (bn_is_zero(r))
      xyzz_dbl(p);
    else
      xyzz_set_inf(p);
    return;
  }
  fp_sqr(pp, u2);
  fp_mul(ppp, u2, pp);
  fp_mul(qq, u1, pp);
  fp_sqr(t, r);
  fp_sub(t, t, ppp);
  fp_sub(t, t, qq);
  fp_sub(t, t, qq);  // X3
  fp_sub(qq, qq, t);
  fp_mul(qq, r, qq);
  fp_mul(s1, s1, ppp);
  fp_sub(p.y, qq, s1);
  p.x = t;
  fp_mul(p.zz, p.zz, q.zz);
  fp_mul(p.zz, p.zz, pp);
  fp_mul(p.zzz, p.zzz, q.zzz);
  fp_mul(p.zzz, p.zzz, ppp);
}

KZG_HD void xyzz_neg(g1_xyzz& p) { fp_neg(p.y, p.y); }

// XYZZ -> affine (Montgomery).  Returns false for infinity.
KZG_HD_NOINLINE bool xyzz_to_affine(fp_t& x, fp_t& y, const g1_xyzz& p) {
  if (xyzz_is_inf(p)) return false;
  fp_t t, ti, a;
  fp_mul(t, p.zz, p.zzz);
  fp_inv(ti, t);
  fp_mul(a, ti, p.zzz);  // 1/ZZ
  fp_mul(x, p.x, a);
  fp_mul(a, ti, p.zz);  // 1/ZZZ
  fp_mul(y, p.y, a);
  return true;
}

// ---------------------------------------------------------------------------
// ZCash compressed encoding (blst_p1_compress / blst_p1_uncompress,
// src/bls.rs:491-531).  48 B big-endian x; byte0 bit7 = compressed, bit6 =
// infinity, bit5 = y lexicographically larger than -y.
// ---------------------------------------------------------------------------
// plain (non-Montgomery) y > (p-1)/2 ?
KZG_HD bool fp_is_lex_larger_plain(const fp_t& y_plain) {
  fp_t half, t;
#pragma unroll
  for (int i = 0; i < 12; i++) half.v[i] = FpParams::half(i);
  return bn_sub(t, half, y_plain) != 0;  // half < y
}

KZG_HD void g1_compress_affine(uint8_t* out48, const fp_t& x_mont, const fp_t& y_mont, bool inf) {
  if (inf) {
    out48[0] = 0xC0;
    for (int i = 1; i < 48; i++) out48[i] = 0;
    return;
  }
  fp_t xp, yp;
  from_mont<FpParams>(xp, x_mont);
  from_mont<FpParams>(yp, y_mont);
  fp_to_be_bytes_plain(out48, xp);
  out48[0] |= 0x80;
  if (fp_is_lex_larger_plain(yp)) out48[0] |= 0x20;
}

KZG_HD void g1_compress_xyzz(uint8_t* out48, const g1_xyzz& p) {
  fp_t x, y;
  bool finite = xyzz_to_affine(x, y, p);
  g1_compress_affine(out48, x, y, !finite);
}

// blst_p1_uncompress: no subgroup check.  *inf set for the point at infinity.
KZG_HD_NOINLINE int32_t g1_uncompress(fp_t& x, fp_t& y, bool& inf, const uint8_t* in48) {
  inf = false;
  uint8_t b0 = in48[0];
  if (!(b0 & 0x80)) return KZG_ERR_EC_INVALID_ENCODING;
  if (b0 & 0x40) {
    uint32_t o = b0 & 0x3F;
    for (int i = 1; i < 48; i++) o |= in48[i];
    if (o) return KZG_ERR_EC_INVALID_ENCODING;
    inf = true;
    bn_zero(x);
    bn_zero(y);
    return KZG_OK;
  }
  fp_t xp;
  fp_from_be_bytes_plain(xp, in48);
  xp.v[11] &= 0x1FFFFFFFu;
  if (bn_geq(xp, modulus<FpParams>())) return KZG_ERR_EC_INVALID_ENCODING;
  to_mont<FpParams>(x, xp);
  fp_t rhs, t, b;
  fp_sqr(t, x);
  fp_mul(rhs, t, x);
  {
    const uint32_t bm[12] = KZG_FP_B_MONT;
#pragma unroll
    for (int i = 0; i < 12; i++) b.v[i] = bm[i];
  }
  fp_add(rhs, rhs, b);
  fp_sqrt_candidate(y, rhs);
  fp_sqr(t, y);
  if (!bn_eq(t, rhs)) return KZG_ERR_EC_NOT_ON_CURVE;
  fp_t yp;
  from_mont<FpParams>(yp, y);
  bool larger = fp_is_lex_larger_plain(yp);
  if (((b0 & 0x20) != 0) != larger) fp_neg(y, y);
  return KZG_OK;
}

// prime-order subgroup test by the definition [r]P == O (kept as the cross-check of the
// fast test below; tests/test_hostmath.py compares the two).  Infinity is in the group.
KZG_HD_NOINLINE bool g1_in_subgroup_naive(const fp_t& x, const fp_t& y, bool inf) {
  if (inf) return true;
  const uint32_t rr[8] = KZG_FR_MOD_PLAIN;
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int i = 254; i >= 0; i--) {
    xyzz_dbl(acc);
    if ((rr[i >> 5] >> (i & 31)) & 1u) xyzz_madd(acc, x, y);
  }
  return xyzz_is_inf(acc);
}

// blst_p1_affine_in_g1 (src/bls.rs:522) via the GLV endomorphism (Scott, eprint 2021/1130):
// phi(x, y) = (beta*x, y) satisfies phi^2 + phi + 1 = 0 on E and acts on G1 as
// multiplication by lambda = -z^2 (lambda^2 + lambda + 1 = z^4 - z^2 + 1 = r).  If
// phi(P) = [-z^2]P for a point P = Q + T (Q in G1, T of order dividing the cofactor),
// then ord(T) divides z^4 - z^2 + 1 = r, and gcd(cofactor, r) = 1 forces T = O.
// So  P in G1  <=>  (beta*x, y) == -[z^2]P : two multiplications by the 64-bit |z|
// (126 doublings + 10 additions) instead of a 255-bit ladder.
KZG_HD_NOINLINE void g1_mul_by_z(g1_xyzz& out, const g1_xyzz& base) {
  const uint64_t zabs = 0xd201000000010000ull;
  g1_xyzz acc = base;
  for (int i = 62; i >= 0; i--) {
    xyzz_dbl(acc);
    if ((zabs >> i) & 1ull) xyzz_add(acc, base);
  }
  out = acc;
}
KZG_HD_NOINLINE bool g1_in_subgroup(const fp_t& x, const fp_t& y, bool inf) {
  if (inf) return true;
  g1_xyzz p, q1, q2;
  xyzz_from_affine(p, x, y);
  g1_mul_by_z(q1, p);
  g1_mul_by_z(q2, q1);  // [z^2]P
  if (xyzz_is_inf(q2)) return false;
  fp_t beta, t;
  {
    const uint32_t bm[12] = KZG_FP_BETA_MONT;
#pragma unroll
    for (int i = 0; i < 12; i++) beta.v[i] = bm[i];
  }
  fp_mul(t, x, beta);
  fp_mul(t, t, q2.zz);
  if (!bn_eq(t, q2.x)) return false;  // x-coordinates: beta*x == X/ZZ
  fp_mul(t, y, q2.zzz);
  fp_neg(t, t);
  return bn_eq(t, q2.y);  // y == -(Y/ZZZ)
}

// Decompress for P1 (src/bls.rs:505-531): uncompress + subgroup check
KZG_HD int32_t g1_decompress(fp_t& x, fp_t& y, bool& inf, const uint8_t* in48) {
  int32_t st = g1_uncompress(x, y, inf, in48);
  if (st != KZG_OK) return st;
  if (!g1_in_subgroup(x, y, inf)) return KZG_ERR_EC_NOT_IN_GROUP;
  return KZG_OK;
}

// acc = k * (x, y), k a plain 255-bit scalar (blst_p1_mult, src/bls.rs:474-489)
KZG_HD_NOINLINE void g1_mul_affine(g1_xyzz& acc, const fp_t& x, const fp_t& y, const fr_t& k_plain) {
  xyzz_set_inf(acc);
  for (int i = 254; i >= 0; i--) {
    xyzz_dbl(acc);
    if ((k_plain.v[i >> 5] >> (i & 31)) & 1u) xyzz_madd(acc, x, y);
  }
}

}  // namespace kzg
