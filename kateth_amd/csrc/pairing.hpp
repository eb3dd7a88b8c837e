// Host-side BLS12-381 pairing check for gfx950's companion CPU:
// replaces bls::verify_pairings (src/bls.rs:572-598 -> blst_miller_loop x2,
// blst_fp12_mul, blst_final_exp, blst_fp12_is_one) and the G2 side of
// Setup::load_json (P2::decompress, src/kzg/setup.rs:67-72).
//
// This is the one piece of arithmetic the engine runs on the host: it is
// executed ONCE per verification call (not per blob) on two G1 points the GPU
// has already reduced the whole batch to -- SURVEY.md section 2b (K9) scopes it
// as host code.  It is not a fallback for anything: there is no device version.
//
// Both pairings of every KZG check have a FIXED G2 argument (the generator and
// [tau]_2 = g2_monomial[1]):
//     e(A, [tau]_2) == e(B, G2)       <=>   e(-A, [tau]_2) * e(B, G2) == 1
// so the Miller-loop line coefficients (slope lambda_k and mu_k = lambda_k*x_T - y_T
// for every doubling/addition step of T) are computed once at context creation,
// and a verification runs an inversion-free two-pairing Miller loop sharing the
// Fp12 squarings, followed by one final exponentiation.
//
// Tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - xi), xi = 1+u, Fp12 = Fp6[w]/(w^2 - v).
// Built on the single-source field.cuh (host instantiation, 32-bit limbs).
#pragma once
#include <vector>

#include "g1.cuh"

namespace kzg {
namespace host {

struct fp2 {
  fp_t c0, c1;
};
struct fp6 {
  fp2 c0, c1, c2;
};
struct fp12 {
  fp6 c0, c1;
};

inline fp_t fp_zero() {
  fp_t z;
  bn_zero(z);
  return z;
}
inline fp2 f2_zero() { return fp2{fp_zero(), fp_zero()}; }
inline fp2 f2_one() { return fp2{fp_one(), fp_zero()}; }
inline bool f2_is_zero(const fp2& a) { return bn_is_zero(a.c0) && bn_is_zero(a.c1); }
inline bool f2_eq(const fp2& a, const fp2& b) { return bn_eq(a.c0, b.c0) && bn_eq(a.c1, b.c1); }
inline fp2 f2_add(const fp2& a, const fp2& b) {
  fp2 r;
  fp_add(r.c0, a.c0, b.c0);
  fp_add(r.c1, a.c1, b.c1);
  return r;
}
inline fp2 f2_sub(const fp2& a, const fp2& b) {
  fp2 r;
  fp_sub(r.c0, a.c0, b.c0);
  fp_sub(r.c1, a.c1, b.c1);
  return r;
}
inline fp2 f2_neg(const fp2& a) {
  fp2 r;
  fp_neg(r.c0, a.c0);
  fp_neg(r.c1, a.c1);
  return r;
}
inline fp2 f2_conj(const fp2& a) {
  fp2 r = a;
  fp_neg(r.c1, a.c1);
  return r;
}
inline fp2 f2_mul(const fp2& a, const fp2& b) {  // Karatsuba, 3 Fp mults
  fp_t t0, t1, s0, s1, t2;
  fp_mul(t0, a.c0, b.c0);
  fp_mul(t1, a.c1, b.c1);
  fp_add(s0, a.c0, a.c1);
  fp_add(s1, b.c0, b.c1);
  fp_mul(t2, s0, s1);
  fp2 r;
  fp_sub(r.c0, t0, t1);
  fp_sub(t2, t2, t0);
  fp_sub(r.c1, t2, t1);
  return r;
}
inline fp2 f2_sqr(const fp2& a) {  // (a0+a1)(a0-a1), 2 a0 a1
  fp_t s, d, m;
  fp_add(s, a.c0, a.c1);
  fp_sub(d, a.c0, a.c1);
  fp_mul(m, a.c0, a.c1);
  fp2 r;
  fp_mul(r.c0, s, d);
  fp_dbl(r.c1, m);
  return r;
}
inline fp2 f2_mul_fp(const fp2& a, const fp_t& s) {
  fp2 r;
  fp_mul(r.c0, a.c0, s);
  fp_mul(r.c1, a.c1, s);
  return r;
}
inline fp2 f2_mul_xi(const fp2& a) {  // (1+u)(a0 + a1 u) = (a0 - a1) + (a0 + a1) u
  fp2 r;
  fp_sub(r.c0, a.c0, a.c1);
  fp_add(r.c1, a.c0, a.c1);
  return r;
}
inline fp2 f2_inv(const fp2& a) {
  fp_t n, t, ni;
  fp_sqr(n, a.c0);
  fp_sqr(t, a.c1);
  fp_add(n, n, t);
  fp_inv(ni, n);
  fp2 r;
  fp_mul(r.c0, a.c0, ni);
  fp_mul(t, a.c1, ni);
  fp_neg(r.c1, t);
  return r;
}
inline fp2 f2_dbl(const fp2& a) { return f2_add(a, a); }

// a^e, e = plain little-endian 32-bit limbs
inline fp2 f2_pow(const fp2& a, const uint32_t* e, int nlimbs) {
  fp2 r = f2_one();
  bool started = false;
  for (int i = nlimbs * 32 - 1; i >= 0; i--) {
    if (started) r = f2_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) {
      r = started ? f2_mul(r, a) : a;
      started = true;
    }
  }
  return r;
}

// ---- Fp6 ---------------------------------------------------------------------
inline fp6 f6_zero() { return fp6{f2_zero(), f2_zero(), f2_zero()}; }
inline fp6 f6_one() { return fp6{f2_one(), f2_zero(), f2_zero()}; }
inline fp6 f6_add(const fp6& a, const fp6& b) { return fp6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline fp6 f6_sub(const fp6& a, const fp6& b) { return fp6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline fp6 f6_neg(const fp6& a) { return fp6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline fp6 f6_mul(const fp6& a, const fp6& b) {  // Karatsuba-style, 6 Fp2 mults
  fp2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
  fp2 c0 = f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), t1), t2);
  c0 = f2_add(f2_mul_xi(c0), t0);
  fp2 c1 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), t0), t1);
  c1 = f2_add(c1, f2_mul_xi(t2));
  fp2 c2 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), t0), t2);
  c2 = f2_add(c2, t1);
  return fp6{c0, c1, c2};
}
inline fp6 f6_mul_by_v(const fp6& a) { return fp6{f2_mul_xi(a.c2), a.c0, a.c1}; }
inline fp6 f6_inv(const fp6& a) {
  fp2 t0 = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
  fp2 t1 = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
  fp2 t2 = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
  fp2 d = f2_add(f2_mul(a.c0, t0), f2_mul_xi(f2_add(f2_mul(a.c2, t1), f2_mul(a.c1, t2))));
  fp2 di = f2_inv(d);
  return fp6{f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)};
}

// ---- Fp12 --------------------------------------------------------------------
inline fp12 f12_one() { return fp12{f6_one(), f6_zero()}; }
inline fp12 f12_mul(const fp12& a, const fp12& b) {
  fp6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
  fp6 c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
  return fp12{f6_add(t0, f6_mul_by_v(t1)), c1};
}
inline fp12 f12_sqr(const fp12& a) {  // complex squaring: 2 Fp6 mults
  fp6 ab = f6_mul(a.c0, a.c1);
  fp6 s = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_by_v(a.c1)));
  fp6 c0 = f6_sub(f6_sub(s, ab), f6_mul_by_v(ab));
  return fp12{c0, f6_add(ab, ab)};
}
inline fp12 f12_conj(const fp12& a) { return fp12{a.c0, f6_neg(a.c1)}; }
inline fp12 f12_inv(const fp12& a) {
  fp6 d = f6_sub(f6_mul(a.c0, a.c0), f6_mul_by_v(f6_mul(a.c1, a.c1)));
  fp6 di = f6_inv(d);
  return fp12{f6_mul(a.c0, di), f6_neg(f6_mul(a.c1, di))};
}
inline bool f12_is_one(const fp12& a) {
  return f2_eq(a.c0.c0, f2_one()) && f2_is_zero(a.c0.c1) && f2_is_zero(a.c0.c2) && f2_is_zero(a.c1.c0) && f2_is_zero(a.c1.c1) &&
         f2_is_zero(a.c1.c2);
}
inline bool f12_eq(const fp12& a, const fp12& b) {
  return f2_eq(a.c0.c0, b.c0.c0) && f2_eq(a.c0.c1, b.c0.c1) && f2_eq(a.c0.c2, b.c0.c2) && f2_eq(a.c1.c0, b.c1.c0) &&
         f2_eq(a.c1.c1, b.c1.c1) && f2_eq(a.c1.c2, b.c1.c2);
}
// multiply by the sparse line value  l = l0 + l2*w^2 + l3*w^3  (w^2 = v, w^3 = v*w):
// as an Fp12 element: c0 = (l0, l2, 0), c1 = (0, l3, 0)
// Sparse products (13 Fp2 multiplications instead of 18, three of them by an Fp scalar): a * (c0 + c1 v) in Fp6 ...
inline fp6 f6_mul_by_01(const fp6& a, const fp2& c0, const fp2& c1) {
  const fp2 aa = f2_mul(a.c0, c0), bb = f2_mul(a.c1, c1);
  const fp2 t1 = f2_add(f2_mul_xi(f2_mul(a.c2, c1)), aa);
  const fp2 t2 = f2_sub(f2_sub(f2_mul(f2_add(c0, c1), f2_add(a.c0, a.c1)), aa), bb);
  const fp2 t3 = f2_add(f2_mul(a.c2, c0), bb);
  return fp6{t1, t2, t3};
}
// ... and a * (s v) with s in Fp
inline fp6 f6_mul_by_1_fp(const fp6& a, const fp_t& s) { return fp6{f2_mul_xi(f2_mul_fp(a.c2, s)), f2_mul_fp(a.c0, s), f2_mul_fp(a.c1, s)}; }
// f * (l0 + l2 v + y v w), y in Fp: the line value with the G1 point's y as its w^3 coefficient
inline fp12 f12_mul_by_line(const fp12& f, const fp2& l0, const fp2& l2, const fp_t& y) {
  const fp6 aa = f6_mul_by_01(f.c0, l0, l2);
  const fp6 bb = f6_mul_by_1_fp(f.c1, y);
  fp2 o = l2;
  fp_add(o.c0, o.c0, y);  // l2 + y (y has no u part)
  const fp6 c1 = f6_sub(f6_sub(f6_mul_by_01(f6_add(f.c1, f.c0), l0, o), aa), bb);
  const fp6 c0 = f6_add(f6_mul_by_v(bb), aa);
  return fp12{c0, c1};
}
// Squaring in the cyclotomic subgroup (Granger-Scott): valid for unitary f with f^(p^4 - p^2 + 1) = 1, i.e. anything
// after the easy part of the final exponentiation.  9 Fp2 squarings instead of the 12 Fp2 products of f12_sqr.
inline void f4_sqr(fp2& c0, fp2& c1, const fp2& a, const fp2& b) {
  const fp2 t0 = f2_sqr(a), t1 = f2_sqr(b);
  c0 = f2_add(f2_mul_xi(t1), t0);
  c1 = f2_sub(f2_sub(f2_sqr(f2_add(a, b)), t0), t1);
}
inline fp12 f12_cyclotomic_sqr(const fp12& f) {
  fp2 z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
  fp2 t0, t1, t2, t3;
  f4_sqr(t0, t1, z0, z1);
  z0 = f2_sub(t0, z0);
  z0 = f2_add(f2_dbl(z0), t0);
  z1 = f2_add(t1, z1);
  z1 = f2_add(f2_dbl(z1), t1);
  f4_sqr(t0, t1, z2, z3);
  f4_sqr(t2, t3, z4, z5);
  z4 = f2_sub(t0, z4);
  z4 = f2_add(f2_dbl(z4), t0);
  z5 = f2_add(t1, z5);
  z5 = f2_add(f2_dbl(z5), t1);
  t0 = f2_mul_xi(t3);
  z2 = f2_add(t0, z2);
  z2 = f2_add(f2_dbl(z2), t0);
  z3 = f2_sub(t2, z3);
  z3 = f2_add(f2_dbl(z3), t2);
  return fp12{fp6{z0, z4, z3}, fp6{z2, z1, z5}};
}

// ---- Frobenius ---------------------------------------------------------------
struct frob_consts {
  fp2 g[6];  // g[i] = xi^(i*(p-1)/6)
};
inline frob_consts make_frob_consts() {
  // e = (p-1)/6 as plain limbs
  fp_t pm1 = modulus<FpParams>();
  pm1.v[0] -= 1;  // p is odd and its low limb is non-zero
  uint32_t e[12];
  uint64_t rem = 0;
  for (int i = 11; i >= 0; i--) {
    uint64_t cur = (rem << 32) | pm1.v[i];
    e[i] = (uint32_t)(cur / 6);
    rem = cur % 6;
  }
  fp2 xi{fp_one(), fp_one()};
  frob_consts c;
  c.g[0] = f2_one();
  c.g[1] = f2_pow(xi, e, 12);
  for (int i = 2; i < 6; i++) c.g[i] = f2_mul(c.g[i - 1], c.g[1]);
  return c;
}
inline fp12 f12_frobenius(const fp12& a, const frob_consts& k) {
  fp12 r;
  r.c0.c0 = f2_conj(a.c0.c0);
  r.c0.c1 = f2_mul(f2_conj(a.c0.c1), k.g[2]);
  r.c0.c2 = f2_mul(f2_conj(a.c0.c2), k.g[4]);
  r.c1.c0 = f2_mul(f2_conj(a.c1.c0), k.g[1]);
  r.c1.c1 = f2_mul(f2_conj(a.c1.c1), k.g[3]);
  r.c1.c2 = f2_mul(f2_conj(a.c1.c2), k.g[5]);
  return r;
}

// ---- G2 (affine over Fp2, E': y^2 = x^3 + 4(1+u)) ---------------------------------
struct g2_affine {
  fp2 x, y;
  bool inf;
};
inline fp2 g2_b() {
  fp_t four;
  bn_zero(four);
  four.v[0] = 4;
  to_mont<FpParams>(four, four);
  return fp2{four, four};
}
inline bool g2_on_curve(const g2_affine& p) {
  if (p.inf) return true;
  fp2 lhs = f2_sqr(p.y), rhs = f2_add(f2_mul(f2_sqr(p.x), p.x), g2_b());
  return f2_eq(lhs, rhs);
}
inline g2_affine g2_add(const g2_affine& a, const g2_affine& b) {  // complete affine addition (setup time only)
  if (a.inf) return b;
  if (b.inf) return a;
  fp2 lam;
  if (f2_eq(a.x, b.x)) {
    if (f2_is_zero(f2_add(a.y, b.y))) return g2_affine{f2_zero(), f2_zero(), true};
    fp2 x2 = f2_sqr(a.x);
    lam = f2_mul(f2_add(f2_dbl(x2), x2), f2_inv(f2_dbl(a.y)));
  } else {
    lam = f2_mul(f2_sub(b.y, a.y), f2_inv(f2_sub(b.x, a.x)));
  }
  g2_affine r;
  r.x = f2_sub(f2_sub(f2_sqr(lam), a.x), b.x);
  r.y = f2_sub(f2_mul(lam, f2_sub(a.x, r.x)), a.y);
  r.inf = false;
  return r;
}
// Jacobian over Fp2 for the subgroup check (no inversions)
struct g2_jac {
  fp2 x, y, z;
};
inline g2_jac g2j_double(const g2_jac& p) {
  if (f2_is_zero(p.z)) return p;
  fp2 a = f2_sqr(p.x), b = f2_sqr(p.y), c = f2_sqr(b);
  fp2 d = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, b)), a), c));
  fp2 e = f2_add(f2_dbl(a), a), f = f2_sqr(e);
  g2_jac r;
  r.x = f2_sub(f, f2_dbl(d));
  r.z = f2_dbl(f2_mul(p.y, p.z));
  fp2 c8 = f2_dbl(f2_dbl(f2_dbl(c)));
  r.y = f2_sub(f2_mul(e, f2_sub(d, r.x)), c8);
  return r;
}
inline g2_jac g2j_add_affine(const g2_jac& p, const g2_affine& q) {  // q finite
  if (f2_is_zero(p.z)) return g2_jac{q.x, q.y, f2_one()};
  fp2 z1z1 = f2_sqr(p.z), u2 = f2_mul(q.x, z1z1), s2 = f2_mul(f2_mul(q.y, p.z), z1z1);
  if (f2_eq(u2, p.x)) {
    if (f2_eq(s2, p.y)) return g2j_double(p);
    return g2_jac{f2_one(), f2_one(), f2_zero()};
  }
  fp2 h = f2_sub(u2, p.x), hh = f2_sqr(h), i = f2_dbl(f2_dbl(hh)), j = f2_mul(h, i);
  fp2 rr = f2_dbl(f2_sub(s2, p.y)), v = f2_mul(p.x, i);
  g2_jac r;
  r.x = f2_sub(f2_sub(f2_sqr(rr), j), f2_dbl(v));
  r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_dbl(f2_mul(p.y, j)));
  r.z = f2_sub(f2_sub(f2_sqr(f2_add(p.z, h)), z1z1), hh);
  return r;
}
inline bool g2_in_subgroup(const g2_affine& p) {  // [r]Q == O  (blst_p2_affine_in_g2)
  if (p.inf) return true;
  const uint32_t rr[8] = KZG_FR_MOD_PLAIN;
  g2_jac acc{f2_one(), f2_one(), f2_zero()};
  for (int i = 254; i >= 0; i--) {
    acc = g2j_double(acc);
    if ((rr[i >> 5] >> (i & 31)) & 1u) acc = g2j_add_affine(acc, p);
  }
  return f2_is_zero(acc.z);
}

// square root in Fp2 (p = 3 mod 4), Adj & Rodriguez-Henriquez alg. 9; false if none
inline bool f2_sqrt(fp2& out, const fp2& a) {
  if (f2_is_zero(a)) {
    out = a;
    return true;
  }
  fp_t pm = modulus<FpParams>();
  // e1 = (p-3)/4, e2 = (p-1)/2
  uint32_t e1[12], e2[12];
  {
    fp_t t = pm;
    t.v[0] -= 3;
    for (int i = 0; i < 12; i++) e1[i] = (t.v[i] >> 2) | (i < 11 ? (t.v[i + 1] << 30) : 0);
    fp_t s = pm;
    s.v[0] -= 1;
    for (int i = 0; i < 12; i++) e2[i] = (s.v[i] >> 1) | (i < 11 ? (s.v[i + 1] << 31) : 0);
  }
  fp2 a1 = f2_pow(a, e1, 12);
  fp2 alpha = f2_mul(f2_sqr(a1), a);
  fp2 x0 = f2_mul(a1, a);
  fp2 minus_one = f2_neg(f2_one());
  fp2 cand;
  if (f2_eq(alpha, minus_one)) {
    cand.c0 = x0.c1;
    fp_neg(cand.c0, cand.c0);
    cand.c1 = x0.c0;  // u * x0
  } else {
    fp2 b = f2_pow(f2_add(f2_one(), alpha), e2, 12);
    cand = f2_mul(b, x0);
  }
  if (!f2_eq(f2_sqr(cand), a)) return false;
  out = cand;
  return true;
}

// blst_p2_uncompress + blst_p2_affine_in_g2 (P2::decompress, src/bls.rs:505-531 via :554-570)
inline int32_t g2_decompress(g2_affine& out, const uint8_t* in96) {
  uint8_t b0 = in96[0];
  if (!(b0 & 0x80)) return KZG_ERR_EC_INVALID_ENCODING;
  if (b0 & 0x40) {
    uint32_t o = b0 & 0x3F;
    for (int i = 1; i < 96; i++) o |= in96[i];
    if (o) return KZG_ERR_EC_INVALID_ENCODING;
    out = g2_affine{f2_zero(), f2_zero(), true};
    return KZG_OK;
  }
  fp_t x1p, x0p;
  fp_from_be_bytes_plain(x1p, in96);
  x1p.v[11] &= 0x1FFFFFFFu;
  fp_from_be_bytes_plain(x0p, in96 + 48);
  if (bn_geq(x1p, modulus<FpParams>()) || bn_geq(x0p, modulus<FpParams>())) return KZG_ERR_EC_INVALID_ENCODING;
  fp2 x;
  to_mont<FpParams>(x.c0, x0p);
  to_mont<FpParams>(x.c1, x1p);
  fp2 rhs = f2_add(f2_mul(f2_sqr(x), x), g2_b());
  fp2 y;
  if (!f2_sqrt(y, rhs)) return KZG_ERR_EC_NOT_ON_CURVE;
  // sign: y lexicographically larger, compared on c1 then c0
  fp_t y1p, y0p;
  from_mont<FpParams>(y1p, y.c1);
  from_mont<FpParams>(y0p, y.c0);
  bool larger = bn_is_zero(y1p) ? fp_is_lex_larger_plain(y0p) : fp_is_lex_larger_plain(y1p);
  if (((b0 & 0x20) != 0) != larger) y = f2_neg(y);
  out = g2_affine{x, y, false};
  if (!g2_in_subgroup(out)) return KZG_ERR_EC_NOT_IN_GROUP;
  return KZG_OK;
}

// ---- precomputed Miller lines for a fixed G2 point ------------------------------
struct miller_lines {
  // step k: line  y_P*w^3 - lambda*x_P*w^2 + mu   with mu = lambda*x_T - y_T
  std::vector<fp2> lambda, mu;
  bool q_is_inf = false;
};
static constexpr uint64_t BLS_X_ABS = 0xd201000000010000ull;

inline miller_lines precompute_lines(const g2_affine& q) {
  miller_lines L;
  if (q.inf) {
    L.q_is_inf = true;
    return L;
  }
  g2_affine t = q;
  for (int bit = 62; bit >= 0; bit--) {
    fp2 x2 = f2_sqr(t.x);
    fp2 lam = f2_mul(f2_add(f2_dbl(x2), x2), f2_inv(f2_dbl(t.y)));
    L.lambda.push_back(lam);
    L.mu.push_back(f2_sub(f2_mul(lam, t.x), t.y));
    t = g2_add(t, t);
    if ((BLS_X_ABS >> bit) & 1) {
      fp2 lam2 = f2_mul(f2_sub(q.y, t.y), f2_inv(f2_sub(q.x, t.x)));
      L.lambda.push_back(lam2);
      L.mu.push_back(f2_sub(f2_mul(lam2, t.x), t.y));
      t = g2_add(t, q);
    }
  }
  return L;
}

struct g1_host_affine {
  fp_t x, y;  // Montgomery
  bool inf;
};

// f = prod_j f_{|z|,Q_j}(P_j) over the given (P_j, lines_j) pairs, squarings shared.
inline fp12 multi_miller(const g1_host_affine* ps, const miller_lines* const* ls, int npairs) {
  fp12 f = f12_one();
  std::vector<size_t> idx(npairs, 0);
  for (int bit = 62; bit >= 0; bit--) {
    f = f12_sqr(f);
    for (int j = 0; j < npairs; j++) {
      if (ps[j].inf || ls[j]->q_is_inf) continue;
      size_t k = idx[j]++;
      fp2 l2 = f2_neg(f2_mul_fp(ls[j]->lambda[k], ps[j].x));
      f = f12_mul_by_line(f, ls[j]->mu[k], l2, ps[j].y);
    }
    if ((BLS_X_ABS >> bit) & 1) {
      for (int j = 0; j < npairs; j++) {
        if (ps[j].inf || ls[j]->q_is_inf) continue;
        size_t k = idx[j]++;
        fp2 l2 = f2_neg(f2_mul_fp(ls[j]->lambda[k], ps[j].x));
        f = f12_mul_by_line(f, ls[j]->mu[k], l2, ps[j].y);
      }
    }
  }
  return f;  // sign of z ignored: f^-1 == 1 iff f == 1 after the final exponentiation
}

// ---- final exponentiation ------------------------------------------------------
inline fp12 f12_pow_x(const fp12& a) {  // a^|z|, a in the cyclotomic subgroup
  fp12 r = a;
  for (int bit = 62; bit >= 0; bit--) {
    r = f12_cyclotomic_sqr(r);
    if ((BLS_X_ABS >> bit) & 1) r = f12_mul(r, a);
  }
  return r;
}

// f^((p^12-1)/r * 3) == 1 ?   (3 does not divide r, so this is equivalent to the
// reference's blst_final_exp + blst_fp12_is_one.)  Easy part (p^6-1)(p^2+1),
// hard part by the decomposition
//   3*(p^4-p^2+1)/r = (z-1)^2 * (z+p) * (z^2+p^2-1) + 3 ,  z = -|z|.
inline bool final_exp_is_one(const fp12& f_in, const frob_consts& fc) {
  fp12 f = f12_mul(f12_conj(f_in), f12_inv(f_in));           // f^(p^6-1): now unitary, inverse = conjugate
  f = f12_mul(f12_frobenius(f12_frobenius(f, fc), fc), f);   // ^(p^2+1)
  // a = f^(z-1)   : f^z = conj(f^|z|)
  fp12 fz = f12_conj(f12_pow_x(f));
  fp12 a = f12_mul(fz, f12_conj(f));                          // f^(z-1)
  fp12 az = f12_conj(f12_pow_x(a));
  a = f12_mul(az, f12_conj(a));                               // f^((z-1)^2)
  // b = a^(z+p)
  fp12 b = f12_mul(f12_conj(f12_pow_x(a)), f12_frobenius(a, fc));
  // c = b^(z^2+p^2-1)
  fp12 bz = f12_conj(f12_pow_x(b));
  fp12 bzz = f12_conj(f12_pow_x(bz));
  fp12 c = f12_mul(f12_mul(bzz, f12_frobenius(f12_frobenius(b, fc), fc)), f12_conj(b));
  // result = c * f^3
  fp12 r = f12_mul(c, f12_mul(f12_sqr(f), f));
  return f12_is_one(r);
}

// generic definition, used by tests to validate the chain above
inline fp12 f12_pow_big(const fp12& a, const std::vector<uint32_t>& e) {
  fp12 r = f12_one();
  for (int i = (int)e.size() * 32 - 1; i >= 0; i--) {
    r = f12_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = f12_mul(r, a);
  }
  return r;
}

// ---- context used by the engine -----------------------------------------------
struct pairing_ctx {
  frob_consts fc;
  miller_lines lines_g2, lines_tau;  // Q = G2 generator, Q = [tau]_2
};

// e(-A, [tau]_2) * e(B, G2) == 1   (bls::verify_pairings, src/bls.rs:572-598, with
// (a1,a2) = (A,[tau]_2) and (b1,b2) = (B, G2) as called at src/kzg/setup.rs:157-160)
inline bool verify_pairings_fixed(const pairing_ctx& pc, const g1_host_affine& a, const g1_host_affine& b) {
  g1_host_affine na = a;
  if (!na.inf) fp_neg(na.y, na.y);
  g1_host_affine ps[2] = {na, b};
  const miller_lines* ls[2] = {&pc.lines_tau, &pc.lines_g2};
  fp12 f = multi_miller(ps, ls, 2);
  return final_exp_is_one(f, pc.fc);
}

}  // namespace host
}  // namespace kzg
