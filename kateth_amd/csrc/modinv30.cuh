// Modular inversion by the Bernstein-Yang "safegcd" divsteps (eprint 2019/266), 30 divsteps per batch on signed
// 30-bit limbs, branch-free inside a batch (every lane of a wave executes the same instructions).
//
// Replaces Fermat inversion (a^(m-2): ~380 squarings + ~100 products, ~200 k VALU instructions even in the radix-2^28
// field) where an inversion sits on a latency-critical path: blst_p1_compress behind P1::compress (src/bls.rs:491-503),
// blst_fr_eucl_inverse (src/bls.rs:167-175) in the quotient kernel's root inverse, the table normalisation and the host's
// XYZZ -> affine conversions.  A 381-bit inversion is ~25 batches of ~600 instructions.
//
// Works on plain residues: for a Montgomery operand a*R the caller multiplies the result (a*R)^-1 by R^3
// (one Montgomery product) to get a^-1 * R.  0 maps to 0, as blst's inverse does.
#pragma once
#include "field.cuh"

namespace kzg {

template <int NL>
struct s30 {
  int32_t v[NL];  // value = sum v[i] * 2^(30 i); limbs 0..NL-2 in [0, 2^30), top limb signed
};

struct FpInv30 {
  static constexpr int NB = 12, NL = 13, MAX_BATCHES = 45;  // typical 26; the paper's bound (49*381 + 80) / 17 = 1,102 divsteps is 37 batches
  static constexpr uint32_t MODINV30 = KZG_FP_MODINV30;
  KZG_HD static constexpr int32_t mod(int i) {
    constexpr int32_t t[NL] = KZG_FP_MOD30;
    return t[i];
  }
};
struct FrInv30 {
  static constexpr int NB = 8, NL = 9, MAX_BATCHES = 32;  // typical 18; the paper's bound (49*255 + 80) / 17 = 739 divsteps is 25 batches
  static constexpr uint32_t MODINV30 = KZG_FR_MODINV30;
  KZG_HD static constexpr int32_t mod(int i) {
    constexpr int32_t t[NL] = KZG_FR_MOD30;
    return t[i];
  }
};

struct divsteps_matrix {
  int32_t u, v, q, r;
};

// 30 divsteps on the low words (half-delta variant: zeta = -(delta + 1/2), starts at -1).  On return
//   2^30 * (f', g') = [[u, v], [q, r]] * (f, g).
KZG_HD int32_t modinv30_divsteps(int32_t zeta, uint32_t f, uint32_t g, divsteps_matrix& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  KZG_UNROLL_FULL
  for (int i = 0; i < 30; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones iff zeta < 0
    const uint32_t c2 = 0u - (g & 1u);     // all ones iff g is odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // (f, u, v) negated iff zeta < 0
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;  // swap case: zeta < 0 and g odd
    zeta = (zeta ^ (int32_t)c1) - 1;
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return zeta;
}

// (f, g) <- t * (f, g) / 2^30   (exact)
template <int NL>
KZG_HD void modinv30_update_fg(s30<NL>& f, s30<NL>& g, const divsteps_matrix& t) {
  constexpr int32_t M30 = (int32_t)((1u << 30) - 1u);
  int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0];
  int64_t cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
  cf >>= 30;  // the low 30 bits are zero by construction
  cg >>= 30;
  KZG_UNROLL_FULL
  for (int i = 1; i < NL; i++) {
    const int32_t fi = f.v[i], gi = g.v[i];
    cf += (int64_t)t.u * fi + (int64_t)t.v * gi;
    cg += (int64_t)t.q * fi + (int64_t)t.r * gi;
    f.v[i - 1] = (int32_t)cf & M30;
    g.v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[NL - 1] = (int32_t)cf;
  g.v[NL - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod m, both kept in (-2m, m)
template <class M>
KZG_HD void modinv30_update_de(s30<M::NL>& d, s30<M::NL>& e, const divsteps_matrix& t) {
  constexpr int NL = M::NL;
  constexpr int32_t M30 = (int32_t)((1u << 30) - 1u);
  const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;  // sign masks
  int32_t md = (t.u & sd) + (t.v & se);  // multiples of m that bring negative d, e back towards the range
  int32_t me = (t.q & sd) + (t.r & se);
  int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0];
  int64_t ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
  // adjust the multiples so that the low 30 bits cancel
  md -= (int32_t)((M::MODINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((M::MODINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)M::mod(0) * md;
  ce += (int64_t)M::mod(0) * me;
  cd >>= 30;
  ce >>= 30;
  KZG_UNROLL_FULL
  for (int i = 1; i < NL; i++) {
    const int32_t di = d.v[i], ei = e.v[i];
    cd += (int64_t)t.u * di + (int64_t)t.v * ei + (int64_t)M::mod(i) * md;
    ce += (int64_t)t.q * di + (int64_t)t.r * ei + (int64_t)M::mod(i) * me;
    d.v[i - 1] = (int32_t)cd & M30;
    e.v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[NL - 1] = (int32_t)cd;
  e.v[NL - 1] = (int32_t)ce;
}

// r in (-2m, m), negated if sign < 0, brought to [0, m)
template <class M>
KZG_HD void modinv30_normalize(s30<M::NL>& r, int32_t sign) {
  constexpr int NL = M::NL;
  constexpr int32_t M30 = (int32_t)((1u << 30) - 1u);
  int32_t cond_add = r.v[NL - 1] >> 31;
  const int32_t cond_neg = sign >> 31;
  KZG_UNROLL_FULL
  for (int i = 0; i < NL; i++) {
    int32_t x = r.v[i] + (M::mod(i) & cond_add);
    r.v[i] = (x ^ cond_neg) - cond_neg;
  }
  KZG_UNROLL_FULL
  for (int i = 0; i < NL - 1; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
  cond_add = r.v[NL - 1] >> 31;  // now in (-m, m)
  KZG_UNROLL_FULL
  for (int i = 0; i < NL; i++) r.v[i] += M::mod(i) & cond_add;
  KZG_UNROLL_FULL
  for (int i = 0; i < NL - 1; i++) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= M30;
  }
}

// r = a^-1 mod m for a canonical plain residue a (0 -> 0).  Returns false if g did not reach 0 within MAX_BATCHES (never
// observed; the callers then fall back to the Fermat power).
template <class M>
KZG_HD bool modinv30_inl(bn<M::NB>& r, const bn<M::NB>& a) {
  constexpr int NL = M::NL, NB = M::NB;
  constexpr uint32_t M30 = (1u << 30) - 1u;
  s30<NL> f, g, d, e;
  KZG_UNROLL_FULL
  for (int i = 0; i < NL; i++) {
    const int bit = 30 * i, w = bit >> 5, s = bit & 31;
    uint32_t x = (w < NB) ? (a.v[w < NB ? w : 0] >> s) : 0u;
    if (s > 2 && w + 1 < NB) x |= a.v[w + 1 < NB ? w + 1 : 0] << (32 - s);
    g.v[i] = (int32_t)(x & M30);
    f.v[i] = M::mod(i);
    d.v[i] = 0;
    e.v[i] = (i == 0) ? 1 : 0;
  }
  int32_t zeta = -1;
  int32_t nz_last = 1;
#pragma unroll 1
  for (int batch = 0; batch < M::MAX_BATCHES; batch++) {
    divsteps_matrix t;
    zeta = modinv30_divsteps(zeta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
    modinv30_update_de<M>(d, e, t);
    modinv30_update_fg<NL>(f, g, t);
    int32_t nz = 0;
    KZG_UNROLL_FULL
    for (int i = 0; i < NL; i++) nz |= g.v[i];
    nz_last = nz;
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__any(nz != 0)) break;  // wave-uniform exit: extra divsteps with g == 0 change nothing
#else
    if (nz == 0) break;
#endif
  }
  modinv30_normalize<M>(d, f.v[NL - 1]);  // f = +-1 (or m for a == 0, where d == 0)
  KZG_UNROLL_FULL
  for (int w = 0; w < NB; w++) {
    const int bit = 32 * w, i = bit / 30, s = bit % 30;
    uint64_t x = (uint64_t)(uint32_t)d.v[i] >> s;
    if (i + 1 < NL) x |= (uint64_t)(uint32_t)d.v[i + 1 < NL ? i + 1 : 0] << (30 - s);
    if (i + 2 < NL && 60 - s < 32) x |= (uint64_t)(uint32_t)d.v[i + 2 < NL ? i + 2 : 0] << (60 - s);
    r.v[w] = (uint32_t)x;
  }
  return nz_last == 0;
}

// out of line: what every caller but the single-item encoder uses (its operands pass through memory; _inl keeps them in registers)
template <class M>
KZG_HD_NOINLINE bool modinv30(bn<M::NB>& r, const bn<M::NB>& a) {
  return modinv30_inl<M>(r, a);
}

// Montgomery-domain inverses (radix 2^384 / 2^256): (a R)^-1 * R^3 / R = a^-1 R.  0 -> 0.
KZG_HD void fp_inv(fp_t& r, const fp_t& a) {
  fp_t t, k;
  if (!modinv30<FpInv30>(t, a)) {
    fp_inv_fermat(r, a);
    return;
  }
  constexpr uint32_t r3[12] = KZG_FP_R3;
  KZG_UNROLL_FULL
  for (int i = 0; i < 12; i++) k.v[i] = r3[i];
  fp_mul(r, t, k);
}
// the same with every callee inline (operands never pass through memory): for kernels on the latency chain of single-item calls
KZG_HD void fr_inv_inl(fr_t& r, const fr_t& a) {
  fr_t t, k;
  if (!modinv30_inl<FrInv30>(t, a)) {  // never taken: MAX_BATCHES exceeds the proven bound
    mont_pow_const_inl<FrParams>(r, a, FrInvExp());
    return;
  }
  constexpr uint32_t r3[8] = KZG_FR_R3;
  KZG_UNROLL_FULL
  for (int i = 0; i < 8; i++) k.v[i] = r3[i];
  fr_mul(r, t, k);
}
KZG_HD void fr_inv(fr_t& r, const fr_t& a) {
  fr_t t, k;
  if (!modinv30<FrInv30>(t, a)) {
    fr_inv_fermat(r, a);
    return;
  }
  constexpr uint32_t r3[8] = KZG_FR_R3;
  KZG_UNROLL_FULL
  for (int i = 0; i < 8; i++) k.v[i] = r3[i];
  fr_mul(r, t, k);
}

}  // namespace kzg
