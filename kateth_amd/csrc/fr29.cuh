// Fr of BLS12-381 in the carry-free radix-2^29 representation (rdx_mont.cuh): 9 limbs, Montgomery radix 2^261.
// Used by the barycentric evaluation of batch verification (k_eval_frac), where an Fr product costs 81 + 81
// v_mad_u64_u32 instead of 2 x 128 mad/addc instructions, two products share a reduction, and the additions
// carry nothing.  Replaces blst_fr_mul / blst_fr_add / blst_fr_sub behind Polynomial::evaluate (src/kzg/poly.rs:10-33).
#pragma once
#include "rdx_mont.cuh"

namespace kzg {

constexpr int F29_N = 9;
constexpr int F29_W = 29;
constexpr uint32_t F29_MASK = (1u << F29_W) - 1u;

struct Fr29P {
  static constexpr int N = F29_N;
  static constexpr int W = F29_W;
  static constexpr uint32_t INV = KZG_FR29_INV;  // = 2^29 - 1: r = 1 (mod 2^32)
  KZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t t[N] = KZG_FR29_MOD;
    return t[i];
  }
};
using fr29 = rdx_t<Fr29P>;

#define KZG_F29_TABLE(fn, MACRO)        \
  KZG_HD constexpr uint32_t fn(int i) { \
    constexpr uint32_t t[F29_N] = MACRO; \
    return t[i];                        \
  }
KZG_F29_TABLE(f29_one_limb, KZG_FR29_ONE)
KZG_F29_TABLE(f29_r2_limb, KZG_FR29_R2)
KZG_F29_TABLE(f29_inv4096_limb, KZG_FR29_INV4096)
KZG_F29_TABLE(f29_2r_t1, KZG_FR29_2R_T1)
#undef KZG_F29_TABLE

KZG_HD void f29_from_bn(fr29& r, const fr_t& a) { rdx_from_bn<Fr29P, 8>(r, a); }
KZG_HD void f29_to_bn(fr_t& r, const fr29& a) { rdx_to_bn<Fr29P, 8>(r, a); }

// r = (a*b [+ c*d]) / 2^261 mod r, N-form (limbs < 2^29, value < 2r).  Requires
// 9*(La*Lb [+ Lc*Ld]) + 9*2^58 < 2^64 for the limb bounds and (Va*Vb [+ Vc*Vd]) < 2^6 for the value bounds in units of r.
KZG_HD void f29_mul(fr29& r, const fr29& a, const fr29& b) { rdx_mul_core<Fr29P, false, false>(r, a, b, a, b); }
KZG_HD void f29_sqr(fr29& r, const fr29& a) { rdx_mul_core<Fr29P, true, false>(r, a, a, a, a); }
KZG_HD void f29_mul2(fr29& r, const fr29& a, const fr29& b, const fr29& c, const fr29& d) { rdx_mul_core<Fr29P, false, true>(r, a, b, c, d); }

KZG_HD void f29_add(fr29& r, const fr29& a, const fr29& b) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F29_N; i++) {
    RDX_ADDCHK(a.l[i], b.l[i]);
    r.l[i] = a.l[i] + b.l[i];
  }
}
// r = a + 2r - b limb-wise; b canonical (strictly normalised, < r)
KZG_HD void f29_sub_2r(fr29& r, const fr29& a, const fr29& b) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F29_N; i++) {
    RDX_SUBCHK(a.l[i], f29_2r_t1(i), b.l[i]);
    r.l[i] = a.l[i] + f29_2r_t1(i) - b.l[i];
  }
}
KZG_HD fr29 f29_const_one() {
  fr29 r;
  KZG_UNROLL_FULL
  for (int i = 0; i < F29_N; i++) r.l[i] = f29_one_limb(i);
  return r;
}
// plain (canonical) -> Montgomery N-form
KZG_HD void f29_to_mont(fr29& r, const fr29& a) {
  fr29 k;
  KZG_UNROLL_FULL
  for (int i = 0; i < F29_N; i++) k.l[i] = f29_r2_limb(i);
  f29_mul(r, a, k);
}
// necessary condition for a == 0 (mod r) when 0 <= a < 64 r: a = k r, and the low 29 bits are exact whatever the
// carries; r = 1 (mod 2^29), so k = a.l[0] mod 2^29.  False positives: 2^-23 of all inputs.
KZG_HD bool f29_maybe_zero(const fr29& a) { return (a.l[0] & F29_MASK) < 64u; }
// exact: one product by ONE brings a into N-form, where 0 (mod r) is exactly {0, r}
KZG_HD bool f29_is_zero_exact(const fr29& a) {
  fr29 t;
  f29_mul(t, a, f29_const_one());
  uint32_t z = 0, e = 0;
  KZG_UNROLL_FULL
  for (int i = 0; i < F29_N; i++) {
    z |= t.l[i];
    e |= t.l[i] ^ Fr29P::mod(i);
  }
  return z == 0 || e == 0;
}
// N-form -> canonical 8 x 32-bit limbs
KZG_HD void f29_to_canonical_bn(fr_t& r, const fr29& a) {
  f29_to_bn(r, a);  // < 2r < 2^256
  canonicalize<FrParams>(r);
}

// k_eval_frac's per-HEX table entry (16 elements 16h .. 16h+15 at the roots +-w rho_k, rho = 1, i, c, ic, s, is, cs, ics with i, c, s
// the primitive 4th, 8th and 16th roots of unity): twenty slots of 9 limbs, each padded to 12 dwords (three 16-byte loads,
// fetched where it is used):
//   0..7   (w rho_k) R                       8..11  w^2 R, i w^2 R, c w^2 R, i c w^2 R
//   12..15 w^4 R, c w^4 R, i w^4 R, c i w^4 R   16, 17 w^8 R, s w^8 R        18  w^16 R        19  w R^2
constexpr int EVAL_TAB_SLOT = 12;
constexpr int EVAL_TAB_SLOTS = 20;
constexpr int EVAL_TAB_DWORDS = EVAL_TAB_SLOTS * EVAL_TAB_SLOT;
constexpr int EVAL_TAB_HEXES = 256;

}  // namespace kzg
