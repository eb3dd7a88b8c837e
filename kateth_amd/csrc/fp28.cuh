// Fp of BLS12-381 in a carry-free radix-2^28 representation, for the fixed-base MSM hot loop only.
//
// Why a second representation: on gfx950 the only wide integer multiply is v_mad_u64_u32
// (32x32 + 64 -> 64, full rate).  With saturated 32-bit limbs (field.cuh) the 64-bit column
// accumulator overflows after one product, so every product needs a second instruction
// (v_addc_co_u32 into a third word): 2 x 288 VALU instructions per Montgomery product, plus the
// VCC wait states hipcc adds around carry chains.  With 14 limbs of 28 bits a column of
// 14 + 14 products (each < 2^58 or so) fits one 64-bit accumulator: 392 v_mad_u64_u32 and NO carry
// instruction, squarings do 105 instead of 196 products, and additions/subtractions are 14
// independent 32-bit operations with no carry chain at all (limbs are allowed to grow, bounds are
// tracked statically in the comments below and re-checked at run time in the CPU build by
// KZG_FP28_CHECK, tests/test_hostmath.py).
//
// Value = sum l[i] * 2^(28 i).  Montgomery radix 2^392.  "N-form" = output of a Montgomery product:
// limbs 0..12 < 2^28, value < 2p.  Replaces blst_fp arithmetic behind blst_p1s_mult_pippenger
// (src/bls.rs:416-437) together with field.cuh.
#pragma once
#include "g1.cuh"
#include "rdx_mont.cuh"

namespace kzg {

constexpr int F28_N = 14;
constexpr int F28_W = 28;
constexpr uint32_t F28_MASK = (1u << F28_W) - 1u;

struct Fp28P {
  static constexpr int N = F28_N;
  static constexpr int W = F28_W;
  static constexpr uint32_t INV = KZG_FP28_INV;
  KZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t t[N] = KZG_FP28_MOD;
    return t[i];
  }
};
using fp28 = rdx_t<Fp28P>;

#define KZG_F28_TABLE(fn, MACRO)                 \
  KZG_HD constexpr uint32_t fn(int i) {          \
    constexpr uint32_t t[F28_N] = MACRO;         \
    return t[i];                                 \
  }
KZG_F28_TABLE(f28_p, KZG_FP28_MOD)
KZG_F28_TABLE(f28_one_limb, KZG_FP28_ONE)
KZG_F28_TABLE(f28_r384_limb, KZG_FP28_R384)
KZG_F28_TABLE(f28_2p_t1, KZG_FP28_2P_T1)
KZG_F28_TABLE(f28_4p_t1, KZG_FP28_4P_T1)
KZG_F28_TABLE(f28_16p_t1, KZG_FP28_16P_T1)
KZG_F28_TABLE(f28_8p_t3, KZG_FP28_8P_T3)
#undef KZG_F28_TABLE

// the bound-check hooks of rdx_mont.cuh under their fp28 names
#define F28_SUBCHK RDX_SUBCHK
#define F28_ADDCHK RDX_ADDCHK

// ---- representation changes (rdx_mont.cuh) -----------------------------------------------------------
// 12 x 32-bit limbs (value < 2^384) -> 14 x 28-bit limbs, strictly normalised, and back
KZG_HD void f28_from_bn(fp28& r, const fp_t& a) { rdx_from_bn<Fp28P, 12>(r, a); }
KZG_HD void f28_to_bn(fp_t& r, const fp28& a) { rdx_to_bn<Fp28P, 12>(r, a); }

// one carry pass, all limbs at once (no serial chain): limbs 0..12 end up <= 2^28 - 1 + (max limb >> 28)
KZG_HD void f28_carry_pass(fp28& a) {
  uint32_t hi[F28_N];
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N - 1; i++) hi[i] = a.l[i] >> F28_W;
  KZG_UNROLL_FULL
  for (int i = F28_N - 1; i >= 1; i--) {
    const uint32_t lo = (i < F28_N - 1) ? (a.l[i] & F28_MASK) : a.l[i];
    F28_ADDCHK(lo, hi[i - 1]);
    a.l[i] = lo + hi[i - 1];
  }
  a.l[0] &= F28_MASK;
}
// full (serial) carry propagation: limbs 0..12 < 2^28 exactly
KZG_HD void f28_normalize(fp28& a) {
  uint32_t c = 0;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N - 1; i++) {
    F28_ADDCHK(a.l[i], c);
    const uint32_t t = a.l[i] + c;
    a.l[i] = t & F28_MASK;
    c = t >> F28_W;
  }
  F28_ADDCHK(a.l[F28_N - 1], c);
  a.l[F28_N - 1] += c;
}

// ---- Montgomery products (rdx_mont.cuh) -----------------------------------------------------------
// r = (a*b [+ c*d]) / 2^392 mod p, N-form.  Requires 14*(La*Lb [+ Lc*Ld]) + 14*2^56 < 2^64 for the limb
// bounds L, and (Va*Vb [+ Vc*Vd]) < 2^11 for the value bounds V in units of p (so the result is < 2p).
KZG_HD void f28_mul(fp28& r, const fp28& a, const fp28& b) { rdx_mul_core<Fp28P, false, false>(r, a, b, a, b); }
KZG_HD void f28_sqr(fp28& r, const fp28& a) { rdx_mul_core<Fp28P, true, false>(r, a, a, a, a); }
// r = (a*b + c*d) / 2^392: two products, ONE reduction
KZG_HD void f28_mul2(fp28& r, const fp28& a, const fp28& b, const fp28& c, const fp28& d) { rdx_mul_core<Fp28P, false, true>(r, a, b, c, d); }

// ---- carry-free add / subtract ------------------------------------------------------------------
KZG_HD void f28_add(fp28& r, const fp28& a, const fp28& b) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_ADDCHK(a.l[i], b.l[i]);
    r.l[i] = a.l[i] + b.l[i];
  }
}
// r = a + KP - b limb-wise, KP a redundant multiple of p (see tools/gen_consts.py) that dominates b's limbs
#define KZG_F28_SUB(name, KP)                                             \
  KZG_HD void name(fp28& r, const fp28& a, const fp28& b) {               \
    KZG_UNROLL_FULL                                                       \
    for (int i = 0; i < F28_N; i++) {                                     \
      F28_SUBCHK(a.l[i], KP(i), b.l[i]);                                  \
      r.l[i] = a.l[i] + KP(i) - b.l[i];                                   \
    }                                                                     \
  }
KZG_F28_SUB(f28_sub_4p, f28_4p_t1)    // b: N-form
KZG_F28_SUB(f28_sub_16p, f28_16p_t1)  // b: limbs <= 2^28 + 16, value < 10p
KZG_F28_SUB(f28_sub_8p3, f28_8p_t3)   // b: limbs <= 3 (2^28 - 1), value < 6p
#undef KZG_F28_SUB
// r = KP - b
KZG_HD void f28_neg_2p(fp28& r, const fp28& b) {  // b canonical (< p, strictly normalised)
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(0u, f28_2p_t1(i), b.l[i]);
    r.l[i] = f28_2p_t1(i) - b.l[i];
  }
}
KZG_HD void f28_neg_4p(fp28& r, const fp28& b) {  // b N-form
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(0u, f28_4p_t1(i), b.l[i]);
    r.l[i] = f28_4p_t1(i) - b.l[i];
  }
}

KZG_HD fp28 f28_one() {
  fp28 r;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) r.l[i] = f28_one_limb(i);
  return r;
}

// Cheap necessary condition for a == 0 (mod p) when 0 <= a < 2^11 p: a = k p with k < 2^11, and the low
// 28 bits of a are exact whatever the carries, so k = a.l[0] * p^-1 mod 2^28 must be < 2^11.
// False positives: 2^-17 of all inputs; the caller then runs f28_is_zero_exact.
KZG_HD bool f28_maybe_zero(const fp28& a) { return ((a.l[0] * (uint32_t)KZG_FP28_PINV) & F28_MASK) < 2048u; }
// exact: one Montgomery product by ONE brings a into N-form, where 0 (mod p) is exactly {0, p}
KZG_HD bool f28_is_zero_exact(const fp28& a) {
  fp28 t;
  f28_mul(t, a, f28_one());
  uint32_t z = 0, e = 0;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    z |= t.l[i];
    e |= t.l[i] ^ f28_p(i);
  }
  return z == 0 || e == 0;
}

// ---- 2^392-Montgomery <-> 2^384-Montgomery (field.cuh) ------------------------------------------------
// x*2^392 (any bounded fp28 value) -> canonical x*2^384 in 12 x 32 limbs
KZG_HD void f28_to_fp(fp_t& r, const fp28& a) {
  fp28 k, t;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) k.l[i] = f28_r384_limb(i);
  f28_mul(t, a, k);  // a * 2^384 / 2^392 = x * 2^384, N-form (< 2p < 2^382)
  f28_to_bn(r, t);
  canonicalize<FpParams>(r);
}
// canonical x*2^384 -> canonical x*2^392 as 12 x 32 limbs (the format of the test-only window tables and of the variable-base MSM's decoded points)
KZG_HD void fp_to_r392(fp_t& r, const fp_t& a) {
  fp_t k;
  constexpr uint32_t t[12] = KZG_FP_R392_PLAIN;
  KZG_UNROLL_FULL
  for (int i = 0; i < 12; i++) k.v[i] = t[i];
  fp_mul(r, a, k);
}

// ---- XYZZ accumulator in fp28 ---------------------------------------------------------------------
// Invariant between additions:  x: limbs <= 2^28 + 16, value < 10p;  y, zz, zzz: N-form.
struct g1_xyzz28 {
  fp28 x, y, zz, zzz;
  uint32_t inf;
};

KZG_HD void xyzz28_set_inf(g1_xyzz28& p) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) p.x.l[i] = p.y.l[i] = p.zz.l[i] = p.zzz.l[i] = 0;
  p.inf = 1;
}

// p = 2 * (x, y), (x, y) finite; x canonical, y = +-y limbs < 2^29 value <= 2p   (mdbl-2008-s-1, a = 0)
KZG_HD void xyzz28_mdbl(g1_xyzz28& p, const fp28& x, const fp28& y) {
  if (f28_is_zero_exact(y)) {  // order-2 point: not on this curve's subgroup, handled for completeness
    xyzz28_set_inf(p);
    return;
  }
  fp28 u, v, w, s, m, t;
  f28_add(u, y, y);  // limbs < 2^30, value <= 4p
  f28_sqr(v, u);
  f28_mul(w, u, v);
  f28_mul(s, x, v);
  f28_sqr(m, x);
  f28_add(t, m, m);
  f28_add(m, m, t);  // 3x^2: limbs < 3*2^28, value < 6p
  fp28 x3;
  f28_sqr(x3, m);
  f28_add(t, s, s);  // limbs < 2^29, value < 4p
  f28_sub_8p3(x3, x3, t);  // value < 10p, limbs < 5*2^28
  f28_carry_pass(x3);
  f28_sub_16p(t, s, x3);  // S - X3: limbs < 3*2^28, value < 18p
  fp28 ny;
  f28_neg_4p(ny, w);  // we need M*(S-X3) - W*y: second product enters with -W
  f28_mul2(p.y, m, t, ny, y);
  p.x = x3;
  p.zz = v;
  p.zzz = w;
  p.inf = 0;
}

// p += (x2, y2) for a FINITE accumulator p and the generic case; x2 canonical; y2 either canonical or
// 2p - canonical (limbs < 2^29, value <= 2p).  Returns false -- with p untouched -- when x2 * ZZ1 == X1 (mod p)
// may hold (P + P or P + (-P): about 2^-17 of all calls are false alarms); the caller then runs
// xyzz28_madd_complete.  The split keeps the operands of the rare doubling out of the hot path's live registers.
// madd-2008-s with Y3 = (R (Q - X3) + (4p - Y1) PPP) / 2^392 as ONE reduction: 7 products + 2 squarings +
// 1 double product, no carry chains anywhere.  Statement order keeps at most seven field elements live.
KZG_HD bool xyzz28_madd_fast(g1_xyzz28& p, const fp28& x2, const fp28& y2) {
  fp28 u2, r, pp, ppp;
  f28_mul(u2, x2, p.zz);      // U2: 14 * 2^56
  f28_mul(r, y2, p.zzz);      // S2: 14 * 2^57
  f28_sub_16p(u2, u2, p.x);   // P = U2 - X1: limbs < 3*2^28, value < 18p
  if (f28_maybe_zero(u2)) return false;
  f28_sub_4p(r, r, p.y);      // R = S2 - Y1: limbs < 3*2^28, value < 6p
  f28_sqr(pp, u2);            // PP: 14 * 9 * 2^56
  f28_mul(ppp, u2, pp);       // PPP
  f28_mul(p.zz, p.zz, pp);    // ZZ3
  f28_mul(p.zzz, p.zzz, ppp); // ZZZ3
  f28_mul(pp, p.x, pp);       // Q = X1 * PP (X1 and PP are dead from here)
  f28_sqr(p.x, r);            // R^2
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = ppp.l[i] + 2u * pp.l[i];  // PPP + 2Q: limbs <= 3 (2^28 - 1), value < 6p
    F28_SUBCHK(p.x.l[i], f28_8p_t3(i), s);
    p.x.l[i] = p.x.l[i] + f28_8p_t3(i) - s;      // X3 = R^2 - PPP - 2Q: limbs < 5*2^28, value < 10p
  }
  f28_carry_pass(p.x);        // limbs <= 2^28 + 3
  f28_sub_16p(pp, pp, p.x);   // Q - X3: limbs < 3*2^28, value < 18p
  f28_neg_4p(p.y, p.y);       // 4p - Y1: limbs < 2^29, value <= 4p
  f28_mul2(p.y, r, pp, p.y, ppp);  // Y3: 14 * (9 + 2) * 2^56 ; (6*18 + 4*2) p^2 / 2^392 < p/16
  return true;
}

// Complete addition (identity, P + P, P + (-P), and the generic case): the out-of-line companion of
// xyzz28_madd_fast.  Same operand conventions.
KZG_HD_NOINLINE void xyzz28_madd_complete(g1_xyzz28& p, const fp28& x2, const fp28& y2) {
  if (p.inf) {
    const fp28 one = f28_one();
    p.x = x2;
    f28_mul(p.y, y2, one);  // brings a negated y (limbs < 2^29) into N-form
    p.zz = one;
    p.zzz = one;
    p.inf = 0;
    return;
  }
  if (xyzz28_madd_fast(p, x2, y2)) return;
  fp28 u2, r;
  f28_mul(u2, x2, p.zz);
  f28_mul(r, y2, p.zzz);
  f28_sub_16p(u2, u2, p.x);
  f28_sub_4p(r, r, p.y);
  if (f28_is_zero_exact(u2)) {
    if (f28_is_zero_exact(r))
      xyzz28_mdbl(p, x2, y2);
    else
      xyzz28_set_inf(p);
    return;
  }
  // false alarm of the cheap test: the generic formulas apply
  fp28 pp, ppp;
  f28_sqr(pp, u2);
  f28_mul(ppp, u2, pp);
  f28_mul(p.zz, p.zz, pp);
  f28_mul(p.zzz, p.zzz, ppp);
  f28_mul(pp, p.x, pp);
  f28_sqr(p.x, r);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = ppp.l[i] + 2u * pp.l[i];
    F28_SUBCHK(p.x.l[i], f28_8p_t3(i), s);
    p.x.l[i] = p.x.l[i] + f28_8p_t3(i) - s;
  }
  f28_carry_pass(p.x);
  f28_sub_16p(pp, pp, p.x);
  f28_neg_4p(p.y, p.y);
  f28_mul2(p.y, r, pp, p.y, ppp);
}

// a == 0 (mod p) for 0 <= a < 2^11 p
KZG_HD bool f28_is_zero(const fp28& a) { return f28_maybe_zero(a) && f28_is_zero_exact(a); }

// p = 2 p   (dbl-2008-s-1, a = 0) on an accumulator that satisfies the invariant of g1_xyzz28
// (x: limbs <= 2^28 + 16, value < 10p; y, zz, zzz: N-form); the invariant holds again afterwards.
KZG_HD void xyzz28_dbl_inl(g1_xyzz28& p) {
  if (p.inf) return;
  if (f28_is_zero(p.y)) {  // a point of order two
    xyzz28_set_inf(p);
    return;
  }
  fp28 u, v, w, s, m, t;
  f28_add(u, p.y, p.y);     // limbs < 2^29, value < 4p
  f28_sqr(v, u);            // 14 * 2^58
  f28_mul(w, u, v);
  f28_mul(s, p.x, v);
  f28_sqr(m, p.x);
  f28_add(t, m, m);
  f28_add(m, m, t);         // 3 X^2: limbs < 3*2^28, value < 6p
  fp28 x3;
  f28_sqr(x3, m);           // 14 * 9 * 2^56
  f28_add(t, s, s);         // limbs < 2^29, value < 4p
  f28_sub_8p3(x3, x3, t);   // X3 = M^2 - 2S: limbs < 5*2^28, value < 10p
  f28_carry_pass(x3);
  f28_sub_16p(t, s, x3);    // S - X3: limbs < 3*2^28, value < 18p
  fp28 nw;
  f28_neg_4p(nw, w);        // limbs < 2^29, value <= 4p
  f28_mul2(p.y, m, t, nw, p.y);  // Y3 = M (S - X3) - W Y1: 14 * (9 + 2) * 2^56
  p.x = x3;
  f28_mul(p.zz, p.zz, v);
  f28_mul(p.zzz, p.zzz, w);
}
KZG_HD_NOINLINE void xyzz28_dbl(g1_xyzz28& p) { xyzz28_dbl_inl(p); }

// p += q, both XYZZ accumulators under the invariant; complete   (add-2008-s).  _inl: for the lane-sum trees, whose lone waves
// would otherwise pass both operands through scratch memory on every level.
// DBL_INL: the P + P case doubles inline too, so that the accumulator's address is never taken: no scratch at all on the latency
// chains of the single-item calls (lane-sum trees, encoder).  Kernels whose register budget is tight keep the out-of-line doubling
// (scratch is then allocated but touched only when two equal points meet).
template <bool DBL_INL = false>
KZG_HD void xyzz28_add_complete_inl(g1_xyzz28& p, const g1_xyzz28& q) {
  if (q.inf) return;
  if (p.inf) {
    p = q;
    return;
  }
  fp28 u1, u2, s1, s2, pp, ppp;
  f28_mul(u1, p.x, q.zz);
  f28_mul(u2, q.x, p.zz);
  f28_mul(s1, p.y, q.zzz);
  f28_mul(s2, q.y, p.zzz);
  f28_sub_4p(u2, u2, u1);  // P: limbs < 3*2^28, value < 6p
  f28_sub_4p(s2, s2, s1);  // R
  if (f28_is_zero(u2)) {
    if (f28_is_zero(s2)) {
      if constexpr (DBL_INL)
        xyzz28_dbl_inl(p);
      else
        xyzz28_dbl(p);
    } else {
      xyzz28_set_inf(p);
    }
    return;
  }
  f28_sqr(pp, u2);
  f28_mul(ppp, u2, pp);
  f28_mul(p.zz, p.zz, q.zz);
  f28_mul(p.zz, p.zz, pp);
  f28_mul(p.zzz, p.zzz, q.zzz);
  f28_mul(p.zzz, p.zzz, ppp);
  f28_mul(pp, u1, pp);     // Q = U1 PP
  f28_sqr(p.x, s2);        // R^2
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = ppp.l[i] + 2u * pp.l[i];
    F28_SUBCHK(p.x.l[i], f28_8p_t3(i), s);
    p.x.l[i] = p.x.l[i] + f28_8p_t3(i) - s;  // X3 = R^2 - PPP - 2Q
  }
  f28_carry_pass(p.x);
  f28_sub_16p(pp, pp, p.x);  // Q - X3
  fp28 ns1;
  f28_neg_4p(ns1, s1);
  f28_mul2(p.y, s2, pp, ns1, ppp);  // Y3 = R (Q - X3) - S1 PPP
}
KZG_HD_NOINLINE void xyzz28_add_complete(g1_xyzz28& p, const g1_xyzz28& q) { xyzz28_add_complete_inl<false>(p, q); }

// accumulator -> the 12 x 32-limb XYZZ format of g1.cuh (2^384 Montgomery, canonical)
KZG_HD void xyzz28_to_xyzz(g1_xyzz& r, const g1_xyzz28& p) {
  if (p.inf) {
    xyzz_set_inf(r);
    return;
  }
  f28_to_fp(r.x, p.x);
  f28_to_fp(r.y, p.y);
  f28_to_fp(r.zz, p.zz);
  f28_to_fp(r.zzz, p.zzz);
}
// the 12 x 32-limb XYZZ format (canonical 2^384-Montgomery) -> radix-2^28 accumulator (invariant of g1_xyzz28)
KZG_HD void xyzz28_from_xyzz(g1_xyzz28& r, const g1_xyzz& p) {
  if (xyzz_is_inf(p)) {
    xyzz28_set_inf(r);
    return;
  }
  fp28 k;
  {
    constexpr uint32_t t[F28_N] = KZG_FP28_R400;
    KZG_UNROLL_FULL
    for (int i = 0; i < F28_N; i++) k.l[i] = t[i];
  }
  // v * 2^384 read as an integer, times 2^400 / 2^392  ->  v * 2^392, N-form
  f28_from_bn(r.x, p.x);
  f28_mul(r.x, r.x, k);
  f28_from_bn(r.y, p.y);
  f28_mul(r.y, r.y, k);
  f28_from_bn(r.zz, p.zz);
  f28_mul(r.zz, r.zz, k);
  f28_from_bn(r.zzz, p.zzz);
  f28_mul(r.zzz, r.zzz, k);
  r.inf = 0;
}
// table entry (canonical x*2^392, y*2^392 as 12 x 32 limbs) -> operands of xyzz28_madd
KZG_HD void f28_load_entry(fp28& x, fp28& y, const fp_t& rx, const fp_t& ry, bool neg) {
  f28_from_bn(x, rx);
  fp28 t;
  f28_from_bn(t, ry);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) y.l[i] = neg ? (f28_2p_t1(i) - t.l[i]) : t.l[i];
}
}  // namespace kzg
