// Fp of BLS12-381 in a SIGNED radix-2^30 representation: 13 centred limbs, Montgomery radix 2^390 -- round 5's field of the
// fixed-base MSM hot loop (k_msm_comb28's successor), beside fp28.cuh (14 x 28-bit limbs, radix 2^392), which every other kernel
// keeps.
//
// Why: on gfx950 every VALU instruction of the adder's stream costs one ~4-cycle issue slot, so a
// Montgomery product costs its instruction count, and that is N^2 + N^2 v_mad for N limbs: 13 limbs are 169 + 169 instead of
// 196 + 196 (-14 %).  Thirteen limbs of an UNSIGNED radix 2^30 do not work: a column of 13 + 13 products of 30-bit limbs
// overflows the 64-bit accumulator (26 * 2^60).  With limbs CENTRED in [-2^29, 2^29] -- operands, the modulus' limbs and the
// quotient digits q_k alike -- a product is at most 2^58, a column of 12 full products + its reduction terms stays below
// 2^61.6 + 2^60.7 (sum |p_j| = 2^31.63), and there is room for ONE lazy operand (an unreduced sum or difference of two
// centred values, limbs <= 2^30): worst column 2^62.92 < 2^63 (tools/exp/fp30_model.py runs the extremes).  Signed values
// also make a subtraction what it is (no multiple of p to add), and a centred quotient keeps every product in (-0.53 p, 0.53 p)
// whatever came in, so values never grow.  What it costs: v_mad_i64_i32 instead of v_mad_u64_u32 (same issue slot), a rounding
// add per output column (the carry of a centred digit is round(A / 2^30), not floor), and a value that came out of an addition
// has limbs too wide for a squaring or for a second lazy position.  A carry pass (f30_carry) repairs that; the hot loop does
// not need one: its three differences are INJECTED into the product that feeds them (f30_mul_inj: the subtrahend's limbs join
// the columns the digits are cut from, one multiply-add per limb -- what the subtraction cost anyway -- and the difference
// comes out C-form).  DESIGN.md section 5.3 has the tally and the measurements.
//
// Value = sum l[i] * 2^(30 i).  "C-form": limbs 0..11 in [-2^29 - 2, 2^29 + 2], limb 12 small (|value| < 2^385);
// "L-form": one sum/difference of two C-forms (limbs 0..11 within +-(2^30 + 4)).  A product takes C x C, or L x C; a squaring
// and a double product (f30_mul2) take C-forms only.  Every product's output is C-form with |value| < 0.53 p (+ what was
// injected).
// The CPU test build (KZG_FP28_CHECK) forms every column in 128 bits as well and aborts on a violated bound.
#pragma once
#include "fp30_consts.cuh"
#include "g1.cuh"

namespace kzg {

constexpr int F30_N = 13;
constexpr int F30_W = 30;
constexpr int32_t F30_H = 1 << 29;
constexpr uint32_t F30_MASK = (1u << 30) - 1u;

struct fp30 {
  int32_t l[F30_N];
};

#define KZG_F30_TABLE(fn, MACRO)               \
  KZG_HD constexpr int32_t fn(int i) {         \
    constexpr int32_t t[F30_N] = MACRO;        \
    return t[i];                               \
  }
KZG_F30_TABLE(f30_p, KZG_FP30_MOD)
KZG_F30_TABLE(f30_one_limb, KZG_FP30_ONE)
KZG_F30_TABLE(f30_r384_limb, KZG_FP30_R384)
KZG_F30_TABLE(f30_r2_limb, KZG_FP30_R2)
#undef KZG_F30_TABLE

#if !defined(__HIP_DEVICE_COMPILE__) && defined(KZG_FP28_CHECK)
extern "C" void kzg_fp28_check_failed(const char* what);
struct f30_col {
  __int128 wide;
  int64_t v;
};
#define F30_COL_INIT(A) \
  f30_col A { 0, 0 }
#define F30_MAC(A, x, y)                                                           \
  do {                                                                             \
    (A).wide += (__int128)(int32_t)(x) * (int32_t)(y);                             \
    (A).v = (int64_t)((uint64_t)(A).v + (uint64_t)((int64_t)(int32_t)(x) * (int32_t)(y))); \
    if ((A).wide != (__int128)(A).v) kzg_fp28_check_failed("fp30 column overflow"); \
  } while (0)
#define F30_LO(A) ((uint32_t)(A).v)
#define F30_SHIFT_EXACT(A)                                                      \
  do {                                                                          \
    if ((A).v & 0x3fffffff) kzg_fp28_check_failed("fp30 inexact quotient column"); \
    (A).v >>= 30;                                                               \
    (A).wide = (A).v;                                                           \
  } while (0)
#define F30_SHIFT_ROUND(A)                        \
  do {                                            \
    (A).v = ((A).v + (int64_t)F30_H) >> 30;       \
    (A).wide = (A).v;                             \
  } while (0)
#define F30_SHIFT_FLOOR(A) \
  do {                     \
    (A).v >>= 30;          \
    (A).wide = (A).v;      \
  } while (0)
#define F30_TOP(A) ((int32_t)(A).v)
#define F30_LIMBCHK(x, bound)                                                             \
  do {                                                                                    \
    if ((int64_t)(x) > (int64_t)(bound) || (int64_t)(x) < -(int64_t)(bound)) kzg_fp28_check_failed("fp30 limb bound"); \
  } while (0)
#else
#define F30_COL_INIT(A) int64_t A = 0
#define F30_MAC(A, x, y) (A) += (int64_t)(int32_t)(x) * (int32_t)(y)
#define F30_LO(A) ((uint32_t)(A))
#define F30_SHIFT_EXACT(A) (A) >>= 30
#define F30_SHIFT_ROUND(A) (A) = ((A) + (int64_t)F30_H) >> 30
#define F30_SHIFT_FLOOR(A) (A) >>= 30
#define F30_TOP(A) ((int32_t)(A))
#define F30_LIMBCHK(x, bound) \
  do {                        \
  } while (0)
#endif

// the low 30 bits as a centred digit in [-2^29, 2^29)
KZG_HD int32_t f30_sbfe(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sbfe((int32_t)x, 0, 30);
#else
  return (int32_t)(x << 2) >> 2;
#endif
}

// ---- Montgomery products ------------------------------------------------------------------------------------------------
// r = (a*b [+ c*d]) / 2^390 mod p, C-form, |r| < 0.53 p.  Operand forms: see the header.  r may alias an operand.
// INJECTION (C0, C1 != 0): r = (a*b [+ c*d]) / 2^390 + C0 * inj0 + C1 * inj1, the integer sum, with their limbs added into the
// columns the result's digits are cut from -- so a difference like P = X2 ZZ1 / 2^390 - X1 comes out of the product ALREADY
// C-form (centred digits), for one multiply-add per limb where a separate subtraction costs one instruction per limb too and a
// carry pass four more.  inj0 / inj1: any limbs within +-2^31 (C- or L-form); |r| < 0.53 p + |C0 inj0| + |C1 inj1|.
// UFORM: the result's digits 0..11 are cut with FLOOR instead of round -- [0, 2^30), "U-form": the same integer, limbs within the
// L-form bound, so it may only ever meet a C-form in a product -- which saves the rounding add of every output column.
template <bool SQR, bool TWO, int C0 = 0, int C1 = 0, bool UFORM = false>
KZG_HD void f30_mul_core_c(fp30& r, const fp30& a, const fp30& b, const fp30& c, const fp30& d, const fp30& inj0, const fp30& inj1) {
  constexpr int N_ = F30_N;
  int32_t q[N_];
  int32_t a2[N_];
  if (SQR) {
    KZG_UNROLL_FULL
    for (int i = 0; i < N_; i++) {
      F30_LIMBCHK(a.l[i], (i < N_ - 1) ? F30_H + 2 : (1 << 25));
      a2[i] = a.l[i] * 2;
    }
  }
  F30_COL_INIT(A);
  KZG_UNROLL_FULL
  for (int k = 0; k < 2 * N_; k++) {
    const int i0 = (k < N_) ? 0 : k - N_ + 1;
    const int i1 = (k < N_) ? k : N_ - 1;
    if (SQR) {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) {
        const int j = k - i;
        if (i < j) F30_MAC(A, a2[i], a.l[j]);
        if (i == j) F30_MAC(A, a.l[i], a.l[i]);
      }
    } else {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) F30_MAC(A, a.l[i], b.l[k - i]);
    }
    if (TWO) {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) F30_MAC(A, c.l[i], d.l[k - i]);
    }
    if (k < N_) {
      KZG_UNROLL_FULL
      for (int i = 0; i < k; i++) F30_MAC(A, q[i], f30_p(k - i));
      q[k] = f30_sbfe(F30_LO(A) * (uint32_t)KZG_FP30_INV);
      F30_MAC(A, q[k], f30_p(0));
      F30_SHIFT_EXACT(A);
    } else {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) F30_MAC(A, q[i], f30_p(k - i));
      if (C0 != 0) F30_MAC(A, inj0.l[k - N_], C0);
      if (C1 != 0) F30_MAC(A, inj1.l[k - N_], C1);
      if (k < 2 * N_ - 1) {
        if (UFORM) {
          r.l[k - N_] = (int32_t)(F30_LO(A) & F30_MASK);
          F30_SHIFT_FLOOR(A);
        } else {
          r.l[k - N_] = f30_sbfe(F30_LO(A));  // r may alias an operand: limb k-13 of every operand was last read in column k-1
          F30_SHIFT_ROUND(A);                 // (A - digit) / 2^30
        }
      } else {
        r.l[N_ - 1] = F30_TOP(A);
      }
    }
  }
}

}  // namespace kzg
#include "mac30_asm.cuh"  // mad30_chain: v_mad_i64_i32 chains (generated, tools/gen_mac_asm.py)
namespace kzg {

#if defined(__HIP_DEVICE_COMPILE__)
// Device version: every column is issued as explicit v_mad_i64_i32 chains that START from the carry of the previous column
// (rdx_mont.cuh has the reasons).  Same arithmetic as f30_mul_core_c, which the CPU tests run.
template <bool SQR, bool TWO, int C0, int C1, bool UFORM, int K>
KZG_HD void f30_column(int64_t& A, int32_t* q, fp30& r, const fp30& a, const int32_t* a2, const fp30& b, const fp30& c, const fp30& d,
                       const fp30& inj0, const fp30& inj1) {
  constexpr int N_ = F30_N;
  constexpr int i0 = (K < N_) ? 0 : K - N_ + 1;
  constexpr int i1 = (K < N_) ? K : N_ - 1;
  constexpr int cnt = i1 - i0 + 1;
  if constexpr (SQR) {
    constexpr int last_pair = (K - 1) / 2;
    constexpr int npairs = (K >= 1 && last_pair >= i0) ? last_pair - i0 + 1 : 0;
    constexpr int diag = (K % 2 == 0) ? 1 : 0;
    int32_t xs[npairs + diag], ys[npairs + diag];
    KZG_UNROLL_FULL
    for (int t = 0; t < npairs; t++) {
      xs[t] = a2[i0 + t];
      ys[t] = a.l[K - i0 - t];
    }
    if constexpr (diag) {
      xs[npairs] = a.l[K / 2];
      ys[npairs] = a.l[K / 2];
    }
    mad30_chain<npairs + diag, false>::run(A, xs, ys);
  } else {
    int32_t xs[cnt], ys[cnt];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      xs[t] = a.l[i0 + t];
      ys[t] = b.l[K - i0 - t];
    }
    mad30_chain<cnt, false>::run(A, xs, ys);
  }
  if constexpr (TWO) {
    int32_t xs[cnt], ys[cnt];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      xs[t] = c.l[i0 + t];
      ys[t] = d.l[K - i0 - t];
    }
    mad30_chain<cnt, false>::run(A, xs, ys);
  }
  if constexpr (K < N_) {
    if constexpr (K > 0) {
      int32_t qs[K], ps[K];
      KZG_UNROLL_FULL
      for (int t = 0; t < K; t++) {
        qs[t] = q[t];
        ps[t] = f30_p(K - t);
      }
      mad30_chain<K, true>::run(A, qs, ps);
    }
    q[K] = f30_sbfe((uint32_t)A * (uint32_t)KZG_FP30_INV);
    const int32_t p0 = f30_p(0);
    mad30_chain<1, true>::run(A, &q[K], &p0);
    A >>= 30;
  } else {
    constexpr int ninj = (C0 != 0 ? 1 : 0) + (C1 != 0 ? 1 : 0);  // injected limbs ride at the end of the reduction chain
    int32_t qs[cnt + ninj], ps[cnt + ninj];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      qs[t] = q[i0 + t];
      ps[t] = f30_p(K - i0 - t);
    }
    if constexpr (C0 != 0) {
      qs[cnt] = inj0.l[K - N_];
      ps[cnt] = C0;
    }
    if constexpr (C1 != 0) {
      qs[cnt + ninj - 1] = inj1.l[K - N_];
      ps[cnt + ninj - 1] = C1;
    }
    mad30_chain<cnt + ninj, true>::run(A, qs, ps);
    if constexpr (UFORM) {
      r.l[K - N_] = (int32_t)((uint32_t)A & F30_MASK);
      A >>= 30;
    } else {
      r.l[K - N_] = f30_sbfe((uint32_t)A);
      A = (A + (int64_t)F30_H) >> 30;
    }
  }
  if constexpr (K + 1 < 2 * N_ - 1) f30_column<SQR, TWO, C0, C1, UFORM, K + 1>(A, q, r, a, a2, b, c, d, inj0, inj1);
}
template <bool SQR, bool TWO, int C0 = 0, int C1 = 0, bool UFORM = false>
KZG_HD void f30_mul_core(fp30& r, const fp30& a, const fp30& b, const fp30& c, const fp30& d, const fp30& inj0, const fp30& inj1) {
  constexpr int N_ = F30_N;
  int32_t q[N_];
  int32_t a2[N_];
  if (SQR) {
    KZG_UNROLL_FULL
    for (int i = 0; i < N_; i++) a2[i] = a.l[i] << 1;
  }
  int32_t top = 0;  // the injected values' limb 12 joins the carry out of column 24 (read before r.l[12] is written: r may alias)
  if (C0 != 0) top += C0 * inj0.l[N_ - 1];
  if (C1 != 0) top += C1 * inj1.l[N_ - 1];
  int64_t A = 0;
  f30_column<SQR, TWO, C0, C1, UFORM, 0>(A, q, r, a, a2, b, c, d, inj0, inj1);
  r.l[N_ - 1] = (int32_t)A + top;  // column 25 holds only the carry
}
#else
template <bool SQR, bool TWO, int C0 = 0, int C1 = 0, bool UFORM = false>
KZG_HD void f30_mul_core(fp30& r, const fp30& a, const fp30& b, const fp30& c, const fp30& d, const fp30& inj0, const fp30& inj1) {
  f30_mul_core_c<SQR, TWO, C0, C1, UFORM>(r, a, b, c, d, inj0, inj1);
}
#endif

KZG_HD void f30_mul(fp30& r, const fp30& a, const fp30& b) { f30_mul_core<false, false>(r, a, b, a, b, a, a); }   // C x C or L x C
KZG_HD void f30_sqr(fp30& r, const fp30& a) { f30_mul_core<true, false>(r, a, a, a, a, a, a); }                   // C
KZG_HD void f30_mul2(fp30& r, const fp30& a, const fp30& b, const fp30& c, const fp30& d) { f30_mul_core<false, true>(r, a, b, c, d, a, a); }  // all C
// U-form result (f30_mul_core_c's header): for a value whose every later use is a product with a C-form (the accumulator's ZZ, ZZZ)
KZG_HD void f30_mul_u(fp30& r, const fp30& a, const fp30& b) { f30_mul_core<false, false, 0, 0, true>(r, a, b, a, b, a, a); }   // C x C, L x C or U x C
// with injection (f30_mul_core_c's header): r = a b / 2^390 + C0 i0, and r = a^2 / 2^390 + C0 i0 + C1 i1, C-form
template <int C0>
KZG_HD void f30_mul_inj(fp30& r, const fp30& a, const fp30& b, const fp30& i0) { f30_mul_core<false, false, C0, 0>(r, a, b, a, b, i0, i0); }
template <int C0, int C1>
KZG_HD void f30_sqr_inj2(fp30& r, const fp30& a, const fp30& inj0, const fp30& inj1) { f30_mul_core<true, false, C0, C1>(r, a, a, a, a, inj0, inj1); }

// ---- limb-wise operations ---------------------------------------------------------------------------------------------------
KZG_HD void f30_sub(fp30& r, const fp30& a, const fp30& b) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) r.l[i] = a.l[i] - b.l[i];
}
KZG_HD void f30_add(fp30& r, const fp30& a, const fp30& b) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) r.l[i] = a.l[i] + b.l[i];
}
KZG_HD void f30_neg(fp30& r, const fp30& a) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) r.l[i] = -a.l[i];
}
// One carry pass, all limbs at once (no serial chain): limb i keeps its centred low 30 bits and hands round(l / 2^30) to limb
// i + 1.  For limbs within +-2^31 (any sum of up to four C-forms) the result is C-form: limbs 0..11 within +-(2^29 + 2).
// WIDE: for limbs anywhere in int32 (X3 = R^2 - PPP - 2Q reaches 2^31 - 1, where l + 2^29 would wrap): the carry as
// ((l >> 1) + 2^28) >> 29, the same number (the two differ only for l = 2^29 mod 2^30 with l odd: never), one instruction more.
template <bool WIDE = false>
KZG_HD void f30_carry(fp30& a) {
  int32_t c[F30_N - 1];
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N - 1; i++) {
    if (WIDE) {
      c[i] = ((a.l[i] >> 1) + (F30_H >> 1)) >> 29;
    } else {
      F30_LIMBCHK(a.l[i], ((int64_t)1 << 31) - F30_H - 1);  // l + 2^29 must not wrap
      c[i] = (a.l[i] + F30_H) >> 30;
    }
  }
  KZG_UNROLL_FULL
  for (int i = F30_N - 1; i >= 1; i--) {
    const int32_t lo = (i < F30_N - 1) ? f30_sbfe((uint32_t)a.l[i]) : a.l[i];
    a.l[i] = lo + c[i - 1];
  }
  a.l[0] = f30_sbfe((uint32_t)a.l[0]);
}

KZG_HD fp30 f30_one() {
  fp30 r;
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) r.l[i] = f30_one_limb(i);
  return r;
}

// Cheap necessary condition for a == 0 (mod p) when |a| < 8 p: a = k p with |k| <= 7, and the low 30 bits of a are exact
// whatever the carries, so k = a.l[0] * p^-1 mod 2^30 (centred) must be that small.  False positives: 15 * 2^-30 of all inputs.
KZG_HD bool f30_maybe_zero(const fp30& a) {
  const int32_t k = f30_sbfe((uint32_t)a.l[0] * (uint32_t)KZG_FP30_PINV);
  return k >= -7 && k <= 7;
}
// exact: a product by the plain ONE maps a to a * 2^-390 in (-0.53 p, 0.53 p), where 0 (mod p) is exactly 0 -- and the centred
// digits of 0 are all zero
KZG_HD bool f30_is_zero_exact(const fp30& a) {
  fp30 one, t;
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) one.l[i] = i == 0 ? 1 : 0;
  f30_mul(t, a, one);
  int32_t z = 0;
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) z |= t.l[i];
  return z == 0;
}
KZG_HD bool f30_is_zero(const fp30& a) { return f30_maybe_zero(a) && f30_is_zero_exact(a); }

// ---- representation changes ---------------------------------------------------------------------------------------------------
// TABLE FORMAT of a field element for this kernel: the canonical residue of x * 2^390 as thirteen centred digits PACKED into
// 48 bytes -- digits 0..11 as 30-bit two's-complement fields at bit 30 i, digit 12 (|.| < 2^22) as the top 24 bits.
KZG_HD void f30_unpack(fp30& r, const uint32_t* w) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N - 1; i++) {
    const int bit = 30 * i, k = bit >> 5, s = bit & 31;
    uint32_t v;
    if (s + 30 <= 32) {
      v = w[k] >> s;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
      v = __builtin_amdgcn_alignbit(w[k + 1], w[k], s);
#else
      v = (w[k] >> s) | (w[k + 1] << (32 - s));
#endif
    }
    r.l[i] = f30_sbfe(v);
  }
  r.l[F30_N - 1] = (int32_t)w[11] >> 8;
}
// C-form value with |value| small enough that digit 12 fits 24 bits (any canonical residue) -> the packed table format.
// Build-time only (table construction): serial centring first.
KZG_HD void f30_pack(uint32_t* w, const fp30& a_in) {
  fp30 a = a_in;
  int32_t c = 0;
  for (int i = 0; i < F30_N - 1; i++) {  // serial: every digit exactly in [-2^29, 2^29)
    const int32_t t = a.l[i] + c;
    a.l[i] = f30_sbfe((uint32_t)t);
    c = (t - a.l[i]) >> 30;
  }
  a.l[F30_N - 1] += c;
  for (int k = 0; k < 12; k++) w[k] = 0;
  for (int i = 0; i < F30_N - 1; i++) {
    const int bit = 30 * i, k = bit >> 5, s = bit & 31;
    const uint32_t v = (uint32_t)a.l[i] & F30_MASK;
    w[k] |= v << s;
    if (s + 30 > 32) w[k + 1] |= v >> (32 - s);
  }
  w[11] |= (uint32_t)a.l[F30_N - 1] << 8;
}
// canonical 12 x 32-bit limbs (value < p) -> C-form digits of the same integer
KZG_HD void f30_from_bn(fp30& r, const fp_t& a) {
  uint32_t u[F30_N];
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) {
    const int bit = 30 * i, k = bit >> 5, s = bit & 31;
    uint64_t v = (uint64_t)a.v[k < 12 ? k : 11] >> s;
    if (k >= 12) v = 0;
    if (k + 1 < 12 && s + 30 > 32) v |= (uint64_t)a.v[k + 1] << (32 - s);
    u[i] = (uint32_t)v & F30_MASK;
  }
  int32_t c = 0;
  for (int i = 0; i < F30_N - 1; i++) {
    const int32_t t = (int32_t)u[i] + c;
    r.l[i] = f30_sbfe((uint32_t)t);
    c = (t - r.l[i]) >> 30;
  }
  r.l[F30_N - 1] = (int32_t)u[F30_N - 1] + c;
}
// x * 2^390 (any C- or L-form value, |value| < 2^385) -> canonical x * 2^384 in 12 x 32 limbs (field.cuh's Montgomery form)
KZG_HD void f30_to_fp(fp_t& r, const fp30& a) {
  fp30 k, t;
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) k.l[i] = f30_r384_limb(i);
  f30_mul(t, a, k);  // x * 2^384 in (-0.53 p, 0.53 p)
  // + p: positive, < 1.53 p; then floor digits, 32-bit limbs, one conditional subtraction
  uint32_t u[F30_N];
  int32_t c = 0;
  for (int i = 0; i < F30_N; i++) {
    const int32_t s = t.l[i] + f30_p(i) + c;
    if (i < F30_N - 1) {
      u[i] = (uint32_t)s & F30_MASK;
      c = s >> 30;
    } else {
      u[i] = (uint32_t)s;  // >= 0, < 2^23
    }
  }
  KZG_UNROLL_FULL
  for (int wq = 0; wq < 12; wq++) {
    const int bit = 32 * wq, i = bit / 30, s = bit % 30;
    uint64_t v = (uint64_t)u[i] >> s;
    if (i + 1 < F30_N) v |= (uint64_t)u[i + 1] << (30 - s);
    if (i + 2 < F30_N && 60 - s < 32) v |= (uint64_t)u[i + 2] << (60 - s);
    r.v[wq] = (uint32_t)v;
  }
  canonicalize<FpParams>(r);
}
// canonical x * 2^384 -> the packed table format (x * 2^390); build-time
KZG_HD void fp_to_packed30(uint32_t* w, const fp_t& a) {
  fp_t k, m;
  constexpr uint32_t t[12] = KZG_FP_R390_PLAIN;
  KZG_UNROLL_FULL
  for (int i = 0; i < 12; i++) k.v[i] = t[i];
  fp_mul(m, a, k);  // canonical x * 2^390 mod p
  fp30 c;
  f30_from_bn(c, m);
  f30_pack(w, c);
}
// table entry -> operands of xyzz30_madd: x, and y or -y
KZG_HD void f30_load_entry(fp30& x, fp30& y, const uint32_t* wx, const uint32_t* wy, bool neg) {
  f30_unpack(x, wx);
  fp30 t;
  f30_unpack(t, wy);
  const uint32_t m = neg ? 0xffffffffu : 0u, one = neg ? 1u : 0u;
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) y.l[i] = (int32_t)(((uint32_t)t.l[i] ^ m) + one);  // two's complement, one v_xad_u32 per limb
}

// ---- XYZZ accumulator ---------------------------------------------------------------------------------------------------------
// The accumulator stands for the point s * (x / zz, y / zzz) with s = -1 when yneg is set: the fast addition leaves the NEGATED
// sum behind (its last product is then a sum of two products instead of a difference, which would need a negated operand) and
// flips the flag; an entry to add is therefore loaded with the sign xyzz30_entry_neg() gives.  Doubling and the complete addition
// work on the raw coordinates and keep the flag.
// Invariant between additions: y C-form, zz and zzz C- or U-form, |value| < 0.53 p; x L-form (a sum of two C-forms), |value| < 3.3 p.
struct g1_xyzz30 {
  fp30 x, y, zz, zzz;
  uint32_t inf;
  uint32_t yneg;
};
KZG_HD void xyzz30_set_inf(g1_xyzz30& p) {
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) p.x.l[i] = p.y.l[i] = p.zz.l[i] = p.zzz.l[i] = 0;
  p.inf = 1;
  p.yneg = 0;
}
// the sign to load a table entry with (f30_load_entry's neg) so that the raw coordinates add it with the accumulator's sign
KZG_HD bool xyzz30_entry_neg(const g1_xyzz30& p, bool neg) { return neg != (p.yneg != 0u); }

// p = 2 * (x, y), (x, y) finite, both C-form   (mdbl-2008-s-1, a = 0)
KZG_HD void xyzz30_mdbl(g1_xyzz30& p, const fp30& x, const fp30& y) {
  if (f30_is_zero_exact(y)) {  // order-2 point: not in this curve's group, handled for completeness
    xyzz30_set_inf(p);
    return;
  }
  fp30 u, v, w, s, m, t, x3;
  f30_add(u, y, y);
  f30_carry(u);
  f30_sqr(v, u);
  f30_mul(w, u, v);
  f30_mul(s, x, v);
  f30_sqr(m, x);
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) m.l[i] *= 3;
  f30_carry(m);
  f30_sqr(x3, m);
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) x3.l[i] -= 2 * s.l[i];  // X3 = M^2 - 2S
  f30_carry(x3);
  f30_sub(t, s, x3);
  f30_carry(t);
  fp30 nw;
  f30_neg(nw, w);
  f30_mul2(p.y, m, t, nw, y);  // Y3 = M (S - X3) - W y
  p.x = x3;
  p.zz = v;
  p.zzz = w;
  p.inf = 0;
}

// p += +-(x2, .) for a FINITE accumulator p and the generic case; x2, y2 C-form: a table entry, y2 with the sign
// xyzz30_entry_neg() asked for.  Returns false -- with p untouched -- when x2 * ZZ1 == X1 (mod p) may hold (P + P or P + (-P);
// 15 * 2^-30 of all calls are false alarms); the caller then runs xyzz30_madd_complete.
// madd-2008-s, 6 products + 2 squarings + 1 double product and NO carry pass: the three differences are injected into the
// products that feed them (f30_mul_inj) and come out C-form --
//   P = X2 ZZ1 - X1,  R = Y2 ZZZ1 - Y1,  V = R^2 - PPP - 3 Q = X3 - Q;   X3 = V + Q (L-form: only ever a product's first operand
//   or an injected value);  -Y3 = R V + Y1 PPP as ONE reduction: the raw result is the negated sum, the flag says so.
KZG_HD bool xyzz30_madd_fast(g1_xyzz30& p, const fp30& x2, const fp30& y2) {
  fp30 u, r, pp, ppp, v;
  f30_mul_inj<-1>(u, x2, p.zz, p.x);        // P (|P| < 3.9 p)
  if (f30_maybe_zero(u)) return false;
  f30_mul_inj<-1>(r, y2, p.zzz, p.y);       // R
  f30_sqr(pp, u);                           // PP
  f30_mul(ppp, u, pp);                      // PPP
  f30_mul_u(p.zz, p.zz, pp);                // ZZ3   (U-form: ZZ only ever meets a table entry's x and PP)
  f30_mul_u(p.zzz, p.zzz, ppp);             // ZZZ3  (U-form: ZZZ only ever meets a table entry's y and PPP)
  f30_mul(pp, p.x, pp);                     // Q = X1 PP (L x C; X1 and PP are dead from here)
  f30_sqr_inj2<-1, -3>(v, r, ppp, pp);      // V = X3 - Q (|V| < 2.7 p)
  f30_add(p.x, v, pp);                      // X3
  f30_mul2(p.y, r, v, p.y, ppp);            // -Y3
  p.yneg ^= 1u;
  return true;
}

// Complete addition (identity, P + P, P + (-P), and the generic case): the out-of-line companion of xyzz30_madd_fast.
KZG_HD_NOINLINE void xyzz30_madd_complete(g1_xyzz30& p, const fp30& x2, const fp30& y2) {
  if (p.inf) {
    const fp30 one = f30_one();
    p.x = x2;
    p.y = y2;  // with the sign the flag asked for
    p.zz = one;
    p.zzz = one;
    p.inf = 0;
    return;
  }
  if (xyzz30_madd_fast(p, x2, y2)) return;
  f30_carry(p.x);  // L-form -> C-form: the formulas below are the plain ones (raw coordinates, the sign flag stays)
  fp30 u2, r;
  f30_mul(u2, x2, p.zz);
  f30_mul(r, y2, p.zzz);
  f30_sub(u2, u2, p.x);
  f30_carry(u2);
  f30_sub(r, r, p.y);
  f30_carry(r);
  if (f30_is_zero_exact(u2)) {
    if (f30_is_zero_exact(r))
      xyzz30_mdbl(p, x2, y2);
    else
      xyzz30_set_inf(p);
    return;
  }
  // false alarm of the cheap test: the generic formulas apply
  fp30 pp, ppp;
  f30_sqr(pp, u2);
  f30_mul(ppp, u2, pp);
  f30_mul(p.zz, p.zz, pp);
  f30_mul(p.zzz, p.zzz, ppp);
  f30_mul(pp, p.x, pp);
  f30_sqr(p.x, r);
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) p.x.l[i] = p.x.l[i] - ppp.l[i] - 2 * pp.l[i];
  f30_carry<true>(p.x);
  f30_sub(pp, pp, p.x);
  f30_carry(pp);
  f30_neg(p.y, p.y);
  f30_mul2(p.y, r, pp, p.y, ppp);
}

// p = 2 p   (dbl-2008-s-1, a = 0) on an accumulator under the invariant; the invariant holds again afterwards
KZG_HD void xyzz30_dbl_inl(g1_xyzz30& p) {
  if (p.inf) return;
  if (f30_is_zero(p.y)) {  // a point of order two
    xyzz30_set_inf(p);
    return;
  }
  fp30 u, v, w, s, m, t, x3;
  f30_carry(p.x);  // L-form -> C-form: it is squared
  f30_add(u, p.y, p.y);
  f30_carry(u);
  f30_sqr(v, u);
  f30_mul(w, u, v);
  f30_mul(s, p.x, v);
  f30_sqr(m, p.x);
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) m.l[i] *= 3;  // 3 X^2: limbs within +-(3 * 2^29)
  f30_carry(m);
  f30_sqr(x3, m);
  KZG_UNROLL_FULL
  for (int i = 0; i < F30_N; i++) x3.l[i] -= 2 * s.l[i];  // X3 = M^2 - 2S
  f30_carry(x3);
  f30_sub(t, s, x3);
  f30_carry(t);
  fp30 nw;
  f30_neg(nw, w);
  f30_mul2(p.y, m, t, nw, p.y);  // Y3 = M (S - X3) - W Y1
  p.x = x3;
  f30_mul(p.zz, p.zz, v);
  f30_mul(p.zzz, p.zzz, w);
}
KZG_HD_NOINLINE void xyzz30_dbl(g1_xyzz30& p) { xyzz30_dbl_inl(p); }

// accumulator -> the 12 x 32-limb XYZZ format of g1.cuh (2^384 Montgomery, canonical)
KZG_HD void xyzz30_to_xyzz(g1_xyzz& r, const g1_xyzz30& p) {
  if (p.inf) {
    xyzz_set_inf(r);
    return;
  }
  f30_to_fp(r.x, p.x);
  f30_to_fp(r.y, p.y);
  if (p.yneg) fp_neg(r.y, r.y);
  f30_to_fp(r.zz, p.zz);
  f30_to_fp(r.zzz, p.zzz);
}

}  // namespace kzg
