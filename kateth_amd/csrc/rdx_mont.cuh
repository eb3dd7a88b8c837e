// Carry-free radix-2^W Montgomery multiplication (W < 32) for gfx950 -- the generic part shared by fp28.cuh
// (Fp: 14 limbs of 28 bits) and fr29.cuh (Fr: 9 limbs of 29 bits).
//
// gfx950's only wide integer multiply is v_mad_u64_u32 (32x32 + 64 -> 64, full rate).  With saturated 32-bit limbs
// (field.cuh) the 64-bit column accumulator overflows after one product, so every product needs a second instruction
// (v_addc_co_u32 into a third word).  With limbs of W < 32 bits a whole column of 2N products fits one 64-bit
// accumulator: N^2 + N^2 v_mad_u64_u32 and NO carry instruction per product, squarings do N(N+1)/2 products, two
// products can share ONE reduction (rdx_mul2), and additions/subtractions are N independent 32-bit operations
// (limbs are allowed to grow; the bounds are tracked statically in the callers' comments and re-checked at run time
// in the CPU test build by KZG_FP28_CHECK, tests/test_hostmath.py).
//
// P provides: N (limbs), W (bits per limb), INV (-m^-1 mod 2^W), mod(i) (limbs of the modulus).
// Value = sum l[i] * 2^(W i); Montgomery radix 2^(N W); "N-form" = output of a product: limbs 0..N-2 < 2^W,
// value < 2m provided the operands' value bounds multiply to less than 2^(N W) / m.
#pragma once
#include "field.cuh"

namespace kzg {

template <class P>
struct rdx_t {
  uint32_t l[P::N];
};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(KZG_FP28_CHECK)
// CPU test build: every column sum is also formed in 128 bits and every limb subtraction is
// checked, so a violated bound aborts the test instead of silently wrapping.
extern "C" void kzg_fp28_check_failed(const char* what);
struct f28_col {
  unsigned __int128 wide;
  uint64_t v;
};
#define RDX_COL_INIT(A) \
  f28_col A { 0, 0 }
#define RDX_MAC(A, x, y)                                            \
  do {                                                              \
    (A).wide += (unsigned __int128)(uint32_t)(x) * (uint32_t)(y);   \
    (A).v += (uint64_t)(uint32_t)(x) * (uint32_t)(y);               \
    if ((A).wide >> 64) kzg_fp28_check_failed("column overflow");   \
  } while (0)
#define RDX_LO(A) ((uint32_t)(A).v)
#define RDX_SHIFT(A)      \
  do {                    \
    (A).v >>= W_;      \
    (A).wide = (A).v;     \
  } while (0)
#define RDX_SUBCHK(a, m, b)                                                              \
  do {                                                                                   \
    if ((uint64_t)(a) + (uint64_t)(m) < (uint64_t)(b)) kzg_fp28_check_failed("limb underflow"); \
    if ((uint64_t)(a) + (uint64_t)(m) - (uint64_t)(b) >> 32) kzg_fp28_check_failed("limb overflow"); \
  } while (0)
#define RDX_ADDCHK(a, b)                                                          \
  do {                                                                            \
    if (((uint64_t)(a) + (uint64_t)(b)) >> 32) kzg_fp28_check_failed("limb overflow"); \
  } while (0)
#else
#define RDX_COL_INIT(A) uint64_t A = 0
#define RDX_MAC(A, x, y) (A) += (uint64_t)(uint32_t)(x) * (uint32_t)(y)
#define RDX_LO(A) ((uint32_t)(A))
#define RDX_SHIFT(A) (A) >>= W_
#define RDX_SUBCHK(a, m, b) \
  do {                      \
  } while (0)
#define RDX_ADDCHK(a, b) \
  do {                   \
  } while (0)
#endif


// ---- representation changes -------------------------------------------------------------------------
// NB x 32-bit limbs -> N x W-bit limbs, strictly normalised (the value must fit N*W bits)
template <class P, int NB>
KZG_HD void rdx_from_bn(rdx_t<P>& r, const bn<NB>& a) {
  // 32-bit funnel shifts only: a 64-bit (hi:lo) >> s makes hipcc spill the source limbs to scratch and re-read them
  // as unaligned 64-bit loads
  constexpr uint32_t MASK_ = (1u << P::W) - 1u;
  KZG_UNROLL_FULL
  for (int i = 0; i < P::N; i++) {
    const int bit = P::W * i, w = bit >> 5, s = bit & 31;
    const uint32_t lo = (w < NB) ? a.v[w < NB ? w : 0] : 0u;
    uint32_t v;
    if (s + P::W <= 32) {
      v = lo >> s;
    } else {
      const uint32_t hi = (w + 1 < NB) ? a.v[w + 1 < NB ? w + 1 : 0] : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
      v = __builtin_amdgcn_alignbit(hi, lo, s);
#else
      v = (lo >> s) | (hi << (32 - s));
#endif
    }
    r.l[i] = v & MASK_;
  }
}
// strictly normalised limbs (all < 2^W) and value < 2^(32 NB) -> NB x 32-bit limbs
template <class P, int NB>
KZG_HD void rdx_to_bn(bn<NB>& r, const rdx_t<P>& a) {
  KZG_UNROLL_FULL
  for (int w = 0; w < NB; w++) {
    const int bit = 32 * w, i = bit / P::W, s = bit % P::W;
    uint64_t v = (i < P::N) ? ((uint64_t)a.l[i < P::N ? i : 0] >> s) : 0;
    if (i + 1 < P::N) v |= (uint64_t)a.l[i + 1 < P::N ? i + 1 : 0] << (P::W - s);
    if (i + 2 < P::N && 2 * P::W - s < 32) v |= (uint64_t)a.l[i + 2 < P::N ? i + 2 : 0] << (2 * P::W - s);
    r.v[w] = (uint32_t)v;
  }
}

// ---- Montgomery products ------------------------------------------------------------------------
// r = (a*b [+ c*d]) / 2^392 mod p, N-form.  Requires 14*(La*Lb [+ Lc*Ld]) + 14*2^56 < 2^64 for the limb
// bounds L, and (Va*Vb [+ Vc*Vd]) < 2^11 for the value bounds V in units of p (so the result is < 2p).
template <class P, bool SQR, bool TWO>
KZG_HD void rdx_mul_core_c(rdx_t<P>& r, const rdx_t<P>& a, const rdx_t<P>& b, const rdx_t<P>& c, const rdx_t<P>& d) {
  constexpr int N_ = P::N, W_ = P::W;
  constexpr uint32_t MASK_ = (1u << W_) - 1u;
  (void)W_;
  uint32_t q[N_];
  uint32_t a2[N_];
  if (SQR) {
    KZG_UNROLL_FULL
    for (int i = 0; i < N_; i++) {
      RDX_ADDCHK(a.l[i], a.l[i]);
      a2[i] = a.l[i] << 1;
    }
  }
  RDX_COL_INIT(A);
  KZG_UNROLL_FULL
  for (int k = 0; k < 2 * N_; k++) {
    const int i0 = (k < N_) ? 0 : k - N_ + 1;
    const int i1 = (k < N_) ? k : N_ - 1;
    if (SQR) {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) {
        const int j = k - i;
        if (i < j) RDX_MAC(A, a2[i], a.l[j]);
        if (i == j) RDX_MAC(A, a.l[i], a.l[i]);
      }
    } else {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) RDX_MAC(A, a.l[i], b.l[k - i]);
    }
    if (TWO) {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) RDX_MAC(A, c.l[i], d.l[k - i]);
    }
    if (k < N_) {
      KZG_UNROLL_FULL
      for (int i = 0; i < k; i++) RDX_MAC(A, q[i], P::mod(k - i));
      q[k] = (P::INV == MASK_) ? ((0u - RDX_LO(A)) & MASK_) : ((RDX_LO(A) * (uint32_t)P::INV) & MASK_);
      RDX_MAC(A, q[k], P::mod(0));
      RDX_SHIFT(A);
    } else {
      KZG_UNROLL_FULL
      for (int i = i0; i <= i1; i++) RDX_MAC(A, q[i], P::mod(k - i));
      if (k < 2 * N_ - 1) {
        r.l[k - N_] = RDX_LO(A) & MASK_;  // r may alias an operand: limb k-14 of every operand was last read in column k-1
        RDX_SHIFT(A);
      } else {
        r.l[N_ - 1] = RDX_LO(A);  // value < 2p: the top limb is small, nothing above it
      }
    }
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device version: every column is issued as explicit v_mad_u64_u32 chains (mac_asm.cuh) that START from the carry of
// the previous column.  Left to itself hipcc starts each column from zero and adds the carry with an extra
// v_lshl_add_u64 (26 per product, 5 % of the hot loop).  Same arithmetic as rdx_mul_core_c, which the CPU tests run.
// Columns are template instances so that every operand list has a compile-time length.
template <class P, bool SQR, bool TWO, int K>
KZG_HD void rdx_column(uint64_t& A, uint32_t* q, rdx_t<P>& r, const rdx_t<P>& a, const uint32_t* a2, const rdx_t<P>& b, const rdx_t<P>& c, const rdx_t<P>& d) {
  constexpr int N_ = P::N, W_ = P::W;
  constexpr uint32_t MASK_ = (1u << W_) - 1u;
  constexpr int i0 = (K < N_) ? 0 : K - N_ + 1;
  constexpr int i1 = (K < N_) ? K : N_ - 1;
  constexpr int cnt = i1 - i0 + 1;
  if constexpr (SQR) {
    // pairs i < j with i + j = K:  i = i0 .. (K-1)/2 ;  diagonal when K is even
    constexpr int last_pair = (K - 1) / 2;
    constexpr int npairs = (K >= 1 && last_pair >= i0) ? last_pair - i0 + 1 : 0;
    constexpr int diag = (K % 2 == 0) ? 1 : 0;
    uint32_t xs[npairs + diag], ys[npairs + diag];
    KZG_UNROLL_FULL
    for (int t = 0; t < npairs; t++) {
      xs[t] = a2[i0 + t];
      ys[t] = a.l[K - i0 - t];
    }
    if constexpr (diag) {
      xs[npairs] = a.l[K / 2];
      ys[npairs] = a.l[K / 2];
    }
    mad28_chain<npairs + diag, false>::run(A, xs, ys);
  } else {
    uint32_t xs[cnt], ys[cnt];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      xs[t] = a.l[i0 + t];
      ys[t] = b.l[K - i0 - t];
    }
    mad28_chain<cnt, false>::run(A, xs, ys);
  }
  if constexpr (TWO) {
    uint32_t xs[cnt], ys[cnt];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      xs[t] = c.l[i0 + t];
      ys[t] = d.l[K - i0 - t];
    }
    mad28_chain<cnt, false>::run(A, xs, ys);
  }
  if constexpr (K < N_) {
    if constexpr (K > 0) {
      uint32_t qs[K], ps[K];
      KZG_UNROLL_FULL
      for (int t = 0; t < K; t++) {
        qs[t] = q[t];
        ps[t] = P::mod(K - t);
      }
      mad28_chain<K, true>::run(A, qs, ps);
    }
    q[K] = (P::INV == MASK_) ? ((0u - (uint32_t)A) & MASK_) : (((uint32_t)A * (uint32_t)P::INV) & MASK_);
    const uint32_t p0 = P::mod(0);
    mad28_chain<1, true>::run(A, &q[K], &p0);
    A >>= W_;
  } else {
    uint32_t qs[cnt], ps[cnt];
    KZG_UNROLL_FULL
    for (int t = 0; t < cnt; t++) {
      qs[t] = q[i0 + t];
      ps[t] = P::mod(K - i0 - t);
    }
    mad28_chain<cnt, true>::run(A, qs, ps);
    r.l[K - N_] = (uint32_t)A & MASK_;  // r may alias an operand: limb K-14 of every operand was last read in column K-1
    A >>= W_;
  }
  if constexpr (K + 1 < 2 * N_ - 1) rdx_column<P, SQR, TWO, K + 1>(A, q, r, a, a2, b, c, d);
}
template <class P, bool SQR, bool TWO>
KZG_HD void rdx_mul_core(rdx_t<P>& r, const rdx_t<P>& a, const rdx_t<P>& b, const rdx_t<P>& c, const rdx_t<P>& d) {
  constexpr int N_ = P::N;
  uint32_t q[N_];
  uint32_t a2[N_];
  if (SQR) {
    KZG_UNROLL_FULL
    for (int i = 0; i < N_; i++) a2[i] = a.l[i] << 1;
  }
  uint64_t A = 0;
  rdx_column<P, SQR, TWO, 0>(A, q, r, a, a2, b, c, d);
  r.l[N_ - 1] = (uint32_t)A;  // column 27 holds only the carry; value < 2p: the top limb is small
}
#else
template <class P, bool SQR, bool TWO>
KZG_HD void rdx_mul_core(rdx_t<P>& r, const rdx_t<P>& a, const rdx_t<P>& b, const rdx_t<P>& c, const rdx_t<P>& d) {
  rdx_mul_core_c<P, SQR, TWO>(r, a, b, c, d);
}
#endif

}  // namespace kzg
