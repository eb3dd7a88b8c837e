// BLS12-381 prime-field arithmetic for gfx950: Fp (381-bit, 12 x u32 limbs) and
// Fr (255-bit, 8 x u32 limbs), Montgomery form, fully reduced representatives.
//
// Replaces the blst entry points kateth reaches through src/bls.rs:8-19
// (blst_fr_add/sub/mul/cneg/eucl_inverse/from_scalar/..., and the blst_fp code
// behind blst_p1_*).  Written for CDNA4: the multiply is a product-scanning
// (FIPS) Montgomery multiplication whose inner step is one `v_mad_u64_u32`
// (32x32+64 -> 64, carry-out in VCC) plus one `v_addc_co_u32` into a 96-bit
// column accumulator -- two VALU instructions per limb product, no carry
// ripple across columns, modulus limbs held in SGPRs.
//
// Every function is __host__ __device__ so tests/ can compile this exact
// source for the CPU and check it against the Python oracle without a GPU; the
// host build is a test harness only, the product never computes on the host
// with it except for the once-per-call pairing (pairing.hpp).
#pragma once
#include <stdint.h>
#include <string.h>

#include "consts.cuh"
#include "host_fp_mulx.hpp"  // x86-64 hosts: the mulx / adcx / adox Montgomery product (generated)

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_HD __host__ __device__ __forceinline__
#define KZG_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define KZG_HD inline __attribute__((always_inline))
#define KZG_HD_NOINLINE inline __attribute__((noinline))
#endif

namespace kzg {

template <int N>
struct bn {
  uint32_t v[N];
};

struct FpParams {
  static constexpr int N = 12;
  static constexpr uint32_t INV = KZG_FP_INV32;
  static constexpr uint64_t INV64 = KZG_FP_INV64;
  KZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t t[N] = KZG_FP_MOD;
    return t[i];
  }
  KZG_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t t[N] = KZG_FP_ONE;
    return t[i];
  }
  KZG_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t t[N] = KZG_FP_R2;
    return t[i];
  }
  KZG_HD static constexpr uint32_t mod_minus_2(int i) {
    constexpr uint32_t t[N] = KZG_FP_MOD_MINUS_2;
    return t[i];
  }
  KZG_HD static constexpr uint32_t half(int i) {
    constexpr uint32_t t[N] = KZG_FP_HALF;
    return t[i];
  }
  KZG_HD static constexpr uint32_t mod2(int i) {
    constexpr uint32_t t[N] = KZG_FP_MOD2;
    return t[i];
  }
};

struct FrParams {
  static constexpr int N = 8;
  static constexpr uint32_t INV = KZG_FR_INV32;
  static constexpr uint64_t INV64 = KZG_FR_INV64;
  KZG_HD static constexpr uint32_t mod(int i) {
    constexpr uint32_t t[N] = KZG_FR_MOD;
    return t[i];
  }
  KZG_HD static constexpr uint32_t one(int i) {
    constexpr uint32_t t[N] = KZG_FR_ONE;
    return t[i];
  }
  KZG_HD static constexpr uint32_t r2(int i) {
    constexpr uint32_t t[N] = KZG_FR_R2;
    return t[i];
  }
  KZG_HD static constexpr uint32_t mod_minus_2(int i) {
    constexpr uint32_t t[N] = KZG_FR_MOD_MINUS_2;
    return t[i];
  }
  KZG_HD static constexpr uint32_t half(int i) {
    constexpr uint32_t t[N] = KZG_FR_HALF;
    return t[i];
  }
  KZG_HD static constexpr uint32_t mod2(int i) {
    constexpr uint32_t t[N] = KZG_FR_MOD2;
    return t[i];
  }
};

using fp_t = bn<12>;
using fr_t = bn<8>;

// ---------------------------------------------------------------------------
// 96-bit column accumulator
// ---------------------------------------------------------------------------
struct acc96 {
  uint64_t lo;
  uint32_t hi;
};

// A += a*b
KZG_HD void mac(acc96& A, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(A.lo), "+v"(A.hi)
      : "v"(a), "v"(b)
      : "vcc");
#else
  uint64_t pr = (uint64_t)a * b;
  uint64_t lo = A.lo + pr;
  A.hi += (lo < pr);
  A.lo = lo;
#endif
}

KZG_HD void acc_shift(acc96& A) {
  A.lo = (A.lo >> 32) | ((uint64_t)A.hi << 32);
  A.hi = 0;
}

}  // namespace kzg
#include "mac_asm.cuh"
namespace kzg {

#define KZG_UNROLL_FULL _Pragma("unroll")

// compile-time-bounded dispatcher: `count` is a constant after unrolling, so
// exactly one branch survives; chains longer than 12 split in two.
template <int MAXN, bool VS>
KZG_HD void mac_list_dyn(acc96& A, const uint32_t* x, const uint32_t* y, int count) {
#if defined(__HIP_DEVICE_COMPILE__)
  switch (count) {
    case 0: break;
    case 1: mac_list<1, VS>(A, x, y); break;
    case 2: mac_list<2, VS>(A, x, y); break;
    case 3: mac_list<3, VS>(A, x, y); break;
    case 4: mac_list<4, VS>(A, x, y); break;
    case 5: mac_list<5, VS>(A, x, y); break;
    case 6: mac_list<6, VS>(A, x, y); break;
    case 7: mac_list<7, VS>(A, x, y); break;
    case 8: mac_list<8, VS>(A, x, y); break;
    case 9: mac_list<9, VS>(A, x, y); break;
    case 10: mac_list<10, VS>(A, x, y); break;
    case 11: mac_list<11, VS>(A, x, y); break;
    case 12: mac_list<12, VS>(A, x, y); break;
    default: break;
  }
#else
  for (int i = 0; i < count; i++) mac(A, x[i], y[i]);
#endif
}

// ---------------------------------------------------------------------------
// plain multi-limb helpers
// ---------------------------------------------------------------------------
template <int N>
KZG_HD bool bn_is_zero(const bn<N>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a.v[i];
  return o == 0;
}

template <int N>
KZG_HD bool bn_eq(const bn<N>& a, const bn<N>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

template <int N>
KZG_HD void bn_zero(bn<N>& a) {
#pragma unroll
  for (int i = 0; i < N; i++) a.v[i] = 0;
}

// r = a + b, returns carry.  Device: native carry chain (v_add_co / v_addc_co).
template <int N>
KZG_HD uint32_t bn_add(bn<N>& r, const bn<N>& a, const bn<N>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  return c;
#else
  if constexpr (N % 2 == 0) {  // host: the same little-endian bytes viewed as N/2 64-bit limbs
    constexpr int M = N / 2;
    uint64_t A[M], B[M], R[M];
    memcpy(A, a.v, 4 * N);
    memcpy(B, b.v, 4 * N);
    unsigned __int128 c = 0;
#pragma GCC unroll 8
    for (int i = 0; i < M; i++) {
      c += (unsigned __int128)A[i] + B[i];
      R[i] = (uint64_t)c;
      c >>= 64;
    }
    memcpy(r.v, R, 4 * N);
    return (uint32_t)c;
  } else {
    uint64_t c = 0;
    for (int i = 0; i < N; i++) {
      c += (uint64_t)a.v[i] + b.v[i];
      r.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return (uint32_t)c;
  }
#endif
}

// r = a - b, returns borrow (1 if a < b)
template <int N>
KZG_HD uint32_t bn_sub(bn<N>& r, const bn<N>& a, const bn<N>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], c, &c);
  return c;
#else
  if constexpr (N % 2 == 0) {
    constexpr int M = N / 2;
    uint64_t A[M], B[M], R[M];
    memcpy(A, a.v, 4 * N);
    memcpy(B, b.v, 4 * N);
    uint64_t borrow = 0;
#pragma GCC unroll 8
    for (int i = 0; i < M; i++) {
      const unsigned __int128 t = (unsigned __int128)A[i] - B[i] - borrow;
      R[i] = (uint64_t)t;
      borrow = (uint64_t)(t >> 64) & 1u;
    }
    memcpy(r.v, R, 4 * N);
    return (uint32_t)borrow;
  } else {
    int64_t c = 0;
    for (int i = 0; i < N; i++) {
      c += (int64_t)a.v[i] - (int64_t)b.v[i];
      r.v[i] = (uint32_t)c;
      c >>= 32;
    }
    return (uint32_t)(c & 1);
  }
#endif
}

// a >= b ?
template <int N>
KZG_HD bool bn_geq(const bn<N>& a, const bn<N>& b) {
  bn<N> t;
  return bn_sub(t, a, b) == 0;
}

template <class F>
KZG_HD bn<F::N> modulus() {
  bn<F::N> m;
#pragma unroll
  for (int i = 0; i < F::N; i++) m.v[i] = F::mod(i);
  return m;
}

template <class F>
KZG_HD bn<F::N> mont_one() {
  bn<F::N> m;
#pragma unroll
  for (int i = 0; i < F::N; i++) m.v[i] = F::one(i);
  return m;
}

// r = (t >= mod) ? t - mod : t   (carry = overflow bit above the top limb)
template <class F>
KZG_HD void reduce_once(bn<F::N>& r, const bn<F::N>& t, uint32_t carry) {
  bn<F::N> s;
  uint32_t borrow = bn_sub(s, t, modulus<F>());
  bool use_s = (carry != 0) || (borrow == 0);
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = use_s ? s.v[i] : t.v[i];
}

// ---------------------------------------------------------------------------
// modular add / sub / neg / double on fully reduced values
// ---------------------------------------------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
// host versions in 64-bit limbs (the once-per-call pairing does ~3 additions per multiplication)
template <class F>
inline void add_mod_host64(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  constexpr int M = F::N / 2;
  typedef unsigned __int128 u128;
  uint64_t A[M], B[M], T[M], D[M];
  memcpy(A, a.v, 8 * M);
  memcpy(B, b.v, 8 * M);
  u128 c = 0;
  uint64_t borrow = 0;
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) {
    c += (u128)A[i] + B[i];
    T[i] = (uint64_t)c;
    c >>= 64;
    const uint64_t p = (uint64_t)F::mod(2 * i) | ((uint64_t)F::mod(2 * i + 1) << 32);
    const u128 u = (u128)T[i] - p - borrow;
    D[i] = (uint64_t)u;
    borrow = (uint64_t)(u >> 64) & 1u;
  }
  const uint64_t mask = ((uint64_t)c != 0 || borrow == 0) ? ~0ull : 0ull;  // take T - p
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) T[i] = (D[i] & mask) | (T[i] & ~mask);
  memcpy(r.v, T, 8 * M);
}
template <class F>
inline void sub_mod_host64(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  constexpr int M = F::N / 2;
  typedef unsigned __int128 u128;
  uint64_t A[M], B[M], T[M];
  memcpy(A, a.v, 8 * M);
  memcpy(B, b.v, 8 * M);
  uint64_t borrow = 0;
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) {
    const u128 u = (u128)A[i] - B[i] - borrow;
    T[i] = (uint64_t)u;
    borrow = (uint64_t)(u >> 64) & 1u;
  }
  const uint64_t mask = borrow ? ~0ull : 0ull;  // add p back
  u128 c = 0;
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) {
    const uint64_t p = ((uint64_t)F::mod(2 * i) | ((uint64_t)F::mod(2 * i + 1) << 32)) & mask;
    c += (u128)T[i] + p;
    T[i] = (uint64_t)c;
    c >>= 64;
  }
  memcpy(r.v, T, 8 * M);
}
#endif

template <class F>
KZG_HD void add_mod(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  add_mod_host64<F>(r, a, b);
#else
  bn<F::N> t;
  uint32_t c = bn_add(t, a, b);
  reduce_once<F>(r, t, c);
#endif
}

template <class F>
KZG_HD void sub_mod(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  sub_mod_host64<F>(r, a, b);
#else
  bn<F::N> t, u;
  uint32_t borrow = bn_sub(t, a, b);
  bn_add(u, t, modulus<F>());
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = borrow ? u.v[i] : t.v[i];
#endif
}

template <class F>
KZG_HD void neg_mod(bn<F::N>& r, const bn<F::N>& a) {
  bn<F::N> t;
  bn_sub(t, modulus<F>(), a);
  bool z = bn_is_zero(a);
#pragma unroll
  for (int i = 0; i < F::N; i++) r.v[i] = z ? 0u : t.v[i];
}

template <class F>
KZG_HD void dbl_mod(bn<F::N>& r, const bn<F::N>& a) {
  add_mod<F>(r, a, a);
}

// ---------------------------------------------------------------------------
// Montgomery multiplication, finely integrated product scanning.
// r = a*b*2^(-32N) mod m ; inputs < m, output < m.
// ---------------------------------------------------------------------------
// test / measurement hook (host only): true sends the host's Fp products through the portable loop even where the mulx path is available
inline bool& host_fp_force_portable() {
  static bool force = false;
  return force;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// host instantiation: the same value representation (little-endian limbs, radix 2^(32N)) viewed
// as N/2 64-bit limbs, CIOS with unsigned __int128 -- ~3x faster than the 32-bit path on a CPU;
// only the once-per-call pairing and the tests run here.
template <class F, bool LAZY = false>
inline void mont_mul_host64(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  constexpr int M = F::N / 2;
  typedef unsigned __int128 u128;
#if defined(KZG_HOST_FP_MULX)
  // Fp on an x86-64 host with BMI2 + ADX: the generated mulx / adcx / adox product (host_fp_mulx.hpp) -- 1.8-2.4 x the portable
  // loop below, which stays as the fallback and as its cross-check (tests/test_hostmath.py::test_host_mulx_product_matches_portable).
  // Round 5: the host tail of every verification call (Horner, two Miller loops, final exponentiation) is ~6 k of these.
  if constexpr (F::N == 12 && F::INV64 == hostmulx::INV64) {
    if (hostmulx::cpu_ok() && !host_fp_force_portable()) {
      uint64_t t6[6], a6[6], b6[6];
      memcpy(a6, a.v, 48);
      memcpy(b6, b.v, 48);
      hostmulx::mont_mul_384(t6, a6, b6);  // < 2p for inputs < 2p
      if (!LAZY) {
        uint64_t d[6], borrow = 0;
        for (int i = 0; i < 6; i++) {
          const u128 u = (u128)t6[i] - hostmulx::P64[i] - borrow;
          d[i] = (uint64_t)u;
          borrow = (uint64_t)(u >> 64) & 1u;
        }
        memcpy(r.v, borrow == 0 ? d : t6, 48);
      } else {
        memcpy(r.v, t6, 48);
      }
      return;
    }
  }
#endif
  uint64_t A[M], B[M], P[M], t[M + 2];
  memcpy(A, a.v, 8 * M);
  memcpy(B, b.v, 8 * M);
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) P[i] = (uint64_t)F::mod(2 * i) | ((uint64_t)F::mod(2 * i + 1) << 32);
#pragma GCC unroll 8
  for (int i = 0; i < M + 2; i++) t[i] = 0;
#pragma GCC unroll 8
  for (int i = 0; i < M; i++) {
    uint64_t c = 0;
#pragma GCC unroll 8
    for (int j = 0; j < M; j++) {
      u128 s = (u128)A[j] * B[i] + t[j] + c;
      t[j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    u128 s = (u128)t[M] + c;
    t[M] = (uint64_t)s;
    t[M + 1] = (uint64_t)(s >> 64);
    uint64_t m = t[0] * F::INV64;
    s = (u128)m * P[0] + t[0];
    c = (uint64_t)(s >> 64);
#pragma GCC unroll 8
    for (int j = 1; j < M; j++) {
      s = (u128)m * P[j] + t[j] + c;
      t[j - 1] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    s = (u128)t[M] + c;
    t[M - 1] = (uint64_t)s;
    t[M] = t[M + 1] + (uint64_t)(s >> 64);
  }
  if (LAZY) {
    memcpy(r.v, t, 8 * M);  // inputs < 2p and 4p < radix  =>  result < 2p, no final subtraction
  } else {
    // conditional subtraction of the modulus in 64-bit limbs
    uint64_t d[M], borrow = 0;
#pragma GCC unroll 8
    for (int i = 0; i < M; i++) {
      const u128 u = (u128)t[i] - P[i] - borrow;
      d[i] = (uint64_t)u;
      borrow = (uint64_t)(u >> 64) & 1u;
    }
    const bool use_d = (t[M] != 0) || (borrow == 0);
    memcpy(r.v, use_d ? d : t, 8 * M);
  }
}
#endif

template <class F, bool LAZY>
KZG_HD void mont_mul_core(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  mont_mul_host64<F, LAZY>(r, a, b);
  return;
#endif
  constexpr int N = F::N;
  uint32_t m[N];
  uint32_t t[N];
  acc96 A{0, 0};
  // column k of a*b + m*p; operand lists are assembled at compile time so each
  // column is issued as one or two asm chains (mac_asm.cuh)
  KZG_UNROLL_FULL
  for (int k = 0; k < N; k++) {
    uint32_t xs[N], ys[N], ms[N], ps[N];
    KZG_UNROLL_FULL
    for (int j = 0; j <= k; j++) {
      xs[j] = a.v[j];
      ys[j] = b.v[k - j];
    }
    KZG_UNROLL_FULL
    for (int j = 0; j < k; j++) {
      ms[j] = m[j];
      ps[j] = F::mod(k - j);
    }
    mac_list_dyn<N, false>(A, xs, ys, k + 1);
    mac_list_dyn<N, true>(A, ms, ps, k);
    m[k] = (uint32_t)A.lo * F::INV;
    uint32_t p0 = F::mod(0);
    mac_chain<1, true>::run(A, &m[k], &p0);
    acc_shift(A);
  }
  KZG_UNROLL_FULL
  for (int k = N; k < 2 * N - 1; k++) {
    uint32_t xs[N], ys[N], ms[N], ps[N];
    KZG_UNROLL_FULL
    for (int j = k - N + 1; j < N; j++) {
      xs[j - (k - N + 1)] = a.v[j];
      ys[j - (k - N + 1)] = b.v[k - j];
      ms[j - (k - N + 1)] = m[j];
      ps[j - (k - N + 1)] = F::mod(k - j);
    }
    mac_list_dyn<N, false>(A, xs, ys, 2 * N - 1 - k);
    mac_list_dyn<N, true>(A, ms, ps, 2 * N - 1 - k);
    t[k - N] = (uint32_t)A.lo;
    acc_shift(A);
  }
  t[N - 1] = (uint32_t)A.lo;
  uint32_t carry = (uint32_t)(A.lo >> 32);
  bn<N> tt;
  KZG_UNROLL_FULL
  for (int i = 0; i < N; i++) tt.v[i] = t[i];
  if (LAZY) {
    (void)carry;
    r = tt;
  } else {
    reduce_once<F>(r, tt, carry);
  }
}

template <class F>
KZG_HD void mont_mul(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  mont_mul_core<F, false>(r, a, b);
}

// ---- lazy-reduction variants: representatives in [0, 2m) -------------------------------------
// Valid because 4m < 2^(32N) for both fields (p < 2^381, r < 2^255):  a, b < 2m  =>
// (a*b + q*m) / 2^(32N) < m (4m / 2^(32N) + 1) < 2m, so the Montgomery product needs no final
// subtraction; add/sub fold back into [0, 2m) with one conditional +-2m.
template <class F>
KZG_HD void mont_mul_lazy(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  mont_mul_core<F, true>(r, a, b);
}
template <class F>
KZG_HD bn<F::N> modulus2() {
  bn<F::N> m;
  KZG_UNROLL_FULL
  for (int i = 0; i < F::N; i++) m.v[i] = F::mod2(i);
  return m;
}
template <class F>
KZG_HD void add_lazy(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  bn<F::N> t, s2;
  bn_add(t, a, b);  // < 4m: no carry out of the top limb
  uint32_t borrow = bn_sub(s2, t, modulus2<F>());
  KZG_UNROLL_FULL
  for (int i = 0; i < F::N; i++) r.v[i] = borrow ? t.v[i] : s2.v[i];
}
template <class F>
KZG_HD void sub_lazy(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  bn<F::N> t, u;
  uint32_t borrow = bn_sub(t, a, b);
  bn_add(u, t, modulus2<F>());
  KZG_UNROLL_FULL
  for (int i = 0; i < F::N; i++) r.v[i] = borrow ? u.v[i] : t.v[i];
}
// value in [0, 2m) congruent to 0 ?
template <class F>
KZG_HD bool is_zero_lazy(const bn<F::N>& a) {
  return bn_is_zero(a) || bn_eq(a, modulus<F>());
}
// [0, 2m) -> canonical [0, m)
template <class F>
KZG_HD void canonicalize(bn<F::N>& a) {
  reduce_once<F>(a, a, 0);
}

// Reference Montgomery multiplication in plain C (CIOS, 64-bit temporaries): the
// compiler schedules it and inserts every hazard wait state itself.  Used by the
// on-device self-test that cross-checks the inline-asm chains of mont_mul.
template <class F>
KZG_HD void mont_mul_plainc(bn<F::N>& r, const bn<F::N>& a, const bn<F::N>& b) {
  constexpr int N = F::N;
  uint32_t t[N + 2];
  KZG_UNROLL_FULL
  for (int i = 0; i < N + 2; i++) t[i] = 0;
  KZG_UNROLL_FULL
  for (int i = 0; i < N; i++) {
    uint64_t c = 0;
    KZG_UNROLL_FULL
    for (int j = 0; j < N; j++) {
      uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    uint64_t s = (uint64_t)t[N] + c;
    t[N] = (uint32_t)s;
    t[N + 1] = (uint32_t)(s >> 32);
    uint32_t m = t[0] * F::INV;
    s = (uint64_t)m * F::mod(0) + t[0];
    c = s >> 32;
    KZG_UNROLL_FULL
    for (int j = 1; j < N; j++) {
      s = (uint64_t)m * F::mod(j) + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = s >> 32;
    }
    s = (uint64_t)t[N] + c;
    t[N - 1] = (uint32_t)s;
    t[N] = t[N + 1] + (uint32_t)(s >> 32);
  }
  bn<N> tt;
  KZG_UNROLL_FULL
  for (int i = 0; i < N; i++) tt.v[i] = t[i];
  reduce_once<F>(r, tt, t[N]);
}

template <class F>
KZG_HD void mont_sqr(bn<F::N>& r, const bn<F::N>& a) {
  mont_mul<F>(r, a, a);
}

// to / from Montgomery form
template <class F>
KZG_HD void to_mont(bn<F::N>& r, const bn<F::N>& a) {
  bn<F::N> r2;
#pragma unroll
  for (int i = 0; i < F::N; i++) r2.v[i] = F::r2(i);
  mont_mul<F>(r, a, r2);
}

template <class F>
KZG_HD void from_mont(bn<F::N>& r, const bn<F::N>& a) {
  bn<F::N> one;
  bn_zero(one);
  one.v[0] = 1;
  mont_mul<F>(r, a, one);
}

// r = a^e (Montgomery domain), e given as N plain limbs, MSB-first square & multiply.
// The exponents used here are public constants (p-2, (p+1)/4, r-2), so the
// data-dependent branch is on public data only.
template <class F, class EXP>
KZG_HD void mont_pow_const_inl(bn<F::N>& r, const bn<F::N>& a, EXP expo) {
  bn<F::N> acc = mont_one<F>();
  bool started = false;
  for (int i = F::N * 32 - 1; i >= 0; i--) {
    if (started) mont_sqr<F>(acc, acc);
    if ((expo(i >> 5) >> (i & 31)) & 1u) {
      if (started)
        mont_mul<F>(acc, acc, a);
      else
        acc = a;
      started = true;
    }
  }
  r = acc;
}

template <class F, class EXP>
KZG_HD_NOINLINE void mont_pow_const(bn<F::N>& r, const bn<F::N>& a, EXP expo) {
  mont_pow_const_inl<F, EXP>(r, a, expo);
}

struct FpInvExp {
  KZG_HD uint32_t operator()(int i) const {
    const uint32_t t[12] = KZG_FP_MOD_MINUS_2;
    return t[i];
  }
};
struct FpSqrtExp {
  KZG_HD uint32_t operator()(int i) const {
    const uint32_t t[12] = KZG_FP_SQRT_EXP;
    return t[i];
  }
};
struct FrInvExp {
  KZG_HD uint32_t operator()(int i) const {
    const uint32_t t[8] = KZG_FR_MOD_MINUS_2;
    return t[i];
  }
};

// ---------------------------------------------------------------------------
// Fp / Fr front-ends
// ---------------------------------------------------------------------------
KZG_HD void fp_mul(fp_t& r, const fp_t& a, const fp_t& b) { mont_mul<FpParams>(r, a, b); }
KZG_HD void fp_sqr(fp_t& r, const fp_t& a) { mont_sqr<FpParams>(r, a); }
KZG_HD void fp_add(fp_t& r, const fp_t& a, const fp_t& b) { add_mod<FpParams>(r, a, b); }
KZG_HD void fp_sub(fp_t& r, const fp_t& a, const fp_t& b) { sub_mod<FpParams>(r, a, b); }
KZG_HD void fp_neg(fp_t& r, const fp_t& a) { neg_mod<FpParams>(r, a); }
KZG_HD void fp_dbl(fp_t& r, const fp_t& a) { dbl_mod<FpParams>(r, a); }
KZG_HD fp_t fp_one() { return mont_one<FpParams>(); }
KZG_HD void fp_inv_fermat(fp_t& r, const fp_t& a) { mont_pow_const<FpParams>(r, a, FpInvExp()); }  // a^(p-2); 0 -> 0.  fp_inv: modinv30.cuh
// candidate square root a^((p+1)/4); caller checks r*r == a
KZG_HD void fp_sqrt_candidate(fp_t& r, const fp_t& a) { mont_pow_const<FpParams>(r, a, FpSqrtExp()); }

KZG_HD void fr_mul(fr_t& r, const fr_t& a, const fr_t& b) { mont_mul<FrParams>(r, a, b); }
KZG_HD void fr_sqr(fr_t& r, const fr_t& a) { mont_sqr<FrParams>(r, a); }
KZG_HD void fr_add(fr_t& r, const fr_t& a, const fr_t& b) { add_mod<FrParams>(r, a, b); }
KZG_HD void fr_sub(fr_t& r, const fr_t& a, const fr_t& b) { sub_mod<FrParams>(r, a, b); }
KZG_HD void fr_neg(fr_t& r, const fr_t& a) { neg_mod<FrParams>(r, a); }
KZG_HD fr_t fr_one() { return mont_one<FrParams>(); }
KZG_HD void fr_inv_fermat(fr_t& r, const fr_t& a) { mont_pow_const<FrParams>(r, a, FrInvExp()); }  // a^(r-2); 0 -> 0.  fr_inv: modinv30.cuh

// ---------------------------------------------------------------------------
// byte <-> limb conversions (wire formats of src/bls.rs:130-159)
// ---------------------------------------------------------------------------
KZG_HD uint32_t load_be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
KZG_HD void store_be32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
}

// 32 big-endian bytes -> 8 plain limbs (little-endian limb order)
KZG_HD void fr_from_be_bytes_plain(fr_t& r, const uint8_t* p) {
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[7 - i] = load_be32(p + 4 * i);
}
KZG_HD void fr_to_be_bytes_plain(uint8_t* p, const fr_t& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) store_be32(p + 4 * i, a.v[7 - i]);
}
// blst_scalar_fr_check (src/bls.rs:113): canonical iff value < r
KZG_HD bool fr_is_canonical(const fr_t& plain) { return !bn_geq(plain, modulus<FrParams>()); }

// 256-bit big-endian digest reduced mod r (Fr::hash_to, src/bls.rs:189-205):
// 2^256 < 3r, so at most two subtractions.
KZG_HD void fr_reduce_256(fr_t& a) {
  fr_t t;
  if (bn_sub(t, a, modulus<FrParams>()) == 0) a = t;
  if (bn_sub(t, a, modulus<FrParams>()) == 0) a = t;
}

KZG_HD void fp_from_be_bytes_plain(fp_t& r, const uint8_t* p) {
#pragma unroll
  for (int i = 0; i < 12; i++) r.v[11 - i] = load_be32(p + 4 * i);
}
KZG_HD void fp_to_be_bytes_plain(uint8_t* p, const fp_t& a) {
#pragma unroll
  for (int i = 0; i < 12; i++) store_be32(p + 4 * i, a.v[11 - i]);
}

}  // namespace kzg

#include "modinv30.cuh"  // fp_inv / fr_inv (safegcd); needs everything above
