// Context-creation kernels: what Setup::load_json does after parsing
// (src/kzg/setup.rs:52-81) plus the batch normaliser of the fixed-base table build (msm_comb.cuh).
#pragma once
#include "fp30.cuh"
#include "fr29.cuh"
#include "msm_fixed.cuh"

namespace kzg {
#if defined(__HIPCC__)

__device__ __forceinline__ uint32_t bitrev12(uint32_t i) { return __builtin_bitreverse32(i) >> 20; }

// thread t: decompress + subgroup-check g1_lagrange[t] (file order), store the
// affine Montgomery point at the bit-reversed index (src/kzg/setup.rs:59-65,
// src/math.rs:72-74).  status[t] = KZG_ERR_* or 0; infinity -> flagged as 100.
static __global__ __launch_bounds__(64) void k_setup_g1(const uint8_t* __restrict__ in48, uint4* __restrict__ bases_brp, int32_t* __restrict__ status) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4096) return;
  fp_t x, y;
  bool inf;
  int32_t st = g1_decompress(x, y, inf, in48 + 48 * t);
  if (st == 0 && inf) st = 100;
  status[t] = st;
  if (st == 0) store_affine96(bases_brp, bitrev12(t), x, y);
}

// thread: normalises KN consecutive XYZZ entries with one shared inversion (Montgomery's trick on zz*zzz) and writes affine
// table entries; fmt selects the stored form of the coordinates (TABLE_FMT_*: packed centred 30-bit digits of x * 2^390 for the
// comb kernel, x * 2^392 for the radix-2^28 test kernels, x * 2^384 otherwise).  An entry at infinity (a vanishing subset sum: impossible for the ceremony, possible for a degenerate setup
// such as repeated points) has no affine form: it is kept out of the shared inversion, stored as zeros and reported through
// *inf_seen, and the caller rejects the setup (P1::lincomb, src/bls.rs:406-437, would accept it -- DESIGN.md section 2).
constexpr int TABLE_FMT_R384 = 0, TABLE_FMT_R392 = 1, TABLE_FMT_PACKED30 = 2;
template <int KN>
static __global__ __launch_bounds__(64) void k_table_normalize(const g1_xyzz* __restrict__ tmp, uint64_t count, uint4* __restrict__ table, uint64_t table_off,
                                                              int fmt, uint32_t* __restrict__ inf_seen) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t first = t * KN;
  if (first >= count) return;
  fp_t w[KN];   // w[k] = zz_k * zzz_k
  fp_t pre[KN]; // prefix products
  fp_t total = fp_one();
  int m = 0;
#pragma unroll
  for (int k = 0; k < KN; k++) {
    if (first + k < count) {
      const g1_xyzz& e = tmp[first + k];
      fp_mul(w[k], e.zz, e.zzz);
      if (bn_is_zero(e.zz)) {
        w[k] = fp_one();
        atomicOr(inf_seen, 1u);
      }
      if (k == 0)
        pre[k] = w[k];
      else
        fp_mul(pre[k], pre[k - 1], w[k]);
      total = pre[k];
      m = k + 1;
    }
  }
  fp_t inv;
  fp_inv(inv, total);  // safegcd (modinv30.cuh): one inversion per KN entries
#pragma unroll
  for (int k = KN - 1; k >= 0; k--) {
    if (k < m) {
      fp_t wi;  // 1 / w[k]
      if (k == 0)
        wi = inv;
      else {
        fp_mul(wi, inv, pre[k - 1]);
        fp_mul(inv, inv, w[k]);
      }
      const g1_xyzz& e = tmp[first + k];
      fp_t a, x, y;
      fp_mul(a, wi, e.zzz);  // 1/zz
      fp_mul(x, e.x, a);
      fp_mul(a, wi, e.zz);  // 1/zzz
      fp_mul(y, e.y, a);  // an entry at infinity has x = y = 0 and stays (0, 0)
      if (fmt == TABLE_FMT_R392) {  // operands of the radix-2^28 adder (test-only window kernels): coordinates times 2^392 instead of 2^384
        fp_to_r392(x, x);
        fp_to_r392(y, y);
      } else if (fmt == TABLE_FMT_PACKED30) {  // the comb table (fp30.cuh): x * 2^390 as thirteen centred 30-bit digits packed into 48 bytes
        fp_t px, py;
        fp_to_packed30(px.v, x);
        fp_to_packed30(py.v, y);
        x = px;
        y = py;
      }
      store_affine96(table, table_off + first + k, x, y);
    }
  }
}

// One wave: S = sum of the 4096 setup points (any order), affine 2^384-Montgomery, *inf = 1 if the sum is the identity.
// The comb's constant term is [c0] S (msm_comb.cuh); for a Lagrange-basis setup S is the generator, but nothing in
// Setup::load_json (src/kzg/setup.rs:46-82) requires that, so the engine does not assume it.
static __global__ __launch_bounds__(64) void k_setup_sum_bases(const uint4* __restrict__ bases, uint4* __restrict__ sum_affine, uint32_t* __restrict__ inf) {
  __shared__ g1_xyzz lds[32];
  const int lane = threadIdx.x;
  g1_xyzz acc;
  xyzz_set_inf(acc);
#pragma unroll 1
  for (uint32_t k = 0; k < 64; k++) {
    fp_t x, y;
    load_affine96(x, y, bases, k * 64u + (uint32_t)lane);
    g1_xyzz mine = acc;
    xyzz_madd(mine, x, y);
    acc = mine;
  }
  wave_reduce_xyzz(acc, lds, lane);
  if (lane == 0) {
    fp_t x, y;
    bn_zero(x);
    bn_zero(y);
    *inf = xyzz_to_affine(x, y, acc) ? 0u : 1u;
    store_affine96(sum_affine, 0, x, y);
  }
}

// one thread per hex: builds eval_tab from the Montgomery (radix 2^256) roots.  In bit-reversed order elements 16h + 2k, 2k + 1
// sit at +-w rho_k with w = roots_brp[16h] and rho = 1, i, c, ic, s, is, cs, ics (i = roots_brp[2], c = roots_brp[4],
// s = roots_brp[8]: the primitive 4th, 8th and 16th roots of unity).  Slots as listed in fr29.cuh.
static __global__ __launch_bounds__(64) void k_setup_eval_tab(const fr_t* __restrict__ roots_brp, uint32_t* __restrict__ eval_tab) {
  const uint32_t hd = blockIdx.x * blockDim.x + threadIdx.x;
  if (hd >= (uint32_t)EVAL_TAB_HEXES) return;
  const fr_t w = roots_brp[16 * hd], ri = roots_brp[2], rc = roots_brp[4], rs = roots_brp[8];
  fr_t c261, c522, t, u, sq;
  {
    const uint32_t a[8] = KZG_FR_R261_PLAIN, b[8] = KZG_FR_R522_PLAIN;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      c261.v[q] = a[q];
      c522.v[q] = b[q];
    }
  }
  uint32_t* out = eval_tab + (uint64_t)hd * EVAL_TAB_DWORDS;
  auto put = [&](int slot, const fr_t& v) {  // Montgomery (2^256) value -> (value 2^261) canonical -> 9 strictly normalised limbs
    fr_t m;
    fr_mul(m, v, c261);  // (v 2^256)(2^261) / 2^256 = v 2^261
    fr29 o;
    f29_from_bn(o, m);
#pragma unroll
    for (int q = 0; q < F29_N; q++) out[EVAL_TAB_SLOT * slot + q] = o.l[q];
#pragma unroll
    for (int q = F29_N; q < EVAL_TAB_SLOT; q++) out[EVAL_TAB_SLOT * slot + q] = 0;
  };
  for (int k = 0; k < 8; k++) put(k, roots_brp[16 * hd + 2 * k]);
  fr_sqr(sq, w);       // w^2
  put(8, sq);
  fr_mul(t, sq, ri);   // i w^2
  put(9, t);
  fr_mul(u, sq, rc);   // c w^2
  put(10, u);
  fr_mul(u, t, rc);    // i c w^2
  put(11, u);
  fr_sqr(sq, sq);      // w^4
  put(12, sq);
  fr_mul(t, sq, rc);   // c w^4
  put(13, t);
  fr_mul(u, sq, ri);   // i w^4
  put(14, u);
  fr_mul(u, t, ri);    // c i w^4
  put(15, u);
  fr_sqr(sq, sq);      // w^8
  put(16, sq);
  fr_mul(t, sq, rs);   // s w^8
  put(17, t);
  fr_sqr(sq, sq);      // w^16
  put(18, sq);
  {  // w R^2
    fr_t m;
    fr_mul(m, w, c522);
    fr29 o;
    f29_from_bn(o, m);
#pragma unroll
    for (int q = 0; q < F29_N; q++) out[EVAL_TAB_SLOT * 19 + q] = o.l[q];
#pragma unroll
    for (int q = F29_N; q < EVAL_TAB_SLOT; q++) out[EVAL_TAB_SLOT * 19 + q] = 0;
  }
}

// roots_of_unity_brp (src/math.rs:16-29 + BRP, src/kzg/setup.rs:74-75), Montgomery form.
// thread t computes omega^t by square-and-multiply and stores at bitrev12(t).
static __global__ __launch_bounds__(64) void k_setup_roots(fr_t* __restrict__ roots_brp) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4096) return;
  fr_t w;
  {
    const uint32_t om[8] = KZG_FR_OMEGA4096_MONT;
#pragma unroll
    for (int q = 0; q < 8; q++) w.v[q] = om[q];
  }
  fr_t acc = fr_one();
  for (int bit = 11; bit >= 0; bit--) {
    fr_sqr(acc, acc);
    if ((t >> bit) & 1u) fr_mul(acc, acc, w);
  }
  roots_brp[bitrev12(t)] = acc;
}

#endif
}  // namespace kzg
