// Context-creation kernels: what Setup::load_json does after parsing
// (src/kzg/setup.rs:52-81) plus the fixed-base table build of msm_fixed.cuh.
#pragma once
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

// thread i: window bases Q[j][i] = 2^(c*j) * L_i for j = 0..W-1 (affine).
static __global__ __launch_bounds__(64) void k_table_window_bases(const uint4* __restrict__ bases_brp, uint4* __restrict__ win_bases, MsmGeom g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4096) return;
  fp_t x, y;
  load_affine96(x, y, bases_brp, i);
  store_affine96(win_bases, i, x, y);
  g1_xyzz acc;
  xyzz_from_affine(acc, x, y);
  for (uint32_t j = 1; j < g.W; j++) {
    for (uint32_t q = 0; q < g.c; q++) xyzz_dbl(acc);
    xyzz_to_affine(x, y, acc);
    store_affine96(win_bases, (uint64_t)j * 4096u + i, x, y);
    xyzz_from_affine(acc, x, y);
  }
}

// thread (i, s) of window j: the chain d*Q for the s-th of `segs` slices of d = 1..entries, XYZZ results to tmp
// (tmp index = i*entries + d-1).  A slice starts from [first]Q by double-and-add (15 steps at most) and then adds Q once
// per entry; with one thread per base the 4,096 chains of 32,768 sequential additions were pure latency (0.44 s per
// window at c = 16), sliced 32 ways they fill the chip.
static __global__ __launch_bounds__(64) void k_table_chain(const uint4* __restrict__ win_bases, uint32_t j, uint32_t entries, uint32_t segs,
                                                           g1_xyzz* __restrict__ tmp) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4096u * segs) return;
  const uint32_t i = t / segs, sg = t % segs;
  const uint32_t len = (entries + segs - 1) / segs;
  const uint32_t first = sg * len + 1;  // d of this slice's first entry
  if (first > entries) return;
  const uint32_t last = (first + len - 1 < entries) ? first + len - 1 : entries;
  fp_t x, y;
  load_affine96(x, y, win_bases, (uint64_t)j * 4096u + i);
  g1_xyzz acc;
  xyzz_from_affine(acc, x, y);
  if (first > 1) {  // acc = [first]Q, MSB-first
    const int top = 31 - __builtin_clz(first);
    for (int bit = top - 1; bit >= 0; bit--) {
      xyzz_dbl(acc);
      if ((first >> bit) & 1u) {
        g1_xyzz mine = acc;
        xyzz_madd(mine, x, y);
        acc = mine;
      }
    }
  }
  g1_xyzz* o = tmp + (uint64_t)i * entries;
  o[first - 1] = acc;
#pragma unroll 1
  for (uint32_t d = first; d < last; d++) {
    xyzz_madd(acc, x, y);
    o[d] = acc;
  }
}

// thread: normalises KN consecutive XYZZ entries with one shared inversion; r392 selects the Montgomery radix of the stored
// coordinates (2^392 for the radix-2^28 MSM kernel, 2^384 otherwise)
//
// (Montgomery's trick on zz*zzz) and writes affine table entries.
template <int KN>
static __global__ __launch_bounds__(64) void k_table_normalize(const g1_xyzz* __restrict__ tmp, uint64_t count, uint4* __restrict__ table, uint64_t table_off,
                                                              bool r392) {
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
      fp_mul(y, e.y, a);
      if (r392) {  // table for k_msm_fixed28: coordinates times 2^392 instead of 2^384
        fp_to_r392(x, x);
        fp_to_r392(y, y);
      }
      store_affine96(table, table_off + first + k, x, y);
    }
  }
}

// one thread per pair: builds eval_tab from the Montgomery (radix 2^256) roots
static __global__ __launch_bounds__(64) void k_setup_eval_tab(const fr_t* __restrict__ roots_brp, uint32_t* __restrict__ eval_tab) {
  const uint32_t pr = blockIdx.x * blockDim.x + threadIdx.x;
  if (pr >= 2048) return;
  const fr_t w = roots_brp[2 * pr];
  fr_t c261, c522, t;
  {
    const uint32_t a[8] = KZG_FR_R261_PLAIN, b[8] = KZG_FR_R522_PLAIN;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      c261.v[q] = a[q];
      c522.v[q] = b[q];
    }
  }
  fr29 o;
  uint32_t* out = eval_tab + (uint64_t)pr * EVAL_TAB_DWORDS;
  fr_mul(t, w, c261);  // (w 2^256)(2^261) / 2^256 = w 2^261
  f29_from_bn(o, t);
#pragma unroll
  for (int q = 0; q < F29_N; q++) out[q] = o.l[q];
  fr_mul(t, w, c522);
  f29_from_bn(o, t);
#pragma unroll
  for (int q = 0; q < F29_N; q++) out[9 + q] = o.l[q];
  fr_sqr(t, w);
  fr_mul(t, t, c261);
  f29_from_bn(o, t);
#pragma unroll
  for (int q = 0; q < F29_N; q++) out[18 + q] = o.l[q];
  out[27] = 0;
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
