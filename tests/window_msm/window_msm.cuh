// TEST-ONLY (never part of kateth_amd/libkateth_amd.so): round 1's window-table fixed-base MSM, kept as two independent
// cross-checks of the product's subset-sum comb (kateth_amd/csrc/msm_comb.cuh).  Same sum -- P1::lincomb_pippenger(
// setup.g1_lagrange_brp, scalars), src/bls.rs:416-437 -- by a different table, a different recoding (signed c-bit windows
// instead of signed bits) and, for k_msm_fixed, a different field representation (12 x 32-bit limbs instead of radix 2^28).
//
// For every window j and base i the table holds all signed-digit multiples d * 2^(c*j) * L_i, d = 1..2^(c-1), as affine
// Montgomery points (96 B each); a commitment is ceil(256/c) * 4096 mixed additions, one gather each.  One wave owns a
// (blob, split) unit; lane l walks points split*P + k*64 + l, recodes each scalar into signed base-2^c digits in registers and
// prefetches the next table entry while the current mixed add runs.
#pragma once
#include "../../kateth_amd/csrc/setup_kernels.cuh"

namespace kzg {

struct MsmGeom {
  uint32_t c;            // window bits
  uint32_t W;            // number of windows = ceil(256 / c)
  uint32_t half;         // 2^(c-1) = entries per (window, base) except the top window
  uint32_t top_entries;  // entries per base in the top window (largest possible top digit)
};

KZG_HD uint64_t table_index(const MsmGeom& g, uint32_t j, uint32_t i, uint32_t d /*1-based*/) {
  if (j + 1 < g.W) return ((uint64_t)j * 4096u + i) * g.half + (d - 1);
  return (uint64_t)(g.W - 1) * 4096u * g.half + (uint64_t)i * g.top_entries + (d - 1);
}
KZG_HD uint64_t table_entries(const MsmGeom& g) {
  return (uint64_t)(g.W - 1) * 4096u * g.half + (uint64_t)4096u * g.top_entries;
}

#if defined(__HIPCC__)

// One wave per (blob, split).  BE_BYTES: scalars are raw blob bytes (32-B
// big-endian, validated here: Blob::from_slice, src/blob.rs:26-37); otherwise
// canonical little-endian limbs produced on device (quotient polynomial).
template <bool BE_BYTES, int OCC>
static __global__ __launch_bounds__(64, OCC) void k_msm_fixed(const uint8_t* __restrict__ scalars, uint32_t splits,
                                                  const uint4* __restrict__ table, MsmGeom g,
                                                  g1_xyzz* __restrict__ partials, int32_t* __restrict__ status) {
  const int lane = threadIdx.x;
  const uint64_t unit = blockIdx.x;
  const uint64_t blob = unit / splits;
  const uint32_t split = (uint32_t)(unit % splits);
  const uint32_t pts_per_split = 4096u / splits;
  const uint32_t per_lane = pts_per_split / 64u;
  const uint32_t mask = (1u << g.c) - 1u;
  const uint8_t* base = scalars + blob * (uint64_t)KZG_BYTES_PER_BLOB_;

  g1_xyzz acc;
  xyzz_set_inf(acc);
  bool bad = false;

  // walker state
  uint32_t sc[8];
  uint32_t carry = 0, j = g.W, k = 0, i = 0;
  // pipeline slot
  fp_t nx, ny;
  bn_zero(nx);
  bn_zero(ny);
  bool nvalid = false, nneg = false;
  const uint32_t total = per_lane * g.W;

#pragma unroll 1
  for (uint32_t t = 0; t <= total; t++) {
    fp_t cx = nx, cy = ny;
    const bool cvalid = nvalid, cneg = nneg;
    nvalid = false;
    if (t < total) {
      if (j == g.W) {  // next scalar
        i = split * pts_per_split + k * 64u + (uint32_t)lane;
        load_scalar<BE_BYTES>(sc, base + (uint64_t)i * 32u);
        if (BE_BYTES) {
          fr_t v;
#pragma unroll
          for (int q = 0; q < 8; q++) v.v[q] = sc[q];
          if (!fr_is_canonical(v)) {
            bad = true;
#pragma unroll
            for (int q = 0; q < 8; q++) sc[q] = 0;
          }
        }
        carry = 0;
        j = 0;
        k++;
      }
      uint32_t u = (sc[0] & mask) + carry;
#pragma unroll
      for (int q = 0; q < 7; q++) sc[q] = (sc[q] >> g.c) | (sc[q + 1] << (32u - g.c));
      sc[7] >>= g.c;
      const bool neg = u > g.half;
      const uint32_t d = neg ? ((1u << g.c) - u) : u;
      carry = neg ? 1u : 0u;
      if (d != 0) {
        load_affine96(nx, ny, table, table_index(g, j, i, d));
        nvalid = true;
        nneg = neg;
      }
      j++;
    }
    if (cvalid) {
      if (cneg) fp_neg(cy, cy);
      xyzz_madd_lazy(acc, cx, cy);  // accumulator coordinates stay in [0, 2p) inside the loop
    }
  }
  xyzz_canonicalize(acc);

  // lane sums go to HBM (12 KB per wave); the cross-lane tree and the encoding run in
  // k_msm_reduce / k_g1_compress so that this kernel has no calls and no LDS
  partials[unit * 64 + lane] = acc;
  if (BE_BYTES) {
    if (__any(bad) && lane == 0) atomicOr(&status[blob], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
}


// The fixed-base walk with the accumulator in the carry-free radix-2^28 representation (fp28.cuh): 392 v_mad_u64_u32
// and no carry instruction per Montgomery product, 9 reductions per mixed add.  The table must hold 2^392-Montgomery
// coordinates (kzg_ctx::msm_radix28).  The generic add runs inline; the first add of a lane (identity accumulator)
// and the ~2^-17 of adds whose cheap "P == +-Q?" test fires go through the out-of-line complete adder on a COPY of
// the accumulator (taking the accumulator's own address would move it to scratch for the whole loop) and re-read the
// table entry, so the hot path keeps neither the raw entry nor the doubling's operands alive.
template <bool BE_BYTES>
static __global__ __launch_bounds__(64, 2) void k_msm_fixed28(const uint8_t* __restrict__ scalars, uint32_t splits,
                                                              const uint4* __restrict__ table, MsmGeom g,
                                                              g1_xyzz* __restrict__ partials, int32_t* __restrict__ status) {
  const int lane = threadIdx.x;
  const uint64_t unit = blockIdx.x;
  const uint64_t blob = unit / splits;
  const uint32_t split = (uint32_t)(unit % splits);
  const uint32_t pts_per_split = 4096u / splits;
  const uint32_t per_lane = pts_per_split / 64u;
  const uint32_t mask = (1u << g.c) - 1u;
  const uint8_t* base = scalars + blob * (uint64_t)KZG_BYTES_PER_BLOB_;

  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  bool bad = false;

  uint32_t sc[8];
  uint32_t carry = 0, j = g.W, k = 0, i = 0;
  fp_t nx, ny;
  bn_zero(nx);
  bn_zero(ny);
  bool nvalid = false, nneg = false;
  uint64_t nidx = 0;
  const uint32_t total = per_lane * g.W;

#pragma unroll 1
  for (uint32_t t = 0; t <= total; t++) {
    fp28 cx, cy;
    f28_load_entry(cx, cy, nx, ny, nneg);
    const bool cvalid = nvalid, cneg = nneg;
    const uint64_t cidx = nidx;
    nvalid = false;
    if (t < total) {
      if (j == g.W) {  // next scalar
        i = split * pts_per_split + k * 64u + (uint32_t)lane;
        load_scalar<BE_BYTES>(sc, base + (uint64_t)i * 32u);
        if (BE_BYTES) {
          fr_t v;
#pragma unroll
          for (int q = 0; q < 8; q++) v.v[q] = sc[q];
          if (!fr_is_canonical(v)) {
            bad = true;
#pragma unroll
            for (int q = 0; q < 8; q++) sc[q] = 0;
          }
        }
        carry = 0;
        j = 0;
        k++;
      }
      uint32_t u = (sc[0] & mask) + carry;
#pragma unroll
      for (int q = 0; q < 7; q++) sc[q] = (sc[q] >> g.c) | (sc[q + 1] << (32u - g.c));
      sc[7] >>= g.c;
      const bool neg = u > g.half;
      const uint32_t d = neg ? ((1u << g.c) - u) : u;
      carry = neg ? 1u : 0u;
      if (d != 0) {
        nidx = table_index(g, j, i, d);
        load_affine96(nx, ny, table, nidx);
        nvalid = true;
        nneg = neg;
      }
      j++;
    }
    if (cvalid) {
      bool done = false;
      if (!acc.inf) done = xyzz28_madd_fast(acc, cx, cy);
      if (!done) {
        g1_xyzz28 tmp = acc;
        fp_t rx, ry;
        load_affine96(rx, ry, table, cidx);
        fp28 sx, sy;  // separate objects: the call takes their address
        f28_load_entry(sx, sy, rx, ry, cneg);
        xyzz28_madd_complete(tmp, sx, sy);
        acc = tmp;
      }
    }
  }
  g1_xyzz out;
  xyzz28_to_xyzz(out, acc);  // back to canonical 2^384-Montgomery limbs for k_msm_reduce
  partials[unit * 64 + lane] = out;
  if (BE_BYTES) {
    if (__any(bad) && lane == 0) atomicOr(&status[blob], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
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

#endif  // __HIPCC__
}  // namespace kzg
