// Fixed-base 4096-point G1 MSM for gfx950 (kernel K2 of SURVEY.md section 2b):
// replaces P1::lincomb_pippenger(setup.g1_lagrange_brp, scalars)
// (src/bls.rs:416-437, called from src/blob.rs:48-53 and src/kzg/poly.rs:68).
//
// MI355X-first design.  kateth's bases never change after load
// (src/kzg/setup.rs:39), and one GPU owns 288 GB of HBM3E, so the windowed
// method is taken to its fixed-base limit: for every window j and base i the
// table holds all signed-digit multiples  d * 2^(c*j) * L_i , d = 1..2^(c-1),
// as affine Montgomery points (96 B each).  A commitment is then
//     C = sum_{i,j} sign(d_ij) * T[j][i][|d_ij|]
// i.e. exactly ceil(256/c) * 4096 complete mixed additions per blob, with no
// bucket pass, no bucket reduction, no sorting and perfect lane balance
// (blst's c = 10 Pippenger does ~133 k additions per blob; c = 14 here does
// 77.8 k).  The price is one 96-byte HBM gather per addition -- this is what
// turns the MSM into HBM gather + integer ALU work.
//
// Work decomposition: one wave (64 lanes) owns a (blob, split) unit; lane l
// walks points  split*P + k*64 + l  (scalar loads are 32 B per lane at
// consecutive addresses -> coalesced), recodes each scalar into signed base-2^c
// digits in registers, prefetches the next table entry while the current
// mixed add runs, and keeps its partial sum in 48 VGPRs (XYZZ).  The 64 lane
// sums go to HBM; k_msm_reduce combines them by a 6-level tree through LDS and
// k_g1_compress emits the 48-byte encodings.
#pragma once
#include "g1_decode28.cuh"

namespace kzg {

constexpr uint64_t KZG_BYTES_PER_BLOB_ = 131072;

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

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

__device__ __forceinline__ void load_affine96(fp_t& x, fp_t& y, const uint4* __restrict__ tbl, uint64_t idx) {
  const uint4* p = tbl + idx * 6;
  uint4 a0 = p[0], a1 = p[1], a2 = p[2], b0 = p[3], b1 = p[4], b2 = p[5];
  x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w;
  x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
  x.v[8] = a2.x; x.v[9] = a2.y; x.v[10] = a2.z; x.v[11] = a2.w;
  y.v[0] = b0.x; y.v[1] = b0.y; y.v[2] = b0.z; y.v[3] = b0.w;
  y.v[4] = b1.x; y.v[5] = b1.y; y.v[6] = b1.z; y.v[7] = b1.w;
  y.v[8] = b2.x; y.v[9] = b2.y; y.v[10] = b2.z; y.v[11] = b2.w;
}

__device__ __forceinline__ void store_affine96(uint4* tbl, uint64_t idx, const fp_t& x, const fp_t& y) {
  uint4* p = tbl + idx * 6;
  p[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
  p[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
  p[2] = make_uint4(x.v[8], x.v[9], x.v[10], x.v[11]);
  p[3] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
  p[4] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
  p[5] = make_uint4(y.v[8], y.v[9], y.v[10], y.v[11]);
}

// 32-byte scalar at `p` -> 8 plain little-endian limbs
template <bool BE_BYTES>
__device__ __forceinline__ void load_scalar(uint32_t* sc, const uint8_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 w0 = q[0], w1 = q[1];
  if (BE_BYTES) {
    sc[7] = bswap32(w0.x); sc[6] = bswap32(w0.y); sc[5] = bswap32(w0.z); sc[4] = bswap32(w0.w);
    sc[3] = bswap32(w1.x); sc[2] = bswap32(w1.y); sc[1] = bswap32(w1.z); sc[0] = bswap32(w1.w);
  } else {
    sc[0] = w0.x; sc[1] = w0.y; sc[2] = w0.z; sc[3] = w0.w;
    sc[4] = w1.x; sc[5] = w1.y; sc[6] = w1.z; sc[7] = w1.w;
  }
}

// Sum the 64 lane accumulators of one wave; result valid in lane 0.
// lds: 32 slots of g1_xyzz owned by this wave.
__device__ __forceinline__ void wave_reduce_xyzz(g1_xyzz& acc, g1_xyzz* lds, int lane) {
#pragma unroll 1
  for (int step = 1; step < 64; step <<= 1) {
    const int m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = acc;
    __syncthreads();
    if ((lane & m) == 0) {
      g1_xyzz other = lds[(lane + step) >> 1];
      g1_xyzz mine = acc;  // copy: keeps the caller's accumulator out of scratch
      xyzz_add(mine, other);
      acc = mine;
    }
    __syncthreads();
  }
}

#if defined(KZG_TEST_WINDOW_MSM)
// TEST-ONLY build (tests/window_msm, -DKZG_TEST_WINDOW_MSM): round 1's window-table MSM kernels (12 x 32-bit limbs, and
// the radix-2^28 one below), kept as independent cross-checks of the comb kernel (msm_comb.cuh); they are not compiled
// into the product library.
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

#endif  // KZG_TEST_WINDOW_MSM

// One wave per unit: sums the lane partials of each group of `lpb` lanes (64: one sum per unit; 32: the comb's half-wave
// mode, two blobs per unit) by a tree through LDS; unit_sums[u * (64 / lpb) + lane / lpb].
static __global__ __launch_bounds__(64) void k_msm_reduce(const g1_xyzz* __restrict__ partials, uint64_t units, uint32_t lpb, g1_xyzz* __restrict__ unit_sums) {
  __shared__ g1_xyzz28 lds[32];
  const int lane = threadIdx.x;
  const uint64_t u = blockIdx.x;
  if (u >= units) return;
  // the tree runs in the radix-2^28 field (a full XYZZ addition is ~7.5 k instead of ~11 k VALU instructions)
  g1_xyzz28 acc;
  {
    const g1_xyzz in = partials[u * 64 + lane];
    xyzz28_from_xyzz(acc, in);
  }
#pragma unroll 1
  for (int step = 1; step < (int)lpb; step <<= 1) {
    const int m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = acc;
    __syncthreads();
    if ((lane & m) == 0) {
      g1_xyzz28 other = lds[(lane + step) >> 1];
      g1_xyzz28 mine = acc;  // copies: the out-of-line adder takes addresses
      xyzz28_add_complete(mine, other);
      acc = mine;
    }
    __syncthreads();
  }
  if ((lane & (int)(lpb - 1)) == 0) {
    g1_xyzz out;
    xyzz28_to_xyzz(out, acc);
    unit_sums[u * (64u / lpb) + (uint32_t)lane / lpb] = out;
  }
}
// One wave per blob: sums the blob's `splits` (<= 64) unit sums.
static __global__ __launch_bounds__(64) void k_msm_reduce_splits(const g1_xyzz* __restrict__ unit_sums, uint32_t splits, uint64_t n,
                                                                 g1_xyzz* __restrict__ sums) {
  __shared__ g1_xyzz lds[32];
  const int lane = threadIdx.x;
  const uint64_t b = blockIdx.x;
  if (b >= n) return;
  g1_xyzz acc;
  xyzz_set_inf(acc);
  if ((uint32_t)lane < splits) acc = unit_sums[b * splits + lane];
  wave_reduce_xyzz(acc, lds, lane);
  if (lane == 0) sums[b] = acc;
}

// One thread per item: XYZZ -> affine -> 48-byte compressed encoding
// (K3: blst_p1_compress, src/bls.rs:499) and/or the 96-byte blst_p1_affine image (so that a caller that wants the
// reference's `P1` back -- Commitment = Proof = P1, src/kzg/mod.rs:9-10 -- needs no square root).  Items whose status
// is non-zero get zero bytes.  Either output pointer may be null.
// `comb_k` (nullable): the comb MSM's constant term K (affine, 2^384-Montgomery), added to every sum first (msm_comb.cuh).
static __global__ __launch_bounds__(64) void k_g1_compress(const g1_xyzz* __restrict__ sums, uint64_t n, const int32_t* __restrict__ status,
                                                    uint8_t* __restrict__ out48, uint8_t* __restrict__ out_affine96, const uint4* __restrict__ comb_k) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  uint8_t tmp[48];
  uint32_t aff[24];
  if (status != nullptr && status[b] != 0) {
    for (int q = 0; q < 48; q++) tmp[q] = 0;
    for (int q = 0; q < 24; q++) aff[q] = 0;
  } else {
    g1_xyzz acc = sums[b];
    if (comb_k != nullptr) {
      fp_t kx, ky;
      load_affine96(kx, ky, comb_k, 0);
      g1_xyzz mine = acc;  // copy: keeps the complete adder's operands addressable
      xyzz_madd(mine, kx, ky);
      acc = mine;
    }
    g1_compress_xyzz28(tmp, out_affine96 ? aff : nullptr, acc);  // inversion in the radix-2^28 field (g1_decode28.cuh)
  }
  if (out48) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out48 + b * 48);
    for (int q = 0; q < 12; q++)
      o[q] = (uint32_t)tmp[4 * q] | ((uint32_t)tmp[4 * q + 1] << 8) | ((uint32_t)tmp[4 * q + 2] << 16) | ((uint32_t)tmp[4 * q + 3] << 24);
  }
  if (out_affine96) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out_affine96 + b * 96);
    for (int q = 0; q < 24; q++) o[q] = aff[q];
  }
}

#endif  // __HIPCC__
}  // namespace kzg
