// Pieces shared by the fixed-base 4096-point G1 MSM (msm_comb.cuh; replaces P1::lincomb_pippenger(setup.g1_lagrange_brp,
// scalars), src/bls.rs:416-437, called from src/blob.rs:48-53 and src/kzg/poly.rs:68): 96-byte affine table entries,
// scalar loads, the lane-sum trees (k_msm_reduce, k_msm_reduce_splits) and the encoder (k_g1_compress: blst_p1_compress,
// src/bls.rs:499).  Round 1's window-table kernels, which this file used to hold, live on as independent cross-checks in the
// test-only library (tests/window_msm/).
#pragma once
#include "g1_decode28.cuh"
#include "issue_fair.cuh"

namespace kzg {

constexpr uint64_t KZG_BYTES_PER_BLOB_ = 131072;

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


// One wave per unit: sums the lane partials of each group of `lpb` lanes (64: one sum per unit; 32: the comb's half-wave
// mode, two blobs per unit) by a tree through LDS; unit_sums[u * (64 / lpb) + lane / lpb].
// n_out = number of sums to store (half-wave mode with an odd batch: the idle half of the last unit stores nothing).
static __global__ __launch_bounds__(64) void k_msm_reduce(const g1_xyzz* __restrict__ partials, uint64_t units, uint32_t lpb, g1_xyzz* __restrict__ unit_sums,
                                                          uint64_t n_out) {
  __shared__ g1_xyzz28 lds[32];
  issue_priority_latency();
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
      const g1_xyzz28 other = lds[(lane + step) >> 1];
      xyzz28_add_complete_inl(acc, other);
    }
    __syncthreads();
  }
  const uint64_t slot = u * (64u / lpb) + (uint32_t)lane / lpb;
  if ((lane & (int)(lpb - 1)) == 0 && slot < n_out) {
    g1_xyzz out;
    xyzz28_to_xyzz(out, acc);
    unit_sums[slot] = out;
  }
}
// The comb's half-wave mode (two blobs per unit, 32 lane sums each) with FOUR blobs per wave: lane l adds the lane sums j and
// j + 16 (j = l % 16) of blob l / 16 as it loads them, then a 4-level tree over 16 lanes.  The same five dependent additions per
// blob as k_msm_reduce, but n / 4 waves instead of n / 2: at 4,096 blobs one wave per SIMD instead of two, whose interleaved
// issue made every level 1.5 times as long (0.25 -> 0.13 ms per 4,096 blobs).
static __global__ __launch_bounds__(64) void k_msm_reduce_half4(const g1_xyzz* __restrict__ partials, uint64_t units, g1_xyzz* __restrict__ unit_sums,
                                                                uint64_t n_out) {
  __shared__ g1_xyzz28 lds[32];
  issue_priority_latency();
  const int lane = threadIdx.x;
  const int q = lane >> 4, j = lane & 15;
  const uint64_t u = (uint64_t)blockIdx.x * 2 + (uint32_t)(q >> 1);  // unit of this lane's blob
  const uint64_t slot = u * 2 + (uint32_t)(q & 1);                   // = blob index
  const bool live = u < units && slot < n_out;
  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  if (live) {
    const g1_xyzz* row = partials + u * 64 + (uint32_t)(q & 1) * 32u;
    const g1_xyzz a = row[j], b = row[j + 16];
    g1_xyzz28 other;
    xyzz28_from_xyzz(acc, a);
    xyzz28_from_xyzz(other, b);
    xyzz28_add_complete_inl(acc, other);
  }
#pragma unroll 1
  for (int step = 1; step < 16; step <<= 1) {
    const int m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = acc;
    __syncthreads();
    if ((lane & m) == 0) {
      const g1_xyzz28 other = lds[(lane + step) >> 1];
      xyzz28_add_complete_inl(acc, other);
    }
    __syncthreads();
  }
  if (j == 0 && live) {
    g1_xyzz out;
    xyzz28_to_xyzz(out, acc);
    unit_sums[slot] = out;
  }
}
// XYZZ sum of one blob -> affine -> 48-byte compressed encoding (K3: blst_p1_compress, src/bls.rs:499) and/or the 96-byte
// blst_p1_affine image (so that a caller that wants the reference's `P1` back -- Commitment = Proof = P1,
// src/kzg/mod.rs:9-10 -- needs no square root).  An item whose status is non-zero gets zero bytes.  Either output pointer may
// be null.  (The comb MSM's constant term K is already in the sum: one lane per blob starts from it, msm_comb.cuh.)
__device__ __noinline__ void g1_finish_item(const g1_xyzz& sum, uint64_t b, const int32_t* __restrict__ status, uint8_t* __restrict__ out48,
                                            uint8_t* __restrict__ out_affine96) {
  uint8_t tmp[48];
  uint32_t aff[24];
  if (status != nullptr && status[b] != 0) {
    for (int q = 0; q < 48; q++) tmp[q] = 0;
    for (int q = 0; q < 24; q++) aff[q] = 0;
  } else {
    g1_compress_xyzz28(tmp, out_affine96 ? aff : nullptr, sum);  // inversion in the radix-2^28 field (g1_decode28.cuh)
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

// One WAVE per blob: sums the blob's `splits` (<= 256) unit sums in the radix-2^28 field.  Each lane first adds up its own
// units t, t + 64, ... one after the other, then the 64 lane sums fold by a tree through LDS.  (A 256-thread workgroup with
// an 8-level tree was slower: the dispatcher packs the four waves of a workgroup onto shared SIMDs, so its first two levels
// cost 4 and 2 additions' time; three sequential additions on one wave cost 3 and leave the SIMD to a single wave.)
// FINISH: lane 0 then encodes the point itself (the latency shape: a single blob is spread over up to 256 units, and a
// separate one-thread k_g1_compress launch would cost a launch and a cold start); otherwise sums[b] is written for
// k_g1_compress.
template <bool FINISH>
static __global__ __launch_bounds__(64) void k_msm_reduce_splits(const g1_xyzz* __restrict__ unit_sums, uint32_t splits, uint64_t n, g1_xyzz* __restrict__ sums,
                                                                 const int32_t* __restrict__ status, uint8_t* __restrict__ out48,
                                                                 uint8_t* __restrict__ out_affine96) {
  __shared__ g1_xyzz28 lds[32];
  issue_priority_latency();
  const int t = threadIdx.x;
  const uint64_t b = blockIdx.x;
  if (b >= n) return;
  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  if ((uint32_t)t < splits) {
    const g1_xyzz in = unit_sums[b * splits + t];
    xyzz28_from_xyzz(acc, in);
  }
#pragma unroll 1
  for (uint32_t u = 64u + (uint32_t)t; u < splits; u += 64u) {
    const g1_xyzz in = unit_sums[b * splits + u];
    g1_xyzz28 other;
    xyzz28_from_xyzz(other, in);
    xyzz28_add_complete_inl(acc, other);
  }
#pragma unroll 1
  for (int step = 1; step < 64 && (uint32_t)step < splits; step <<= 1) {
    const int m = 2 * step - 1;
    if ((t & m) == step) lds[t >> 1] = acc;
    __syncthreads();
    if ((t & m) == 0) {
      const g1_xyzz28 other = lds[(t + step) >> 1];
      xyzz28_add_complete_inl(acc, other);
    }
    __syncthreads();
  }
  if (t == 0) {
    g1_xyzz out;
    xyzz28_to_xyzz(out, acc);
    if (FINISH)
      g1_finish_item(out, b, status, out48, out_affine96);
    else
      sums[b] = out;
  }
}

// One thread per item: g1_finish_item over n sums.
static __global__ __launch_bounds__(64) void k_g1_compress(const g1_xyzz* __restrict__ sums, uint64_t n, const int32_t* __restrict__ status,
                                                    uint8_t* __restrict__ out48, uint8_t* __restrict__ out_affine96) {
  issue_priority_latency();
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const g1_xyzz acc = sums[b];
  g1_finish_item(acc, b, status, out48, out_affine96);
}

#endif  // __HIPCC__
}  // namespace kzg
