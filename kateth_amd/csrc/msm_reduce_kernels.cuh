// Lane-sum trees and point encoding behind the fixed-base MSM (compiled once: engine.hip owns this header).
#pragma once
#include "msm_fixed.cuh"

namespace kzg {
#if defined(__HIPCC__)

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
      xyzz28_add_complete_inl<true>(acc, other);
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
    xyzz28_add_complete_inl<true>(acc, other);
  }
#pragma unroll 1
  for (int step = 1; step < 16; step <<= 1) {
    const int m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = acc;
    __syncthreads();
    if ((lane & m) == 0) {
      const g1_xyzz28 other = lds[(lane + step) >> 1];
      xyzz28_add_complete_inl<true>(acc, other);
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
__device__ __forceinline__ void g1_finish_item(const g1_xyzz& sum, uint64_t b, const int32_t* __restrict__ status, uint8_t* __restrict__ out48,
                                               uint8_t* __restrict__ out_affine96) {
  uint32_t w[12], aff[24];
  if (status != nullptr && status[b] != 0) {
#pragma unroll
    for (int q = 0; q < 12; q++) w[q] = 0;
#pragma unroll
    for (int q = 0; q < 24; q++) aff[q] = 0;
  } else if (out_affine96) {
    g1_encode_xyzz28_words<true>(w, aff, sum);  // inversion in the radix-2^28 field (g1_decode28.cuh); all in registers
  } else {
    g1_encode_xyzz28_words<false>(w, aff, sum);
  }
  if (out48) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out48 + b * 48);
#pragma unroll
    for (int q = 0; q < 12; q++) o[q] = w[q];
  }
  if (out_affine96) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out_affine96 + b * 96);
#pragma unroll
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
    xyzz28_add_complete_inl<true>(acc, other);
  }
#pragma unroll 1
  for (int step = 1; step < 64 && (uint32_t)step < splits; step <<= 1) {
    const int m = 2 * step - 1;
    if ((t & m) == step) lds[t >> 1] = acc;
    __syncthreads();
    if ((t & m) == 0) {
      const g1_xyzz28 other = lds[(t + step) >> 1];
      xyzz28_add_complete_inl<true>(acc, other);
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

#endif
}  // namespace kzg
