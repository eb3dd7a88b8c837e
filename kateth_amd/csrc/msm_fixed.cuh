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





#endif  // __HIPCC__
}  // namespace kzg
