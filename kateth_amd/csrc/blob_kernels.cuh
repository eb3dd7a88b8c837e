// Per-blob kernels around the MSM: synthetic input generation, Fiat-Shamir
// challenge (K4), barycentric evaluation (K5), quotient polynomial (K6),
// point decompression (K7).  See each kernel for the reference lines it replaces.
#pragma once
#include "g1.cuh"
#include "sha256.cuh"

namespace kzg {
#if defined(__HIPCC__)

// element(b, i) = SHA-256(seed_le64 || b_le64 || i_le32) mod r, 32 B big-endian
// (seeded counterpart of Blob::random, src/blob.rs:66-76).  One thread per element.
__global__ __launch_bounds__(256) void k_synth_blobs(uint64_t seed, uint64_t first_index, uint64_t elems, uint8_t* __restrict__ out) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  const uint64_t b = first_index + (e >> 12);
  const uint32_t i = (uint32_t)(e & 4095u);
  uint32_t w[16];
  w[0] = __builtin_bswap32((uint32_t)seed);
  w[1] = __builtin_bswap32((uint32_t)(seed >> 32));
  w[2] = __builtin_bswap32((uint32_t)b);
  w[3] = __builtin_bswap32((uint32_t)(b >> 32));
  w[4] = __builtin_bswap32(i);
  w[5] = 0x80000000u;
#pragma unroll
  for (int q = 6; q < 15; q++) w[q] = 0;
  w[15] = 160;  // message length in bits
  sha256_state s;
  sha256_init(s);
  sha256_block(s, w);
  fr_t v;
#pragma unroll
  for (int q = 0; q < 8; q++) v.v[7 - q] = s.h[q];
  fr_reduce_256(v);
  uint4* o = reinterpret_cast<uint4*>(out + e * 32);
  o[0] = make_uint4(__builtin_bswap32(v.v[7]), __builtin_bswap32(v.v[6]), __builtin_bswap32(v.v[5]), __builtin_bswap32(v.v[4]));
  o[1] = make_uint4(__builtin_bswap32(v.v[3]), __builtin_bswap32(v.v[2]), __builtin_bswap32(v.v[1]), __builtin_bswap32(v.v[0]));
}

#endif
}  // namespace kzg
