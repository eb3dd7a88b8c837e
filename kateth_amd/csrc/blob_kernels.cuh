// Per-blob kernels around the MSM: synthetic input generation, Fiat-Shamir challenge (K4), point decompression (K7), scalar
// parsing.  See each kernel for the reference lines it replaces.  Compiled ONCE: engine_blob.hip owns this header; the
// device functions the kernels are made of live in blob_device.cuh.
#pragma once
#include "blob_device.cuh"

namespace kzg {
#if defined(__HIPCC__)

// element(b, i) = SHA-256(seed_le64 || b_le64 || i_le32) mod r, 32 B big-endian
// (seeded counterpart of Blob::random, src/blob.rs:66-76).  One thread per element.
static __global__ __launch_bounds__(256) void k_synth_blobs(uint64_t seed, uint64_t first_index, uint64_t elems, uint8_t* __restrict__ out) {
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

static __global__ __launch_bounds__(64) void k_g1_decompress(const uint8_t* __restrict__ in_a, uint64_t n_a, int32_t* __restrict__ status_a,
                                                      const uint8_t* __restrict__ in_b, uint64_t n_b, int32_t* __restrict__ status_b,
                                                      uint4* __restrict__ affine, uint8_t* __restrict__ inf) {
  g1_decompress_item((uint64_t)blockIdx.x * blockDim.x + threadIdx.x, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
}

// items [first, first + count) of the same list: a batch whose points are decoded in two launches (beside the hash kernel as
// far as SIMDs are free, the rest after it)
static __global__ __launch_bounds__(64) void k_g1_decompress_range(uint64_t first, uint64_t count, const uint8_t* __restrict__ in_a, uint64_t n_a,
                                                            int32_t* __restrict__ status_a, const uint8_t* __restrict__ in_b, uint64_t n_b,
                                                            int32_t* __restrict__ status_b, uint4* __restrict__ affine, uint8_t* __restrict__ inf) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) g1_decompress_item(first + t, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
}

// The message is  header(32 B) || blob(131072 B) || commitment(48 B): SHA block k >= 1 covers blob bytes
// [64k-32, 64k+32).  Each lane streams its blob in ALIGNED 256-byte chunks (two whole cache lines per
// step, next chunk prefetched while four blocks are hashed): a 32-byte carry from the previous chunk
// plus the chunk's 256 bytes make exactly four blocks and the next carry.  (Fetching 64 B per step --
// half a line, a new DRAM row per access, 65,536 concurrent streams -- left the kernel memory-stalled:
// 13.4 ms at n = 65,536 against 6.7 ms of pure instruction issue.)
static __global__ __launch_bounds__(64) void k_challenge(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, uint64_t n,
                                                  fr_t* __restrict__ z_plain) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const uint8_t* blob = blobs + b * 131072ull;
  const uint8_t* com = commitments48 + b * 48;
  sha256_state s;
  sha256_init(s);
  uint32_t carry[8];
  carry[0] = 0x4653424cu;  // "FSBL"
  carry[1] = 0x4f425645u;  // "OBVE"
  carry[2] = 0x52494659u;  // "RIFY"
  carry[3] = 0x5f56315fu;  // "_V1_"
  carry[4] = 0;
  carry[5] = 0;
  carry[6] = 0;
  carry[7] = 4096;  // u128 big-endian degree
  uint32_t cur[64], nxt[64];
  load_be_chunk256(nxt, blob);
#pragma unroll 1
  for (uint32_t j = 0; j < 512; j++) {
#pragma unroll
    for (int q = 0; q < 64; q++) cur[q] = nxt[q];
    if (j + 1 < 512) load_be_chunk256(nxt, blob + 256u * (j + 1));
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      w[q] = carry[q];
      w[8 + q] = cur[q];
    }
    sha256_block(s, w);
    sha256_block(s, cur + 8);
    sha256_block(s, cur + 24);
    sha256_block(s, cur + 40);
#pragma unroll
    for (int q = 0; q < 8; q++) carry[q] = cur[56 + q];
  }
  // block 2048: last 32 blob bytes (the carry) + first 32 commitment bytes
  uint32_t w[16];
#pragma unroll
  for (int q = 0; q < 8; q++) w[q] = carry[q];
  load_be_words16(w + 8, com);
  load_be_words16(w + 12, com + 16);
  sha256_block(s, w);
  // block 2049: last 16 commitment bytes, padding, bit length of 131152 bytes
  load_be_words16(w, com + 32);
  w[4] = 0x80000000u;
#pragma unroll
  for (int q = 5; q < 15; q++) w[q] = 0;
  w[15] = 131152u * 8u;
  sha256_block(s, w);
  fr_t v;
#pragma unroll
  for (int q = 0; q < 8; q++) v.v[7 - q] = s.h[q];
  fr_reduce_256(v);
  z_plain[b] = v;
}

static __global__ __launch_bounds__(128) void k_challenge_split(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, uint64_t n,
                                                                fr_t* __restrict__ z_plain) {
  __shared__ uint32_t wk[2][64 * 64];
  asm volatile("" ::: "v255", "a8");  // one wave per SIMD, whatever the dispatcher would like to pack (see k_challenge_pair)
  challenge_split_workgroup(wk, blockIdx.x, blobs, commitments48, n, z_plain);
}

static __global__ __launch_bounds__(256) void k_challenge_pair(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, uint64_t n,
                                                               fr_t* __restrict__ z_plain) {
  // Claim more than half of a SIMD's register file (nothing is stored there): a second workgroup then cannot put a wave next
  // to one of this kernel's on the same SIMD, so the dispatcher has to give every workgroup a CU of its own (256 workgroups
  // = 16,384 blobs on 256 CUs; the 130 KiB of LDS say the same).  Left to itself it paired workgroups on some CUs and the hash
  // took 4.7 instead of 3.7 ms.
  asm volatile("" ::: "v255", "a8");
  challenge_pair_workgroup(sha_pair_lds, blockIdx.x, blobs, commitments48, n, z_plain);
}

static __global__ __launch_bounds__(256) void k_challenge_pair_and_decode(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48,
                                                                          uint64_t n, fr_t* __restrict__ z_plain, uint32_t sha_wgs,
                                                                          const uint8_t* __restrict__ in_a, uint64_t n_a, int32_t* __restrict__ status_a,
                                                                          const uint8_t* __restrict__ in_b, uint64_t n_b, int32_t* __restrict__ status_b,
                                                                          uint4* __restrict__ affine, uint8_t* __restrict__ inf) {
  // as in k_challenge_pair: more than half of a SIMD's registers, so that the four waves of a workgroup land on four SIMDs (at 225
  // VGPRs two of them fit on one); the decoding workgroups lose nothing, a CU holds one workgroup of this launch anyway (LDS)
  asm volatile("" ::: "v255", "a8");
  if (blockIdx.x < sha_wgs) {
    challenge_pair_workgroup(sha_pair_lds, blockIdx.x, blobs, commitments48, n, z_plain);
  } else {
    const uint64_t t = (uint64_t)(blockIdx.x - sha_wgs) * 256 + threadIdx.x;
    g1_decompress_item(t, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
  }
}

// Challenges AND point decoding of a small batch in ONE launch: workgroups [0, sha_wgs) hash (two waves per 64 blobs),
// the rest decode 128 points each.  Both are single long dependency chains per lane; as two concurrent kernels the
// dispatcher stacked their waves on the same SIMDs (decode 4.6 ms instead of 1.9 ms beside a 3.7 ms hash).  Workgroups
// of one grid are dealt round-robin over the CUs, so with at most a few workgroups per CU every wave gets a SIMD.
static __global__ __launch_bounds__(128) void k_challenge_and_decode(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48,
                                                                     uint64_t n, fr_t* __restrict__ z_plain, uint32_t sha_wgs,
                                                                     const uint8_t* __restrict__ in_a, uint64_t n_a, int32_t* __restrict__ status_a,
                                                                     const uint8_t* __restrict__ in_b, uint64_t n_b, int32_t* __restrict__ status_b,
                                                                     uint4* __restrict__ affine, uint8_t* __restrict__ inf) {
  __shared__ uint32_t wk[2][64 * 64];
  if (blockIdx.x < sha_wgs) {
    challenge_split_workgroup(wk, blockIdx.x, blobs, commitments48, n, z_plain);
  } else {
    const uint64_t t = (uint64_t)(blockIdx.x - sha_wgs) * 128 + threadIdx.x;
    g1_decompress_item(t, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
  }
}

// z given by the caller (Setup::proof, src/kzg/setup.rs:185-194): parse + range check
static __global__ __launch_bounds__(64) void k_fr_parse(const uint8_t* __restrict__ in32, uint64_t n, fr_t* __restrict__ out_plain, int32_t* __restrict__ status) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t sc[8];
  load_scalar_be_(sc, in32 + i * 32);
  fr_t v;
#pragma unroll
  for (int q = 0; q < 8; q++) v.v[q] = sc[q];
  if (!fr_is_canonical(v)) {
    if (status[i] == 0) status[i] = KZG_ERR_FF_NOT_IN_FIELD;
    bn_zero(v);
  }
  out_plain[i] = v;
}

// plain field elements -> 32 big-endian bytes each (an item whose status is non-zero gets zero bytes)
static __global__ __launch_bounds__(256) void k_fr_store_be(const fr_t* __restrict__ plain, uint64_t n, const int32_t* __restrict__ status, uint8_t* __restrict__ out32) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t v = plain[i];
  if (status != nullptr && status[i] != 0) bn_zero(v);
  uint4* o = reinterpret_cast<uint4*>(out32 + i * 32);
  o[0] = make_uint4(__builtin_bswap32(v.v[7]), __builtin_bswap32(v.v[6]), __builtin_bswap32(v.v[5]), __builtin_bswap32(v.v[4]));
  o[1] = make_uint4(__builtin_bswap32(v.v[3]), __builtin_bswap32(v.v[2]), __builtin_bswap32(v.v[1]), __builtin_bswap32(v.v[0]));
}

#endif  // __HIPCC__

}  // namespace kzg

