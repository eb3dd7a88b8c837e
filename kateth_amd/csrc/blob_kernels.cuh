// Per-blob kernels around the MSM: synthetic input generation, Fiat-Shamir
// challenge (K4), barycentric evaluation (K5), quotient polynomial (K6),
// point decompression (K7).  See each kernel for the reference lines it replaces.
#pragma once
#include "issue_fair.cuh"
#include "g1_decode28.cuh"
#include "sha256.cuh"

namespace kzg {
#if defined(__HIPCC__)

__device__ __forceinline__ void store_affine96_(uint4* tbl, uint64_t idx, const fp_t& x, const fp_t& y) {
  uint4* p = tbl + idx * 6;
  p[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
  p[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
  p[2] = make_uint4(x.v[8], x.v[9], x.v[10], x.v[11]);
  p[3] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
  p[4] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
  p[5] = make_uint4(y.v[8], y.v[9], y.v[10], y.v[11]);
}

// 32 big-endian bytes (16-B aligned) -> 8 plain little-endian limbs
__device__ __forceinline__ void load_scalar_be_(uint32_t* sc, const uint8_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 w0 = q[0], w1 = q[1];
  sc[7] = __builtin_bswap32(w0.x);
  sc[6] = __builtin_bswap32(w0.y);
  sc[5] = __builtin_bswap32(w0.z);
  sc[4] = __builtin_bswap32(w0.w);
  sc[3] = __builtin_bswap32(w1.x);
  sc[2] = __builtin_bswap32(w1.y);
  sc[1] = __builtin_bswap32(w1.z);
  sc[0] = __builtin_bswap32(w1.w);
}

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

// ---------------------------------------------------------------------------
// K7: P1::decompress (src/bls.rs:505-531) for n points, one thread per point.
// status[i] = 0 / KZG_ERR_EC_*.  If `affine` != null the decoded point is stored
// (canonical 2^392-Montgomery x,y -- the operand format of k_var_buckets; infinity -> all-zero entry and inf[i] = 1).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void g1_decompress_item(uint64_t t, const uint8_t* __restrict__ in_a, uint64_t n_a, int32_t* __restrict__ status_a,
                                                   const uint8_t* __restrict__ in_b, uint64_t n_b, int32_t* __restrict__ status_b,
                                                   uint4* __restrict__ affine, uint8_t* __restrict__ inf) {
  // two input arrays in one launch (proofs then commitments): item t decodes a[t] or b[t - n_a]
  if (t >= n_a + n_b) return;
  const bool second = t >= n_a;
  const uint64_t i = second ? t - n_a : t;
  uint8_t buf[48];
  const uint32_t* src = reinterpret_cast<const uint32_t*>((second ? in_b : in_a) + i * 48);
#pragma unroll
  for (int q = 0; q < 12; q++) {
    uint32_t w = src[q];
    buf[4 * q] = (uint8_t)w;
    buf[4 * q + 1] = (uint8_t)(w >> 8);
    buf[4 * q + 2] = (uint8_t)(w >> 16);
    buf[4 * q + 3] = (uint8_t)(w >> 24);
  }
  fp_t x, y;
  bool is_inf = false;
  int32_t st = g1_decompress28(x, y, is_inf, buf, true);  // radix-2^28 field path; stored points stay in the 2^392 domain (k_var_buckets)
  (second ? status_b : status_a)[i] = st;
  if (affine != nullptr) {
    if (st != 0 || is_inf) {
      bn_zero(x);
      bn_zero(y);
    }
    store_affine96_(affine, t, x, y);
    inf[t] = (st == 0 && is_inf) ? 1 : 0;
  }
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

// ---------------------------------------------------------------------------
// K4: Blob::challenge (src/blob.rs:78-97) -- z = SHA-256("FSBLOBVERIFY_V1_" ||
// u128_be(4096) || blob || commitment48) mod r, one thread per blob, 2050
// sequential blocks.  The commitment BYTES are hashed as given (for a valid
// encoding compress(decompress(c)) == c).  Output: plain (non-Montgomery) limbs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load_be_words16(uint32_t* w, const uint8_t* __restrict__ p) {  // 16 B aligned source
  uint4 v = *reinterpret_cast<const uint4*>(p);
  w[0] = __builtin_bswap32(v.x);
  w[1] = __builtin_bswap32(v.y);
  w[2] = __builtin_bswap32(v.z);
  w[3] = __builtin_bswap32(v.w);
}

__device__ __forceinline__ void load_be_chunk256(uint32_t* c, const uint8_t* __restrict__ p) {  // 256 B, 16-B aligned
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    uint4 v = q[k];
    c[4 * k] = __builtin_bswap32(v.x);
    c[4 * k + 1] = __builtin_bswap32(v.y);
    c[4 * k + 2] = __builtin_bswap32(v.z);
    c[4 * k + 3] = __builtin_bswap32(v.w);
  }
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

// the PRODUCER wave of the two latency kernels: message schedule (W + K) of block k into wk[k & 1], one block ahead of the
// consumer(s); one workgroup barrier per block
__device__ __forceinline__ void challenge_producer(uint32_t (*wk)[64 * 64], int lane, const uint8_t* __restrict__ blob, const uint8_t* __restrict__ com) {
  constexpr uint32_t NBLK = 2050;
  uint32_t w[16], nxt[16];
  // block 0: "FSBLOBVERIFY_V1_" || u128_be(4096) || blob[0:32]
  w[0] = 0x4653424cu;
  w[1] = 0x4f425645u;
  w[2] = 0x52494659u;
  w[3] = 0x5f56315fu;
  w[4] = 0;
  w[5] = 0;
  w[6] = 0;
  w[7] = 4096;
  load_be_words16(w + 8, blob);
  load_be_words16(w + 12, blob + 16);
  // block 1 = blob[32:96], fetched while block 0 is expanded
  load_be_words16(nxt, blob + 32);
  load_be_words16(nxt + 4, blob + 48);
  load_be_words16(nxt + 8, blob + 64);
  load_be_words16(nxt + 12, blob + 80);
#pragma unroll 1
  for (uint32_t k = 0; k < NBLK; k++) {
    sha256_expand_to_lds(wk[k & 1], lane, w);
#pragma unroll
    for (int q = 0; q < 16; q++) w[q] = nxt[q];
    const uint32_t k2 = k + 2;  // the block after next
    if (k2 < 2048) {  // blob[64 k2 - 32, 64 k2 + 32)
      const uint8_t* src = blob + 64ull * k2 - 32;
      load_be_words16(nxt, src);
      load_be_words16(nxt + 4, src + 16);
      load_be_words16(nxt + 8, src + 32);
      load_be_words16(nxt + 12, src + 48);
    } else if (k2 == 2048) {  // last 32 blob bytes || first 32 commitment bytes
      load_be_words16(nxt, blob + 131040);
      load_be_words16(nxt + 4, blob + 131056);
      load_be_words16(nxt + 8, com);
      load_be_words16(nxt + 12, com + 16);
    } else if (k2 == 2049) {  // last 16 commitment bytes, padding, bit length of 131,152 bytes
      load_be_words16(nxt, com + 32);
      nxt[4] = 0x80000000u;
#pragma unroll
      for (int q = 5; q < 15; q++) nxt[q] = 0;
      nxt[15] = 131152u * 8u;
    }
    __syncthreads();
  }
}

// The same hash for LATENCY-bound batch sizes (a handful of waves on an otherwise idle chip: the proof path's 4,096-blob
// chunks, single-blob calls): 128-thread workgroups of 64 blobs, wave 1 expands the message schedule one block ahead
// (sha256_expand_to_lds), wave 0 runs the rounds.  3.7 ms instead of 5.6 ms per 2,050-block stream; the total
// instruction count is slightly higher, so batches that fill the chip keep k_challenge.
__device__ __forceinline__ void challenge_split_workgroup(uint32_t (*wk)[64 * 64], uint64_t wg, const uint8_t* __restrict__ blobs,
                                                          const uint8_t* __restrict__ commitments48, uint64_t n, fr_t* __restrict__ z_plain) {
  issue_priority_latency();  // a latency-bound stream: never behind an MSM wave of another stream (issue_fair.cuh)
  const int lane = threadIdx.x & 63;
  const bool producer = threadIdx.x >= 64;
  uint64_t b = wg * 64 + lane;
  const bool live = b < n;
  if (!live) b = n - 1;  // idle lanes shadow the last blob: every wave must reach every barrier
  const uint8_t* blob = blobs + b * 131072ull;
  const uint8_t* com = commitments48 + b * 48;
  constexpr uint32_t NBLK = 2050;
  if (producer) {
    challenge_producer(wk, lane, blob, com);
  } else {
    sha256_state s;
    sha256_init(s);
#pragma unroll 1
    for (uint32_t k = 0; k < NBLK; k++) {
      __syncthreads();
      sha256_rounds_from_lds(s, wk[k & 1], lane);
    }
    if (live) {
      fr_t v;
#pragma unroll
      for (int q = 0; q < 8; q++) v.v[7 - q] = s.h[q];
      fr_reduce_256(v);
      z_plain[b] = v;
    }
  }
}
static __global__ __launch_bounds__(128) void k_challenge_split(const uint8_t* __restrict__ blobs, const uint8_t* __restrict__ commitments48, uint64_t n,
                                                                fr_t* __restrict__ z_plain) {
  __shared__ uint32_t wk[2][64 * 64];
  asm volatile("" ::: "v255", "a8");  // one wave per SIMD, whatever the dispatcher would like to pack (see k_challenge_pair)
  challenge_split_workgroup(wk, blockIdx.x, blobs, commitments48, n, z_plain);
}
// Batches small enough for FOUR waves per 64 blobs to have a SIMD each (n <= 16,384 on 256 CUs; single items): the rounds
// run on lane pairs (sha256.cuh, sha_pair_asm.cuh: 10 instead of 14 instructions per round on the critical chain), so 64 blobs
// take two consumer waves + two producer waves.
// FOUR blocks per workgroup barrier (round 4): the producer expands the schedules of blocks 4s .. 4s+3 into one of two buffer
// sets while the consumers run the four blocks of the other set as ONE generated statement (sha256_blocks_pair_asm4), which
// reads every block's W + K from LDS during the block before it.  With a barrier per block the LDS latency of a block's first
// reads and the barrier itself were exposed 2,050 times per hash (~300 of 3,070 cycles per block); now 513 times.
constexpr uint32_t SHA_PAIR_STEP = 4;                                  // blocks per barrier
constexpr uint32_t SHA_PAIR_BLOCK_QUADS = 16 * SHA_PAIR_ROW_QUADS;     // [16 rows][64 slots + the Y lanes' zero quad]
constexpr uint32_t SHA_PAIR_LDS_BYTES = 2 * SHA_PAIR_STEP * SHA_PAIR_BLOCK_QUADS * 16;  // 133,120: dynamic (launch + hipFuncSetAttribute)
constexpr uint32_t SHA_PAIR_STEPS = 2050 / SHA_PAIR_STEP;              // 512 full steps, then blocks 2048 and 2049

// message words of the TWO blocks producer j (0 / 1) expands in step s -- blocks 4s + 2j and 4s + 2j + 1; in the last step
// (s = 512) block 2048 + j alone -- of one blob's challenge message
//   "FSBLOBVERIFY_V1_" || u128_be(4096) || blob || commitment48 || padding:   block k >= 1 covers blob bytes [64k - 32, 64k + 32)
__device__ __forceinline__ void challenge_step_words(uint32_t* w /* 32 */, uint32_t s, uint32_t j, const uint8_t* __restrict__ blob,
                                                     const uint8_t* __restrict__ com) {
  if (s == 0 && j == 0) {
    w[0] = 0x4653424cu;  // "FSBL"
    w[1] = 0x4f425645u;  // "OBVE"
    w[2] = 0x52494659u;  // "RIFY"
    w[3] = 0x5f56315fu;  // "_V1_"
    w[4] = 0;
    w[5] = 0;
    w[6] = 0;
    w[7] = 4096;
#pragma unroll
    for (int q = 0; q < 6; q++) load_be_words16(w + 8 + 4 * q, blob + 16 * q);
  } else if (s < SHA_PAIR_STEPS) {
    const uint8_t* src = blob + 256ull * s - 32 + 128 * j;
#pragma unroll
    for (int q = 0; q < 8; q++) load_be_words16(w + 4 * q, src + 16 * q);
  } else if (j == 0) {  // block 2048: last 32 blob bytes || first 32 commitment bytes
    load_be_words16(w, blob + 131040);
    load_be_words16(w + 4, blob + 131056);
    load_be_words16(w + 8, com);
    load_be_words16(w + 12, com + 16);
  } else {  // block 2049: last 16 commitment bytes, padding, bit length of 131,152 bytes
    load_be_words16(w, com + 32);
    w[4] = 0x80000000u;
#pragma unroll
    for (int q = 5; q < 15; q++) w[q] = 0;
    w[15] = 131152u * 8u;
  }
}

__device__ __forceinline__ void challenge_pair_workgroup(uint4* sched, uint64_t wg, const uint8_t* __restrict__ blobs,
                                                         const uint8_t* __restrict__ commitments48, uint64_t n, fr_t* __restrict__ z_plain) {
  issue_priority_latency();  // a latency-bound stream: never behind an MSM wave of another stream (issue_fair.cuh)
  // 256 threads, a wave per SIMD: [0, 128) consumer lane pairs, [128, 192) producer 0, [192, 256) producer 1 -- each producer
  // expands two of a step's four blocks for the 64 blobs (one producer wave needs as many issue slots per block as the rounds do,
  // plus its global loads and LDS writes: alone it was what the consumers waited for)
  const int tid = threadIdx.x;
  const bool producer = tid >= 128;
  const uint32_t pj = producer ? (uint32_t)(tid - 128) >> 6 : 0u;
  const int p = producer ? (tid & 63) : sha_pair_slot(tid);  // blob within the workgroup
  uint64_t b = wg * 64 + p;
  const bool live = b < n;
  if (!live) b = n - 1;  // idle lanes shadow the last blob: every wave must reach every barrier
  if (tid < (int)(2 * SHA_PAIR_STEP * 16)) sched[tid * SHA_PAIR_ROW_QUADS + 64] = make_uint4(0, 0, 0, 0);  // the Y lanes' quad of every row
  __syncthreads();
  if (producer) {
    const uint8_t* blob = blobs + b * 131072ull;
    const uint8_t* com = commitments48 + b * 48;
    uint32_t cur[32], nxt[32];
    challenge_step_words(nxt, 0, pj, blob, com);
#pragma unroll 1
    for (uint32_t s = 0; s <= SHA_PAIR_STEPS; s++) {
#pragma unroll
      for (int q = 0; q < 32; q++) cur[q] = nxt[q];
      if (s < SHA_PAIR_STEPS) challenge_step_words(nxt, s + 1, pj, blob, com);  // fetched while this step is expanded
      uint4* set = sched + (s & 1u) * SHA_PAIR_STEP * SHA_PAIR_BLOCK_QUADS;
      if (s < SHA_PAIR_STEPS) {
        sha256_expand_to_lds_quads(set + (2 * pj) * SHA_PAIR_BLOCK_QUADS, p, cur);
        sha256_expand_to_lds_quads(set + (2 * pj + 1) * SHA_PAIR_BLOCK_QUADS, p, cur + 16);
      } else {
        sha256_expand_to_lds_quads(set + pj * SHA_PAIR_BLOCK_QUADS, p, cur);
      }
      __syncthreads();
    }
  } else {
    const bool is_y = sha_pair_is_y(tid);
    sha256_state init;
    sha256_init(init);
    sha256_half st;
#pragma unroll
    for (int q = 0; q < 4; q++) st.s[q] = is_y ? init.h[q] : init.h[4 + q];
    const uint32_t k1 = is_y ? 2u : 6u, k2 = is_y ? 13u : 11u, k3 = is_y ? 22u : 25u, ymask = is_y ? 0xffffffffu : 0u;
    // the lane's quad in row 0 of block 0 of the even / odd set (Y: the zero quad), toggled by a subtraction
    constexpr uint32_t BLK = SHA_PAIR_BLOCK_QUADS * 16;
    const uint32_t even = sha_lds_address(sched + (is_y ? 64 : p)), both = 2u * even + SHA_PAIR_STEP * BLK;
    uint32_t q0 = even;
#pragma unroll 1
    for (uint32_t s = 0; s < SHA_PAIR_STEPS; s++) {
      __syncthreads();
      sha256_blocks_pair_asm4(st.s[0], st.s[1], st.s[2], st.s[3], q0, q0 + BLK, q0 + 2 * BLK, q0 + 3 * BLK, k1, k2, k3, ymask);
      q0 = both - q0;
    }
    __syncthreads();
    sha256_blocks_pair_asm2(st.s[0], st.s[1], st.s[2], st.s[3], q0, q0 + BLK, k1, k2, k3, ymask);
    uint32_t other[4];
#pragma unroll
    for (int q = 0; q < 4; q++) other[q] = sha_pair_swap(st.s[q]);
    if (live && !is_y) {  // X holds h[4..7], its neighbour's words are h[0..3]
      fr_t v;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        v.v[7 - q] = other[q];
        v.v[3 - q] = st.s[q];
      }
      fr_reduce_256(v);
      z_plain[b] = v;
    }
  }
}
extern __shared__ uint4 sha_pair_lds[];  // SHA_PAIR_LDS_BYTES (dynamic: above the 64-KiB static limit)
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

// ---------------------------------------------------------------------------
// K1 + K5 + K6: Blob::from_slice validation (src/blob.rs:26-37),
// Polynomial::evaluate (src/kzg/poly.rs:10-33) and the quotient of
// Polynomial::prove (src/kzg/poly.rs:44-66), one 512-thread workgroup per blob,
// 8 elements per thread held in registers.
//
// The reference performs one field inversion per element (4096 + 4096 per
// proof); here all 4096 denominators (z - w_i) are inverted with ONE inversion
// per blob: per-thread prefix products, a 512-leaf product tree in LDS, a single
// Fermat inversion of the root, and the inverse pushed back down the tree.
//   y   = (z^4096 - 1)/4096 * sum_i e_i w_i / (z - w_i)          (z outside the domain)
//   q_i = (e_i - y) / (w_i - z) = (y - e_i) * inv(z - w_i)
// In-domain z == w_m (poly.rs:14-18, :50-64): y = e_m and
//   q_m = w_m^-1 * sum_{j != m} (e_j - y) w_j / (w_m - w_j) = -w_m^-1 * sum_{j != m} q_j w_j .
// status[b] |= KZG_ERR_BLOB_INVALID_FIELD_ELEMENT when an element is >= r.
// Outputs are plain (non-Montgomery) little-endian limbs.
// ---------------------------------------------------------------------------
// The root of k_poly's product tree is prod_i (z - w_i) = z^4096 - 1 (with the matching factor
// replaced by 1 for an in-domain z = w_m: prod_{i != m} (w_m - w_i) = 4096 / w_m).  Its inverse is
// therefore computed here, one blob per lane, instead of serially inside every workgroup -- and with it the factor
// (z^4096 - 1) / 4096 of the barycentric sum, which needs the same twelve squarings: inv_root[2b] and inv_root[2b + 1].
static __global__ __launch_bounds__(64) void k_poly_root_inverse(const fr_t* __restrict__ z_plain, uint64_t n, fr_t* __restrict__ inv_root) {
  issue_priority_latency();
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  fr_t z, zn, r;
  to_mont<FrParams>(z, z_plain[b]);
  zn = z;
  for (int q = 0; q < 12; q++) fr_sqr(zn, zn);
  fr_sub(zn, zn, fr_one());
  fr_t f;
  {
    const uint32_t c4096[8] = KZG_FR_INV4096_MONT;
#pragma unroll
    for (int q = 0; q < 8; q++) f.v[q] = c4096[q];
  }
  if (bn_is_zero(zn)) {  // z is a 4096th root of unity: inverse of 4096 / z
    fr_mul(r, z, f);
  } else {
    fr_inv(r, zn);
  }
  inv_root[2 * b] = r;
  fr_mul(f, f, zn);  // (z^4096 - 1) / 4096, Montgomery (zero for an in-domain z: y is the matching element then)
  inv_root[2 * b + 1] = f;
}

// (512, 4): at most 128 VGPRs, so that two 8-wave workgroups share a CU (129 VGPRs would halve the occupancy)
template <bool QUOTIENT>
static __global__ __launch_bounds__(512, 4) void k_poly(const uint8_t* __restrict__ blobs, const fr_t* __restrict__ z_plain,
                                              const fr_t* __restrict__ roots_brp, const fr_t* __restrict__ inv_root,
                                              fr_t* __restrict__ y_plain, fr_t* __restrict__ q_plain, int32_t* __restrict__ status) {
  __shared__ fr_t tree[1024];
  __shared__ int sh_domain;
  __shared__ int sh_bad;
  __shared__ fr_t sh_y;
  issue_priority_latency();  // short beside an MSM launch of another stream (host-buffer proof pipeline)
  const int t = threadIdx.x;
  const uint64_t b = blockIdx.x;
  const uint8_t* blob = blobs + b * 131072ull;
  if (t == 0) {
    sh_domain = -1;
    sh_bad = 0;
  }
  __syncthreads();
  fr_t z;
  to_mont<FrParams>(z, z_plain[b]);
  // The thread's eight blob elements are NOT kept in registers: with the eight prefix products they would be 128 VGPRs before any
  // temporary, the kernel's whole budget at four waves per SIMD (592 bytes of scratch per lane while they were); they are read
  // again where they are used -- twice more, 16-KiB coalesced rows that mostly still sit in the L2.  An element stays PLAIN:
  // mont_mul(plain, X*R) = plain*X, so neither a to_mont nor a from_mont per element is needed.
  auto element = [&](int k, bool& noncanonical) -> fr_t {
    uint32_t sc[8];
    load_scalar_be_(sc, blob + (uint64_t)(k * 512 + t) * 32u);
    fr_t v;
#pragma unroll
    for (int q = 0; q < 8; q++) v.v[q] = sc[q];
    noncanonical = !fr_is_canonical(v);
    if (noncanonical) bn_zero(v);
    return v;
  };
  fr_t pre[8];
  fr_t run = fr_one();
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int i = k * 512 + t;
    bool nc;
    (void)element(k, nc);
    bad |= nc;
    fr_t d;
    fr_sub(d, z, roots_brp[i]);
    if (bn_is_zero(d)) {
      sh_domain = i;  // at most one index can match
      d = fr_one();
    }
    fr_mul(run, run, d);
    pre[k] = run;
  }
  if (bad) sh_bad = 1;
  tree[512 + t] = run;
  __syncthreads();
  // product tree: node j = node 2j * node 2j+1
  for (int width = 256; width >= 1; width >>= 1) {
    if (t < width) {
      fr_t a = tree[2 * (width + t)], c = tree[2 * (width + t) + 1], r;
      fr_mul(r, a, c);
      tree[width + t] = r;
    }
    __syncthreads();
  }
  if (t == 0) tree[1] = inv_root[2 * b];  // = 1 / tree[1], from k_poly_root_inverse
  __syncthreads();
  // push inverses down: children of j get inv(j) * sibling product
  for (int width = 1; width <= 256; width <<= 1) {
    if (t < width) {
      const int j = width + t;
      fr_t ip = tree[j], a = tree[2 * j], c = tree[2 * j + 1], ra, rc;
      fr_mul(ra, ip, c);
      fr_mul(rc, ip, a);
      tree[2 * j] = ra;
      tree[2 * j + 1] = rc;
    }
    __syncthreads();
  }
  const int domain = sh_domain;
  fr_t inv_run = tree[512 + t];  // inverse of this thread's total product
  __syncthreads();
  fr_t ysum;
  bn_zero(ysum);
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    const int i = k * 512 + t;
    const fr_t w = roots_brp[i];
    fr_t d, inv_d, term;
    fr_sub(d, z, w);
    if (i == domain) d = fr_one();
    if (k == 0)
      inv_d = inv_run;
    else
      fr_mul(inv_d, inv_run, pre[k - 1]);
    fr_mul(inv_run, inv_run, d);
    pre[k] = inv_d;  // slot k now holds 1/(z - w_i)
    fr_mul(term, w, inv_d);      // (w R)(inv_d R)/R = w inv_d R
    bool nc;
    const fr_t ek = element(k, nc);
    fr_mul(term, ek, term);      // plain e * (w inv_d R) / R = plain e w / (z - w)
    if (i != domain) fr_add(ysum, ysum, term);
  }
  // block sum of ysum
  tree[t] = ysum;
  __syncthreads();
  for (int width = 256; width >= 1; width >>= 1) {
    if (t < width) {
      fr_t a = tree[t], c = tree[t + width], r;
      fr_add(r, a, c);
      tree[t] = r;
    }
    __syncthreads();
  }
  if (t == 0) {
    fr_t total = tree[0];
    fr_mul(total, total, inv_root[2 * b + 1]);  // plain sum * Montgomery (z^4096 - 1) / 4096 = plain y
    sh_y = total;
  }
  __syncthreads();
  if (domain >= 0 && (domain & 511) == t) {
    bool nc;
    sh_y = element(domain >> 9, nc);  // y = e_m (poly.rs:14-18)
  }
  __syncthreads();
  const fr_t y = sh_y;  // plain
  if (t == 0) {
    y_plain[b] = y;
    if (sh_bad) atomicOr(&status[b], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
  if (QUOTIENT) {
    fr_t* qout = q_plain + b * 4096ull;
    fr_t ssum;
    bn_zero(ssum);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = k * 512 + t;
      fr_t q;
      bool nc;
      const fr_t ek = element(k, nc);
      fr_sub(q, y, ek);          // plain
      fr_mul(q, q, pre[k]);      // plain * (1/(z - w_i)) R / R: plain quotient element
      if (i == domain) bn_zero(q);
      if (domain >= 0) {  // block-uniform
        fr_t qw;
        fr_mul(qw, q, roots_brp[i]);  // plain q_i w_i
        fr_add(ssum, ssum, qw);
      }
      qout[i] = q;
    }
    if (domain >= 0) {  // rare in-domain branch (poly.rs:50-64)
      __syncthreads();
      tree[t] = ssum;
      __syncthreads();
      for (int width = 256; width >= 1; width >>= 1) {
        if (t < width) {
          fr_t a = tree[t], c = tree[t + width], r;
          fr_add(r, a, c);
          tree[t] = r;
        }
        __syncthreads();
      }
      if (t == 0) {
        fr_t wm = roots_brp[domain], wi, qm;
        fr_inv_fermat(wi, wm);  // rare branch, one thread: the leaner out-of-line power keeps the kernel at 128 VGPRs
        fr_mul(qm, tree[0], wi);  // plain sum * Montgomery 1/w_m = plain
        fr_neg(qm, qm);
        qout[domain] = qm;
      }
    }
  }
}


#endif
}  // namespace kzg
