// Device functions behind the per-blob kernels -- SHA-256 message loading and the workgroup bodies of the challenge kernels, the
// point-decoding item -- and their constants.  No kernels here: blob_kernels.cuh (engine_blob.hip) instantiates them once.
#pragma once
#include "issue_fair.cuh"
#include "g1_decode28.cuh"
#include "sha256.cuh"
#include "scalar_load.cuh"

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




#endif  // __HIPCC__

}  // namespace kzg

