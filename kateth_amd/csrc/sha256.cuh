// SHA-256 (FIPS 180-4) for gfx950 -- replaces blst_sha256 (src/bls.rs:194).
// One hash state per lane; the 64-word schedule lives in 16 rotating VGPRs.
#pragma once
#include "field.cuh"
#include "sha_pair_asm.cuh"

namespace kzg {

struct sha256_state {
  uint32_t h[8];
};

KZG_HD uint32_t rotr32(uint32_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}

// a ^ b ^ c in one instruction on gfx950 (v_bitop3_b32, truth table 0x96)
KZG_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// Ch and Maj as single v_bitop3_b32 instructions (result bit = table[(a << 2) | (b << 1) | c]); hipcc
// otherwise builds Maj from an and, a xor and one bitop3
KZG_HD uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA);
#else
  return (e & f) ^ (~e & g);
#endif
}
KZG_HD uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
#else
  return (a & b) ^ (a & c) ^ (b & c);
#endif
}

KZG_HD void sha256_init(sha256_state& s) {
  s.h[0] = 0x6a09e667u;
  s.h[1] = 0xbb67ae85u;
  s.h[2] = 0x3c6ef372u;
  s.h[3] = 0xa54ff53au;
  s.h[4] = 0x510e527fu;
  s.h[5] = 0x9b05688cu;
  s.h[6] = 0x1f83d9abu;
  s.h[7] = 0x5be0cd19u;
}

KZG_HD constexpr uint32_t sha256_k(int i) {
  constexpr uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
      0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
      0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
      0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
      0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
      0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  return K[i];
}

// one 64-byte block given as 16 big-endian words
KZG_HD void sha256_block(sha256_state& s, const uint32_t* win) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = win[i];
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3];
  uint32_t e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
    uint32_t ch = sha_ch(e, f, g);
    uint32_t t1 = h + S1 + ch + sha256_k(i) + wi;
    uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
    uint32_t mj = sha_maj(a, b, c);
    uint32_t t2 = S0 + mj;
    h = g;
    g = f;
    f = e;
    e = d + t1;
    d = c;
    c = b;
    b = a;
    a = t1 + t2;
  }
  s.h[0] += a;
  s.h[1] += b;
  s.h[2] += c;
  s.h[3] += d;
  s.h[4] += e;
  s.h[5] += f;
  s.h[6] += g;
  s.h[7] += h;
}


#if defined(__HIPCC__)
// ---- SHA-256 split over two cooperating waves (latency-bound hashing of few long messages) ------------------
// A block costs one lane 1,410 dependent-issue VALU instructions, 480 of which are the message schedule and do not
// depend on the compression chain.  A PRODUCER wave expands the schedule (W[t] + K[t], t = 0..63) one block ahead into an
// LDS double buffer; the CONSUMER wave runs only the 64 rounds (14 instructions each).  Critical path per block:
// ~900 instructions instead of 1,410.  Layout wk[buf][t][lane]: consecutive lanes -> consecutive banks.
__device__ __forceinline__ void sha256_expand_to_lds(uint32_t* __restrict__ wk /* [64][64] */, int lane, const uint32_t* win /* 16 words */) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = win[i];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    wk[i * 64 + lane] = wi + sha256_k(i);
  }
}
// the same schedule in the layout of the lane-pair rounds: wk as [16 rows][SHA_PAIR_ROW_QUADS] uint4, row t / 4 holds
// W[t] + K[t] of four consecutive rounds for 64 slots (one ds_write_b128 per row here, one ds_read_b128 per row in
// sha_pair_asm.cuh); the row's last quad stays zero (the Y lanes read it)
__device__ __forceinline__ void sha256_expand_to_lds_quads(uint4* __restrict__ wk, int slot, const uint32_t* win /* 16 words */) {
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = win[i];
  uint4* out = wk + slot;
#pragma unroll
  for (int g = 0; g < 16; g++) {
    uint32_t q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = 4 * g + j;
      uint32_t wi;
      if (i < 16) {
        wi = w[i];
      } else {
        const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
        const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
        wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        w[i & 15] = wi;
      }
      q[j] = wi + sha256_k(i);
    }
    out[g * SHA_PAIR_ROW_QUADS] = make_uint4(q[0], q[1], q[2], q[3]);
  }
}
__device__ __forceinline__ void sha256_rounds_from_lds(sha256_state& s, const uint32_t* __restrict__ wk, int lane) {
  uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3];
  uint32_t e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    const uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
    const uint32_t t1 = h + S1 + sha_ch(e, f, g) + wk[i * 64 + lane];
    const uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
    const uint32_t t2 = S0 + sha_maj(a, b, c);
    h = g;
    g = f;
    f = e;
    e = d + t1;
    d = c;
    c = b;
    b = a;
    a = t1 + t2;
  }
  s.h[0] += a;
  s.h[1] += b;
  s.h[2] += c;
  s.h[3] += d;
  s.h[4] += e;
  s.h[5] += f;
  s.h[6] += g;
  s.h[7] += h;
}

// ---- the 64 rounds on a PAIR of lanes -------------------------------------------------------------------------------
// Latency path, one step further: the consumer's 14 instructions per round are two almost independent halves,
//   T1 = h + Sigma1(e) + Ch(e, f, g) + (W + K)      and      T2 = Sigma0(a) + Maj(a, b, c),
// joined only by e' = d + T1, a' = T1 + T2.  A lane X keeps (e, f, g, h), its partner Y keeps (a, b, c, d), and both run the
// SAME instructions: rotations by per-lane amounts (v_alignbit_b32 takes the shift from a VGPR), Maj(a, b, c) =
// Ch(~(a ^ b), b, c) so one Ch serves both (the selector is a0 for X and ~(a0 ^ a1) for Y: one v_bitop3_b32 with the lane's
// role mask as third operand), W + K read from LDS by X and from an all-zero region by Y.
// Partners are the lanes j and 7 - j of every group of eight (DPP row_half_mirror), so that X lanes fill DPP banks 0 and 2 and
// Y lanes banks 1 and 3: what only one role does is ONE instruction with a bank mask instead of a select and the operation --
//   hw = (W + K) + h         X banks only (Y keeps its zero)
//   t  = Sigma + Ch + hw     X: T1     Y: T2
//   n  = mirror(a3) + t      X banks only: e' = d + T1 (d is Y's a3)
//   n  = mirror(t) + t       Y banks only: a' = T1 + T2
// 10 instructions per round on the chain that bounds a single hash (14 on one lane; 11 with selects on quad_perm pairs, the
// form of rounds 2-3).  The rounds themselves are generated assembly (sha_pair_asm.cuh, tools/gen_sha_pair_asm.py:
// sha256_blocks_pair_asm1 / 2 / 4 for one, two and four consecutive blocks): a lone wave pays an issue slot for every s_nop the
// compiler puts around a DPP hazard it cannot schedule away.
struct sha256_half {
  uint32_t s[4];  // X: e, f, g, h      Y: a, b, c, d
};
// role and blob slot of consumer thread `tid` (eight consecutive threads cover four blobs)
__device__ __forceinline__ bool sha_pair_is_y(int tid) { return (tid & 4) != 0; }
__device__ __forceinline__ int sha_pair_slot(int tid) { return (tid >> 3) * 4 + ((tid & 4) ? 7 - (tid & 7) : (tid & 7)); }
__device__ __forceinline__ uint32_t sha_pair_swap(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t sha_lds_address(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
#endif

// Generic (slow-path) hashing of a short host/device byte string; used for
// the tiny transcripts (batch challenge) and by tests.
KZG_HD_NOINLINE void sha256_bytes(uint8_t* out32, const uint8_t* msg, uint64_t len) {
  sha256_state s;
  sha256_init(s);
  uint32_t w[16];
  uint64_t off = 0;
  for (; off + 64 <= len; off += 64) {
    for (int i = 0; i < 16; i++) w[i] = load_be32(msg + off + 4 * i);
    sha256_block(s, w);
  }
  uint8_t tail[128];
  uint64_t rem = len - off;
  for (uint64_t i = 0; i < 128; i++) tail[i] = 0;
  for (uint64_t i = 0; i < rem; i++) tail[i] = msg[off + i];
  tail[rem] = 0x80;
  int nb = (rem + 9 <= 64) ? 1 : 2;
  uint64_t bits = len * 8;
  for (int i = 0; i < 8; i++) tail[nb * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (int b = 0; b < nb; b++) {
    for (int i = 0; i < 16; i++) w[i] = load_be32(tail + 64 * b + 4 * i);
    sha256_block(s, w);
  }
  for (int i = 0; i < 8; i++) store_be32(out32 + 4 * i, s.h[i]);
}

}  // namespace kzg
