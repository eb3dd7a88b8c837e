// The kernels of blob_kernels.cuh -- SHA-256 challenges (four kernel families), G1 point decoding, scalar parsing / storing, the
// synthetic blob generator -- compiled ONCE, here, with the host launchers the other translation units call
// (engine_internal.hpp).  Round 5: one definition per kernel in the library.
#include "engine_internal.hpp"
#include "blob_kernels.cuh"

// Fiat-Shamir challenges of n blobs: up to one workgroup pair per SIMD the two-wave kernel (shorter critical path per
// SHA-256 block); beyond that the chip is full and the one-lane-per-blob kernel does less total work.
void launch_challenge(const kzg_ctx* ctx, hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z) {
  if (n == 0) return;
  ProfScope ps(ctx, PROF_CHALLENGE, st);
  uint64_t split_max = (uint64_t)ctx->num_cus * 4 * 64 / 2;  // 2 waves per 64 blobs, one wave per SIMD: 32,768 on 256 CUs
  if (ctx->knobs.challenge_split_max) split_max = ctx->knobs.challenge_split_max;
  if ((uint64_t)blocks_for(n, 64) <= (uint64_t)ctx->num_cus && !ctx->knobs.challenge_split_max)  // one workgroup per CU
  {  // four waves per 64 blobs, a SIMD each; 130 KiB of dynamic LDS (two sets of four block schedules)
    (void)hipFuncSetAttribute((const void*)k_challenge_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SHA_PAIR_LDS_BYTES);
    hipLaunchKernelGGL(k_challenge_pair, dim3(blocks_for(n, 64)), dim3(256), SHA_PAIR_LDS_BYTES, st, blobs, commitments48, n, z);
  }
  else if (n <= split_max)
    hipLaunchKernelGGL(k_challenge_split, dim3(blocks_for(n, 64)), dim3(128), 0, st, blobs, commitments48, n, z);
  else
    hipLaunchKernelGGL(k_challenge, dim3(blocks_for(n, 64)), dim3(64), 0, st, blobs, commitments48, n, z);
}

// Small batches: challenges of n blobs and decoding of n_a + n_b points in one launch (k_challenge_and_decode).
// Hash and decode in one launch only while every workgroup gets a CU of its own (the lane-pair kernel's workgroups take a CU
// each -- four waves, 130 KiB of LDS; in round 3's three-wave form two on one CU shared SIMDs: 4.5 ms instead of 3.7 ms per hash at 12,288 blobs); beyond that the hash
// runs alone -- still on lane pairs up to one workgroup per CU = 16,384 blobs -- and the points are decoded beside the evaluation.
bool fused_prep_fits(const kzg_ctx* ctx, uint64_t n_blobs, uint64_t n_points) {
  if (ctx->knobs.challenge_split_max) return n_blobs <= KZG_FUSED_PREP_MAX;  // tests force the two-wave / one-lane kernels
  return (uint64_t)blocks_for(n_blobs, 64) + blocks_for(n_points, 256) <= (uint64_t)ctx->num_cus;
}
void launch_challenge_and_decode(const kzg_ctx* ctx, hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z,
                                               const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b,
                                               int32_t* status_b, uint4* affine, uint8_t* inf) {
  ProfScope ps(ctx, PROF_CHALLENGE, st);
  const uint32_t sha_wgs = (uint32_t)blocks_for(n, 64);
  if ((uint64_t)sha_wgs + blocks_for(n_a + n_b, 256) <= (uint64_t)ctx->num_cus && !ctx->knobs.challenge_split_max) {
    // every wave still gets a SIMD of its own with four hash waves per 64 blobs: the rounds run on lane pairs
    const uint32_t dec_wgs = (uint32_t)blocks_for(n_a + n_b, 256);
    (void)hipFuncSetAttribute((const void*)k_challenge_pair_and_decode, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SHA_PAIR_LDS_BYTES);
    hipLaunchKernelGGL(k_challenge_pair_and_decode, dim3(sha_wgs + dec_wgs), dim3(256), SHA_PAIR_LDS_BYTES, st, blobs, commitments48, n, z, sha_wgs, in_a,
                       n_a, status_a, in_b, n_b, status_b, affine, inf);
    return;
  }
  const uint32_t dec_wgs = (uint32_t)blocks_for(n_a + n_b, 128);
  hipLaunchKernelGGL(k_challenge_and_decode, dim3(sha_wgs + dec_wgs), dim3(128), 0, st, blobs, commitments48, n, z, sha_wgs, in_a, n_a, status_a, in_b,
                     n_b, status_b, affine, inf);
}


void launch_g1_decompress(hipStream_t st, const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b, int32_t* status_b, uint4* affine,
                          uint8_t* inf) {
  if (n_a + n_b == 0) return;
  hipLaunchKernelGGL(k_g1_decompress, dim3(blocks_for(n_a + n_b, 64)), dim3(64), 0, st, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
}
void launch_g1_decompress_range(hipStream_t st, uint64_t first, uint64_t count, const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b,
                                int32_t* status_b, uint4* affine, uint8_t* inf) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_g1_decompress_range, dim3(blocks_for(count, 64)), dim3(64), 0, st, first, count, in_a, n_a, status_a, in_b, n_b, status_b, affine, inf);
}
void launch_fr_parse(hipStream_t st, const uint8_t* in32, uint64_t n, fr_t* out_plain, int32_t* status) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fr_parse, dim3(blocks_for(n, 64)), dim3(64), 0, st, in32, n, out_plain, status);
}
void launch_fr_store_be(hipStream_t st, const fr_t* plain, uint64_t n, const int32_t* status, uint8_t* out32) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fr_store_be, dim3(blocks_for(n, 256)), dim3(256), 0, st, plain, n, status, out32);
}
void launch_synth_blobs(hipStream_t st, uint64_t seed, uint64_t first_index, uint64_t n, uint8_t* d_blobs) {
  const uint64_t elems = n * 4096;
  if (elems == 0) return;
  hipLaunchKernelGGL(k_synth_blobs, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, seed, first_index, elems, d_blobs);
}

void warm_code_object_blob() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void*)k_synth_blobs);
  (void)hipGetLastError();
}
