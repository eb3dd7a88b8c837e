// Shared by the engine translation units (engine.hip, engine_proof.hip, engine_verify.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kateth_amd.h"
#include "blob_kernels.cuh"
#include "msm_fixed.cuh"
#include "pairing.hpp"

using namespace kzg;

int32_t fail(int32_t code, const std::string& msg);
const std::string& last_error_text();

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return fail(KZG_FAIL_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
  } while (0)

struct TraceTimer {  // KATETH_AMD_TRACE=1: host-side wall-clock marks on stderr
  bool on;
  std::chrono::steady_clock::time_point t0;
  const char* what;
  explicit TraceTimer(const char* w) : on(getenv("KATETH_AMD_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), what(w) {}
  void mark(const char* label) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[kateth_amd trace] %s: %s +%.3f ms\n", what, label, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

struct kzg_ctx {
  int device = 0;
  MsmGeom geom{};
  uint4* d_table = nullptr;      // fixed-base table, table_entries(geom) * 96 B
  uint4* d_bases_brp = nullptr;  // 4096 affine Lagrange points, BRP order
  fr_t* d_roots_brp = nullptr;   // 4096 roots of unity, Montgomery, BRP order
  uint32_t* d_eval_tab = nullptr;  // 2048 x {w R, w R^2, w^2 R} in radix-2^29 limbs (k_eval_frac, verify_kernels.cuh)
  uint4* d_gen_affine = nullptr; // G1 generator, affine, 2^392-Montgomery (96 B): a term of batch verification's second lincomb
  host::pairing_ctx* pairing = nullptr;  // host: Frobenius constants + Miller lines of G2 and [tau]_2
  uint64_t table_bytes = 0;
  uint32_t num_cus = 256;
  hipStream_t side_stream = nullptr;  // non-blocking stream for work that overlaps the caller's stream
  hipStream_t copy_stream = nullptr;  // non-blocking stream for the chunked host-to-device copies of the host-buffer entry points
  // true (default): table in 2^392-Montgomery form, k_msm_fixed28 (radix-2^28 limbs, fp28.cuh);
  // KATETH_AMD_MSM_RADIX=32 at context creation: 2^384-Montgomery table, k_msm_fixed (12 x 32-bit limbs)
  bool msm_radix28 = true;
  // workspace (grown on demand, guarded by lock)
  mutable std::mutex lock;
  mutable void* ws = nullptr;
  mutable size_t ws_bytes = 0;
  mutable hipEvent_t ws_event = nullptr;  // recorded after the last enqueued user of `ws`; the next user's stream waits on it
  // profiling (kzg_profile_begin/end): event pairs around k_msm_fixed launches
  mutable bool profiling = false;
  mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  mutable size_t prof_used = 0;
};

int32_t ws_reserve(const kzg_ctx* ctx, size_t bytes);
int32_t ws_acquire(const kzg_ctx* ctx, hipStream_t st);
int32_t ws_release(const kzg_ctx* ctx, hipStream_t st);
int32_t prof_next(const kzg_ctx* ctx, hipEvent_t* e0, hipEvent_t* e1);
uint32_t choose_splits(const kzg_ctx* ctx, uint64_t n);
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// Fiat-Shamir challenges of n blobs: up to one workgroup pair per SIMD the two-wave kernel (shorter critical path per
// SHA-256 block); beyond that the chip is full and the one-lane-per-blob kernel does less total work.
static inline void launch_challenge(const kzg_ctx* ctx, hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z) {
  if (n == 0) return;
  const uint64_t split_max = (uint64_t)ctx->num_cus * 4 * 64 / 2;  // 2 waves per 64 blobs, one wave per SIMD: 32,768 on 256 CUs
  if (n <= split_max)
    hipLaunchKernelGGL(k_challenge_split, dim3(blocks_for(n, 64)), dim3(128), 0, st, blobs, commitments48, n, z);
  else
    hipLaunchKernelGGL(k_challenge, dim3(blocks_for(n, 64)), dim3(64), 0, st, blobs, commitments48, n, z);
}

// Small batches: challenges of n blobs and decoding of n_a + n_b points in one launch (k_challenge_and_decode).
constexpr uint64_t KZG_FUSED_PREP_MAX = 16384;  // 512 hash waves + 512 decode waves (verify): still one wave per SIMD on 256 CUs
static inline void launch_challenge_and_decode(hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z,
                                               const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b,
                                               int32_t* status_b, uint4* affine, uint8_t* inf) {
  const uint32_t sha_wgs = (uint32_t)blocks_for(n, 64);
  const uint32_t dec_wgs = (uint32_t)blocks_for(n_a + n_b, 128);
  hipLaunchKernelGGL(k_challenge_and_decode, dim3(sha_wgs + dec_wgs), dim3(128), 0, st, blobs, commitments48, n, z, sha_wgs, in_a, n_a, status_a, in_b,
                     n_b, status_b, affine, inf);
}

// The fixed-base MSM kernel alone over `n` scalar vectors on the device: 64 lane sums per (blob, split) unit into
// partials[unit * 64 + lane].
template <bool BE_BYTES>
static int32_t msm_launch(const kzg_ctx* ctx, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits,
                          hipStream_t st) {
  hipEvent_t pe0, pe1;
  int32_t rc = prof_next(ctx, &pe0, &pe1);
  if (rc) return rc;
  if (pe0) HIP_TRY(hipEventRecord(pe0, st));
  if (ctx->msm_radix28)
    hipLaunchKernelGGL((k_msm_fixed28<BE_BYTES>), dim3((unsigned)(n * splits)), dim3(64), 0, st, d_scalars, splits, ctx->d_table, ctx->geom,
                       partials, d_status);
  else
    hipLaunchKernelGGL((k_msm_fixed<BE_BYTES, 2>), dim3((unsigned)(n * splits)), dim3(64), 0, st, d_scalars, splits, ctx->d_table, ctx->geom,
                       partials, d_status);
  HIP_TRY(hipGetLastError());
  if (pe1) HIP_TRY(hipEventRecord(pe1, st));
  return 0;
}
// Lane sums of n blobs -> 48-byte encodings.  Two tree stages: the 64 lane sums of every (blob, split) unit, then the units
// of a blob -- 6 + log2(splits) levels of latency instead of the splits + 5 a sequential walk over the splits costs (a
// single blob uses 64 splits).  `partials` must have room for n * splits unit sums after the n * splits * 64 lane sums.
static inline int32_t msm_finish(uint64_t n, uint8_t* d_out48, const int32_t* d_status, g1_xyzz* partials, g1_xyzz* sums, uint32_t splits,
                                 hipStream_t st) {
  g1_xyzz* unit_sums = (splits == 1) ? sums : partials + (size_t)n * splits * 64;
  hipLaunchKernelGGL(k_msm_reduce, dim3((unsigned)(n * splits)), dim3(64), 0, st, partials, n * splits, unit_sums);
  if (splits > 1) hipLaunchKernelGGL(k_msm_reduce_splits, dim3((unsigned)n), dim3(64), 0, st, unit_sums, splits, n, sums);
  hipLaunchKernelGGL(k_g1_compress, dim3(blocks_for(n, 64)), dim3(64), 0, st, sums, n, d_status, d_out48);
  HIP_TRY(hipGetLastError());
  return 0;
}
// MSM + reduce + compress
template <bool BE_BYTES>
static int32_t msm_pipeline(const kzg_ctx* ctx, const uint8_t* d_scalars, uint64_t n, uint8_t* d_out48, int32_t* d_status, g1_xyzz* partials,
                            g1_xyzz* sums, uint32_t splits, hipStream_t st) {
  int32_t rc = msm_launch<BE_BYTES>(ctx, d_scalars, n, d_status, partials, splits, st);
  if (rc) return rc;
  return msm_finish(n, d_out48, d_status, partials, sums, splits, st);
}
