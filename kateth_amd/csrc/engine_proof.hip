// compute_blob_kzg_proof / compute_kzg_proof entry points
#include "engine_internal.hpp"
#include "poly_kernels.cuh"  // this translation unit owns the quotient / evaluation kernels of the proof path

__global__ __launch_bounds__(256) void k_merge_status(int32_t* __restrict__ primary, const int32_t* __restrict__ secondary, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && primary[i] == 0) primary[i] = secondary[i];
}

// workspace of a proof call over n items: chunks of cn items, two slots of per-chunk preparation results (z, 1/(z^4096 - 1), y,
// commitment statuses, quotient scalars) and one set of MSM buffers
struct ProofLayout {
  uint64_t cn = 0;
  uint32_t splits = 1;
  size_t o_z[2], o_y[2], o_cs[2], o_q[2], o_ir[2], o_part = 0, o_sum = 0, o_msm = 0, total = 0;
};
static ProofLayout proof_layout(const kzg_ctx* ctx, uint64_t n) {
  ProofLayout L;
  const uint64_t chunk_max = ctx->knobs.proof_chunk ? ctx->knobs.proof_chunk : 16384;
  L.cn = n < chunk_max ? n : chunk_max;
  L.splits = choose_splits(ctx, L.cn);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  for (int sl = 0; sl < 2; sl++) {
    L.o_z[sl] = take(L.cn * sizeof(fr_t));
    L.o_ir[sl] = take(2 * L.cn * sizeof(fr_t));  // per blob: 1 / prod (z - w_i) and (z^4096 - 1) / 4096
    L.o_y[sl] = take(L.cn * sizeof(fr_t));
    L.o_cs[sl] = take(L.cn * sizeof(int32_t));
    L.o_q[sl] = take(L.cn * 4096 * sizeof(fr_t));
  }
  L.o_part = take(L.cn * L.splits * 65 * sizeof(g1_xyzz));  // 64 lane sums + 1 unit sum per (blob, split)
  L.o_sum = take(L.cn * sizeof(g1_xyzz));
  L.o_msm = take(msm_scratch_bytes(ctx, L.cn));
  L.total = off;
  return L;
}

// proofs for n (blob, commitment) or (blob, z) items resident on the device.
//   d_commitments48 != null : blob proofs (z from the Fiat-Shamir challenge)
//   d_z32 != null           : proofs at caller-supplied points; y written to d_y32
static int32_t proof_dev_locked(const kzg_ctx* ctx, const uint8_t* d_blobs, const uint8_t* d_commitments48, const uint8_t* d_z32, uint64_t n,
                                uint8_t* d_out48, uint8_t* d_out_affine96, uint8_t* d_y32, int32_t* d_status, hipStream_t st) {
  if (n == 0) return 0;
  // Chunks of up to 16,384 blobs (quotient scratch: 128 KiB per blob = 2 GiB per chunk).  The per-chunk preparation
  // starts with ~6 ms of pure latency (one SHA-256 stream per blob, 2,050 sequential blocks) during which the GPU is
  // almost idle, and it cannot be hidden behind the previous chunk's MSM: that launch keeps every wave slot filled
  // with its own queued waves until it ends (a second stream, also at the highest stream priority, only got going
  // when the MSM finished).  So FEW, LARGE chunks win -- measured at n = 16,384, c = 12: 4 x 4,096 serial 244 ms,
  // 4 x 4,096 on two streams 242 ms, 2 x 8,192 224 ms, 8 x 2,048 279 ms (profiles/r01/proof_pipeline_variants.txt).
  // The two-stream pipeline stays selectable (KATETH_AMD_PROOF_OVERLAP=1, KATETH_AMD_PROOF_CHUNK).
  bool overlap = ctx->knobs.proof_overlap > 0;
  const ProofLayout L = proof_layout(ctx, n);
  const uint64_t cn = L.cn, nchunks = (n + cn - 1) / cn;
  const uint32_t splits = L.splits;
  int32_t rc = ws_reserve(ctx, L.total, st);
  if (rc) return rc;
  uint8_t* ws = ws_ptr(ctx);
  const size_t *o_z = L.o_z, *o_y = L.o_y, *o_cs = L.o_cs, *o_q = L.o_q, *o_ir = L.o_ir;
  const size_t o_msm = L.o_msm;
  g1_xyzz* partials = reinterpret_cast<g1_xyzz*>(ws + L.o_part);
  g1_xyzz* sums = reinterpret_cast<g1_xyzz*>(ws + L.o_sum);
  hipStream_t side = overlap ? ctx->side_stream : st;
  // events come from the context's pool (guarded by ctx->lock, which the caller holds): 4 per chunk + 1, created once
  std::vector<hipEvent_t>& pool = ctx->proof_events;
  while (pool.size() < 4 * nchunks + 1) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(KZG_FAIL_HIP, "event create failed");
    pool.push_back(e);
  }
  hipEvent_t* ev_prep = pool.data();
  hipEvent_t* ev_done = pool.data() + nchunks;
  hipEvent_t* ev_fork = pool.data() + 2 * nchunks;
  hipEvent_t* ev_join = pool.data() + 3 * nchunks;
  hipEvent_t ev_start = pool[4 * nchunks];
  HIP_TRY(hipMemsetAsync(d_status, 0, n * sizeof(int32_t), st));
  (void)hipEventRecord(ev_start, st);
  (void)hipStreamWaitEvent(side, ev_start, 0);
  auto prep = [&](uint64_t k) {  // enqueue chunk k's preparation on the side stream
    const int sl = (int)(k & 1);
    const uint64_t base = k * cn;
    const uint64_t m = (n - base < cn) ? (n - base) : cn;
    fr_t* z = reinterpret_cast<fr_t*>(ws + o_z[sl]);
    fr_t* y = reinterpret_cast<fr_t*>(ws + o_y[sl]);
    fr_t* invr = reinterpret_cast<fr_t*>(ws + o_ir[sl]);
    int32_t* cstat = reinterpret_cast<int32_t*>(ws + o_cs[sl]);
    fr_t* q = reinterpret_cast<fr_t*>(ws + o_q[sl]);
    const uint8_t* blobs = d_blobs + base * (uint64_t)KZG_BYTES_PER_BLOB;
    if (k >= 2) (void)hipStreamWaitEvent(side, ev_done[k - 2], 0);  // slot reuse
    (void)hipMemsetAsync(cstat, 0, m * sizeof(int32_t), side);
    if (d_commitments48) {
      const uint8_t* com = d_commitments48 + base * 48;
      if (fused_prep_fits(ctx, m, m)) {
        // commitment check and SHA-256 challenge in one launch: both are long per-lane dependency chains and must not share SIMDs
        launch_challenge_and_decode(ctx, side, blobs, com, m, z, com, m, cstat, (const uint8_t*)nullptr, (uint64_t)0, (int32_t*)nullptr, (uint4*)nullptr,
                                    (uint8_t*)nullptr);
      } else {
        // chunks above the fused-launch limit: the commitment check (one lane per point) runs beside the SHA-256
        // challenge on a second stream
        hipStream_t dec = overlap ? side : ctx->side_stream;
        if (dec != side) {
          (void)hipEventRecord(ev_fork[k], side);
          (void)hipStreamWaitEvent(dec, ev_fork[k], 0);
        }
        launch_g1_decompress(dec, com, m, cstat, (const uint8_t*)nullptr, (uint64_t)0, (int32_t*)nullptr, (uint4*)nullptr, (uint8_t*)nullptr);
        if (dec != side) (void)hipEventRecord(ev_join[k], dec);
        launch_challenge(ctx, side, blobs, com, m, z);
        if (dec != side) (void)hipStreamWaitEvent(side, ev_join[k], 0);
      }
    } else {
      launch_fr_parse(side, d_z32 + base * 32, m, z, cstat);
    }
    {
      ProfScope ps(ctx, PROF_POLY, side);
      hipLaunchKernelGGL(k_poly_root_inverse, dim3(blocks_for(m, 64)), dim3(64), 0, side, z, m, invr);
      hipLaunchKernelGGL(k_poly<true>, dim3((unsigned)m), dim3(512), 0, side, blobs, z, ctx->d_roots_brp, invr, y, q, d_status + base);
    }
    hipLaunchKernelGGL(k_merge_status, dim3(blocks_for(m, 256)), dim3(256), 0, side, d_status + base, cstat, m);
    (void)hipEventRecord(ev_prep[k], side);
  };
  if (rc == 0) {
    prep(0);
    for (uint64_t k = 0; k < nchunks; k++) {
      if (k + 1 < nchunks) prep(k + 1);
      const int sl = (int)(k & 1);
      const uint64_t base = k * cn;
      const uint64_t m = (n - base < cn) ? (n - base) : cn;
      fr_t* y = reinterpret_cast<fr_t*>(ws + o_y[sl]);
      fr_t* q = reinterpret_cast<fr_t*>(ws + o_q[sl]);
      (void)hipStreamWaitEvent(st, ev_prep[k], 0);
      rc = msm_pipeline(ctx, false, reinterpret_cast<const uint8_t*>(q), m, d_out48 ? d_out48 + base * 48 : nullptr,
                               d_out_affine96 ? d_out_affine96 + base * 96 : nullptr, d_status + base, partials, sums, splits, ws + o_msm, st);
      if (rc) break;
      if (d_y32) launch_fr_store_be(st, y, m, d_status + base, d_y32 + base * 32);
      (void)hipEventRecord(ev_done[k], st);
    }
    if (rc == 0 && hipGetLastError() != hipSuccess) rc = fail(KZG_FAIL_HIP, "proof pipeline launch failed");
  }
  if (rc != 0) (void)hipStreamSynchronize(side);
  return rc;
}

extern "C" int32_t kzg_compute_blob_proof_batch_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, uint64_t n,
                                                    void* d_out48, void* d_status, void* hip_stream) try {
  if (!ctx || (n && (!d_blobs || !d_commitments48 || !d_out48 || !d_status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> guard(ctx->lock);
  hipStream_t st = (hipStream_t)hip_stream;
  WsCall ws(ctx, st);
  int32_t rc = ws.begin();
  if (rc == 0)
    rc = proof_dev_locked(ctx, (const uint8_t*)d_blobs, (const uint8_t*)d_commitments48, nullptr, n, (uint8_t*)d_out48, nullptr, nullptr,
                          (int32_t*)d_status, st);
  if (rc == 0) rc = ws.end();
  return rc;
} catch (...) {
  return abi_exception();
}

// Host-buffer wrapper shared by the proof entry points.  Device buffers come from the context's pools (stage_lock): two
// staging slots of up to 4,096 blobs, the small inputs and results in the host-i/o pool -- a steady-state call allocates
// nothing.
//
// The batch is walked in PASSES, double-buffered: while the device path (proof_dev_locked: hash + commitment check,
// evaluation + quotient, MSM, encoding) works on pass k on the compute stream, pass k+1 is copied in on the copy stream --
// events only, the host never waits inside the loop.  Passes are 4,096 blobs (the MSM's most efficient shape); batches
// above 5,120 blobs start with a pass of 1,024 so that only 2.3 ms of the transfer are exposed.  16,384 blobs: 105.3 k
// blobs/s (99.8 k with "copy everything, then compute"); up to 5,120 blobs one pass: copy, then compute (87-92 k at 4,096).
// Measured alternatives at 4,096 blobs (round 3, profiles/r03/host_proof_variants.json): one pass 46.2 ms; passes of 1,024 +
// 3,072: 47.0 ms (the second hash latency costs what the hidden copy saves); a pipeline of 1,024-blob chunks on four
// streams, every chunk's MSM as one wave per SIMD: 44.7 ms at 4,096 but 170 ms at 16,384 -- the short kernels of later chunks
// (k_poly: 8-wave workgroups of 126 VGPRs) find no register space beside two resident MSM chunks (2 x 232 VGPRs per SIMD)
// and wait for a whole chunk to drain, so the pipeline degenerates to two-deep serial passes.
int32_t proof_host(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* side, size_t side_bytes, bool side_is_commitment, uint64_t n,
                          uint8_t* out48, uint8_t* out_affine96, uint8_t* out_y32, int32_t* status) {
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> stage_guard(ctx->stage_lock);
  int32_t rc = stage_init(ctx);
  if (rc) return rc;
  constexpr uint64_t PASS = 4096;
  std::vector<uint64_t> plan;
  {
    uint64_t rest = n;
    if (n > PASS + 1024) {  // a short first pass only where later passes can hide behind it (measured: at 4,096 blobs two passes cost what they save)
      plan.push_back(1024);
      rest -= 1024;
    }
    for (; rest > PASS; rest -= PASS) plan.push_back(PASS);
    if (rest) plan.push_back(rest);
  }
  uint64_t max_pass = 0;
  for (uint64_t m : plan) max_pass = m > max_pass ? m : max_pass;
  const uint64_t nslots = plan.size() > 1 ? 2 : 1;
  const size_t slot_bytes = (size_t)max_pass * KZG_BYTES_PER_BLOB;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_side = take(n * side_bytes), o_out = take(n * 48), o_aff = take(n * 96), o_y = take(n * 32), o_st = take(n * sizeof(int32_t));
  rc = stage_reserve(ctx, nslots * slot_bytes, off);
  if (rc) return rc;
  uint8_t* d_side = ctx->hostio + o_side;
  uint8_t* d_out = out48 ? ctx->hostio + o_out : nullptr;
  uint8_t* d_aff = out_affine96 ? ctx->hostio + o_aff : nullptr;
  uint8_t* d_y = ctx->hostio + o_y;
  int32_t* d_status = reinterpret_cast<int32_t*>(ctx->hostio + o_st);
  hipStream_t st = ctx->stage_streams[0];  // an idle non-blocking stream (the null stream would serialise against every blocking stream of the process)
  const bool one_pass = plan.size() == 1;  // nothing to overlap: the copy rides on the compute stream, no event between hardware queues
  hipStream_t copy_st = one_pass ? st : ctx->stage_copy_stream;
  std::lock_guard<std::mutex> guard(ctx->lock);
  WsCall ws(ctx, st);
  do {
    if (hipMemcpyAsync(d_side, side, n * side_bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
      rc = fail(KZG_FAIL_HIP, "host-to-device copy failed");
      break;
    }
    // one workspace slot for all passes (they follow each other on `st`), sized for the largest pass BEFORE anything is in
    // flight: growing it between passes would free memory under the pipeline
    rc = ws.begin();
    if (rc == 0) rc = ws_reserve(ctx, proof_layout(ctx, max_pass).total, st);
    uint64_t base = 0;
    for (size_t k = 0; k < plan.size() && rc == 0; base += plan[k], k++) {
      const int slot = (int)(k & 1);
      const uint64_t m = plan[k];
      uint8_t* d_blobs = ctx->stage + (size_t)slot * slot_bytes;
      // pass k-2 (same slot) must be done with the staging buffer before it is overwritten
      if (k >= 2 && hipStreamWaitEvent(copy_st, ctx->stage_done[slot], 0) != hipSuccess) {
        rc = fail(KZG_FAIL_HIP, "stream wait failed");
        break;
      }
      if (hipMemcpyAsync(d_blobs, blobs + base * (size_t)KZG_BYTES_PER_BLOB, m * (size_t)KZG_BYTES_PER_BLOB, hipMemcpyHostToDevice, copy_st) != hipSuccess ||
          (!one_pass && (hipEventRecord(ctx->stage_copied[slot], copy_st) != hipSuccess || hipStreamWaitEvent(st, ctx->stage_copied[slot], 0) != hipSuccess))) {
        rc = fail(KZG_FAIL_HIP, "host-to-device copy failed");
        break;
      }
      rc = proof_dev_locked(ctx, d_blobs, side_is_commitment ? d_side + base * side_bytes : nullptr, side_is_commitment ? nullptr : d_side + base * side_bytes,
                            m, d_out ? d_out + base * 48 : nullptr, d_aff ? d_aff + base * 96 : nullptr, out_y32 ? d_y + base * 32 : nullptr,
                            d_status + base, st);
      if (rc == 0 && hipEventRecord(ctx->stage_done[slot], st) != hipSuccess) rc = fail(KZG_FAIL_HIP, "event record failed");
    }
    if (rc) break;
    rc = ws.end();
    if (rc) break;
    if ((out48 && hipMemcpyAsync(out48, d_out, n * 48, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        (out_affine96 && hipMemcpyAsync(out_affine96, d_aff, n * 96, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        (out_y32 && hipMemcpyAsync(out_y32, d_y, n * 32, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        hipMemcpyAsync(status, d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = fail(KZG_FAIL_HIP, "device-to-host copy failed");
  } while (0);
  if (rc) (void)hipDeviceSynchronize();
  return rc;
}

extern "C" int32_t kzg_compute_blob_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n,
                                                uint8_t* out48, int32_t* status) try {
  if (!ctx || (n && (!blobs || !commitments48 || !out48 || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return (is_group(ctx) ? multi_proof : proof_host)(ctx, blobs, commitments48, 48, true, n, out48, nullptr, nullptr, status);
} catch (...) {
  return abi_exception();
}
extern "C" int32_t kzg_compute_blob_proof_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n,
                                                       uint8_t* out_affine96, int32_t* status) try {
  if (!ctx || (n && (!blobs || !commitments48 || !out_affine96 || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return (is_group(ctx) ? multi_proof : proof_host)(ctx, blobs, commitments48, 48, true, n, nullptr, out_affine96, nullptr, status);
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_compute_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_proof48,
                                           uint8_t* out_y32, int32_t* status) try {
  if (!ctx || (n && (!blobs || !z32 || !out_proof48 || !out_y32 || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return (is_group(ctx) ? multi_proof : proof_host)(ctx, blobs, z32, 32, false, n, out_proof48, nullptr, out_y32, status);
} catch (...) {
  return abi_exception();
}
extern "C" int32_t kzg_compute_proof_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_proof_affine96,
                                                  uint8_t* out_y32, int32_t* status) try {
  if (!ctx || (n && (!blobs || !z32 || !out_proof_affine96 || !out_y32 || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return (is_group(ctx) ? multi_proof : proof_host)(ctx, blobs, z32, 32, false, n, nullptr, out_proof_affine96, out_y32, status);
} catch (...) {
  return abi_exception();
}

void warm_code_object_proof() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, (const void*)k_poly_root_inverse);
  (void)hipGetLastError();
}
