// verify_blob_kzg_proof_batch and the single-item verification entry points.  Device work: per-item validation, challenges,
// evaluations, transcript digests, random-linear-combination MSMs.  Host work:
// hashing ~n/8 bytes of transcript nodes, the W-step Horner combine of the
// MSM windows, and the single two-pairing check (pairing.hpp).

#include <chrono>

#include "engine_internal.hpp"
#include "verify_kernels.cuh"
struct kzg_verify_session {
  const kzg_ctx* ctx = nullptr;
  uint64_t n = 0;
  hipStream_t st = nullptr;
  uint8_t* buf = nullptr;  // one device allocation, carved below
  uint4* aff = nullptr;    // [2n+1] affine points: proofs, commitments, generator
  uint8_t* inf = nullptr;  // [2n+1]
  fr_t* z = nullptr;       // [n] plain
  fr_t* y = nullptr;       // [n] plain
  fr_t* scal = nullptr;    // [2n+1] plain: r_i*z_i (n), r_i (n), -sum r_i*y_i
  uint8_t* msm_a = nullptr;  // scratch of the two lincomb MSMs (carved from buf: no allocation in phase 2)
  uint8_t* msm_b = nullptr;
  fr_t* rpow2 = nullptr;     // [64] r^(2^k)
  fr_t* ysum = nullptr;      // per-block partial sums of r_i*y_i
};

extern "C" void kzg_verify_session_destroy(kzg_verify_session* s) {
  if (!s) return;
  if (s->buf) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->st);
    (void)hipFree(s->buf);
  }
  delete s;
}

// Window sizes whose TOP window is full (255 mod c close to c): c = 8 (32 windows,
// top raw digit 7 bits -> all 128 signed buckets used) or c = 4 for a handful of
// terms.  With e.g. c = 12 the top window has 8 distinct digits and a few
// buckets receive n/8 points each -- a serial chain that dominated the batch.
static VarGeom choose_var_geom(uint64_t nterms) {
  VarGeom g;
  g.c = nterms >= 64 ? 8u : 4u;
  g.W = (256 + g.c - 1) / g.c;
  g.half = 1u << (g.c - 1);
  return g;
}

// host: sum_j 2^(c*j) * window[j]
static void host_horner(g1_xyzz& out, const std::vector<g1_xyzz>& win, const VarGeom& g) {
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int j = (int)g.W - 1; j >= 0; j--) {
    for (uint32_t k = 0; k < g.c; k++) xyzz_dbl(acc);
    xyzz_add(acc, win[j]);
  }
  out = acc;
}

static void host_affine_from_xyzz(host::g1_host_affine& out, const g1_xyzz& p) {
  out.inf = !xyzz_to_affine(out.x, out.y, p);
  if (out.inf) {
    bn_zero(out.x);
    bn_zero(out.y);
  }
}
static void host_affine_to_be96(uint8_t* out96, const host::g1_host_affine& a) {
  if (a.inf) {
    memset(out96, 0, 96);
    return;
  }
  fp_t xp, yp;
  from_mont<FpParams>(xp, a.x);
  from_mont<FpParams>(yp, a.y);
  fp_to_be_bytes_plain(out96, xp);
  fp_to_be_bytes_plain(out96 + 48, yp);
}
static bool host_affine_from_be96(host::g1_host_affine& a, const uint8_t* in96) {
  bool zero = true;
  for (int i = 0; i < 96; i++) zero = zero && (in96[i] == 0);
  if (zero) {
    a.inf = true;
    bn_zero(a.x);
    bn_zero(a.y);
    return true;
  }
  fp_t xp, yp;
  fp_from_be_bytes_plain(xp, in96);
  fp_from_be_bytes_plain(yp, in96 + 48);
  if (bn_geq(xp, modulus<FpParams>()) || bn_geq(yp, modulus<FpParams>())) return false;
  to_mont<FpParams>(a.x, xp);
  to_mont<FpParams>(a.y, yp);
  a.inf = false;
  return true;
}

// variable-base MSM over `nterms` device-resident terms, split into an asynchronous launch
// (kernels + window read-back enqueued on `st`) and a finish (synchronise, Horner on the host)
// so that independent MSMs can run concurrently on different streams.
struct MsmVarLayout {
  VarGeom g{};
  uint32_t nb = 0, K = 1;
  size_t o_counts = 0, o_offsets = 0, o_cursors = 0, o_entries = 0, o_part = 0, o_bsum = 0, o_win = 0, total = 0;
};
static MsmVarLayout msm_var_layout(uint64_t nterms) {
  MsmVarLayout L;
  if (nterms == 0) return L;
  L.g = choose_var_geom(nterms);
  L.nb = L.g.W * L.g.half;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  L.o_counts = take((size_t)(L.nb + 1) * 4);
  L.o_offsets = take((size_t)(L.nb + 1) * 4);
  L.o_cursors = take((size_t)(L.nb + 1) * 4);
  L.o_entries = take((size_t)nterms * L.g.W * 4);
  // split every bucket over K threads (power of two <= 64) so that a thread chains ~16 additions
  uint64_t load = nterms / L.g.half + 1;
  while (L.K < 64 && (uint64_t)L.K * 16 < load) L.K <<= 1;
  L.o_part = take((size_t)L.nb * L.K * sizeof(g1_xyzz28));
  L.o_bsum = take((size_t)L.nb * sizeof(g1_xyzz28));
  L.o_win = take((size_t)L.g.W * sizeof(g1_xyzz));
  L.total = off;
  return L;
}

struct MsmVarJob {
  VarGeom g{};
  uint8_t* buf = nullptr;
  bool owns_buf = false;
  std::vector<g1_xyzz> win;
  const g1_xyzz* d_win = nullptr;
  hipStream_t st = nullptr;
  bool active = false;
};

static int32_t msm_var_launch(MsmVarJob& job, const uint4* d_points, const uint8_t* d_inf, const fr_t* d_scalars, uint64_t nterms, hipStream_t st,
                              uint8_t* prealloc = nullptr) {
  job.active = false;
  job.st = st;
  if (nterms == 0) return 0;
  const MsmVarLayout L = msm_var_layout(nterms);
  const VarGeom g = L.g;
  job.g = g;
  const uint32_t nb = L.nb, K = L.K;
  if (prealloc) {
    job.buf = prealloc;
    job.owns_buf = false;
  } else {
    HIP_TRY(hipMalloc(&job.buf, L.total));
    job.owns_buf = true;
  }
  uint8_t* buf = job.buf;
  uint32_t* counts = (uint32_t*)(buf + L.o_counts);
  uint32_t* offsets = (uint32_t*)(buf + L.o_offsets);
  uint32_t* cursors = (uint32_t*)(buf + L.o_cursors);
  uint32_t* entries = (uint32_t*)(buf + L.o_entries);
  g1_xyzz28* bpart = (g1_xyzz28*)(buf + L.o_part);
  g1_xyzz28* bsum = (g1_xyzz28*)(buf + L.o_bsum);
  g1_xyzz* winsum = (g1_xyzz*)(buf + L.o_win);
  job.win.resize(g.W);
  job.active = true;
  HIP_TRY(hipMemsetAsync(counts, 0, (size_t)(nb + 1) * 4, st));
  hipLaunchKernelGGL(k_var_count, dim3(blocks_for(nterms, 256)), dim3(256), 0, st, d_scalars, d_inf, nterms, g, counts);
  hipLaunchKernelGGL(k_var_scan, dim3(1), dim3(1024), 0, st, counts, nb, offsets, cursors);
  hipLaunchKernelGGL(k_var_scatter, dim3(blocks_for(nterms, 256)), dim3(256), 0, st, d_scalars, d_inf, nterms, g, cursors, entries);
  hipLaunchKernelGGL(k_var_buckets, dim3(blocks_for((uint64_t)nb * K, 64)), dim3(64), 0, st, d_points, offsets, entries, nb, K, bpart);
  hipLaunchKernelGGL(k_var_fold, dim3(blocks_for((uint64_t)nb * K, 64)), dim3(64), 0, st, bpart, nb, K, bsum);
  hipLaunchKernelGGL(k_var_windows, dim3(g.W), dim3(64), 0, st, bsum, g, winsum);
  HIP_TRY(hipGetLastError());
  // the window sums are read back in msm_var_finish: a device-to-host copy into pageable memory blocks the host
  // until the stream has drained, which would keep a second job from being enqueued beside this one
  job.d_win = winsum;
  return 0;
}

static int32_t msm_var_finish(MsmVarJob& job, g1_xyzz& result) {
  xyzz_set_inf(result);
  if (!job.active) return 0;
  int32_t rc = 0;
  if (hipMemcpyAsync(job.win.data(), job.d_win, (size_t)job.g.W * sizeof(g1_xyzz), hipMemcpyDeviceToHost, job.st) != hipSuccess ||
      hipStreamSynchronize(job.st) != hipSuccess)
    rc = fail(KZG_FAIL_HIP, "variable-base MSM read-back failed");
  if (rc == 0) host_horner(result, job.win, job.g);
  if (job.owns_buf) (void)hipFree(job.buf);
  job.buf = nullptr;
  job.active = false;
  return rc;
}

static int32_t msm_var(const kzg_ctx* ctx, const uint4* d_points, const uint8_t* d_inf, const fr_t* d_scalars, uint64_t nterms, hipStream_t st,
                       g1_xyzz& result) {
  (void)ctx;
  MsmVarJob job;
  int32_t rc = msm_var_launch(job, d_points, d_inf, d_scalars, nterms, st);
  if (rc) {
    if (job.buf && job.owns_buf) (void)hipFree(job.buf);
    return rc;
  }
  return msm_var_finish(job, result);
}

static void scan_first_error(const int32_t* st, uint64_t n, int32_t* idx, int32_t* code) {
  *idx = -1;
  *code = 0;
  for (uint64_t i = 0; i < n; i++)
    if (st[i]) {
      *idx = (int32_t)i;
      *code = st[i];
      return;
    }
}

extern "C" int32_t kzg_verify_phase1_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48,
                                         uint64_t n, uint8_t* out_root32, int32_t* err6, kzg_verify_session** session, void* hip_stream) {
  if (!ctx || !out_root32 || !err6 || !session || (n && (!d_blobs || !d_commitments48 || !d_proofs48)))
    return fail(KZG_FAIL_ARGUMENT, "null argument");
  *session = nullptr;
  TraceTimer tt("phase1");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  kzg_verify_session* s = new (std::nothrow) kzg_verify_session();
  if (!s) return fail(KZG_FAIL_ARGUMENT, "out of host memory");
  s->ctx = ctx;
  s->n = n;
  s->st = st;
  for (int k = 0; k < 6; k++) err6[k] = (k % 2 == 0) ? -1 : 0;
  const uint64_t groups = (n + 255) / 256;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_aff = take((2 * n + 1) * 96), o_inf = take(2 * n + 1), o_z = take(n * 32 + 32), o_y = take(n * 32 + 32),
               o_scal = take((2 * n + 1) * 32), o_stat = take(3 * n * 4 + 4), o_leaves = take(n * 32 + 32), o_nodes = take(groups * 32 + 32),
               o_msm_a = take(msm_var_layout(n).total + 256), o_msm_b = take(msm_var_layout(2 * n + 1).total + 256), o_rpow = take(64 * 32),
               o_ysum = take(((n + 255) / 256 + 1) * 32);
  if (hipMalloc(&s->buf, off) != hipSuccess) {
    delete s;
    return fail(KZG_FAIL_HIP, "hipMalloc(verify session) failed");
  }
  tt.mark("alloc");
  s->aff = (uint4*)(s->buf + o_aff);
  s->inf = s->buf + o_inf;
  s->z = (fr_t*)(s->buf + o_z);
  s->y = (fr_t*)(s->buf + o_y);
  s->scal = (fr_t*)(s->buf + o_scal);
  s->msm_a = s->buf + o_msm_a;
  s->msm_b = s->buf + o_msm_b;
  s->rpow2 = (fr_t*)(s->buf + o_rpow);
  s->ysum = (fr_t*)(s->buf + o_ysum);
  int32_t* stat = (int32_t*)(s->buf + o_stat);
  uint32_t* leaves = (uint32_t*)(s->buf + o_leaves);
  uint32_t* nodes = (uint32_t*)(s->buf + o_nodes);
  int32_t rc = 0;
  std::vector<int32_t> h_stat(3 * n);
  std::vector<uint32_t> h_nodes(groups * 8);
  do {
    // generator term
    if (hipMemcpyAsync(s->aff + (2 * n) * 6, ctx->d_gen_affine, 96, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemsetAsync(s->inf, 0, 2 * n + 1, st) != hipSuccess || hipMemsetAsync(stat, 0, 3 * n * 4 + 4, st) != hipSuccess) {
      rc = fail(KZG_FAIL_HIP, "verify session init failed");
      break;
    }
    if (n) {
      const uint8_t* blobs = (const uint8_t*)d_blobs;
      const uint8_t* com = (const uint8_t*)d_commitments48;
      const uint8_t* prf = (const uint8_t*)d_proofs48;
      // SHA-256 challenge first, alone: its 1,024 long-lived waves (one per SIMD at n = 65,536) must be
      // spread evenly -- launched next to the decode kernel they were placed around its waves and the
      // kernel took 3x longer (profiles/r01: 23 ms vs 7.5 ms).  The point decoding then runs on the side
      // stream concurrently with the evaluation kernel, whose short blocks rebalance dynamically.
      hipEvent_t ev_fork = nullptr, ev_join = nullptr;
      if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
        rc = fail(KZG_FAIL_HIP, "event create failed");
        break;
      }
      hipStream_t side = ctx->side_stream;
      // Small batches (everything together below one wave per SIMD) are latency-bound instead: there hashing and
      // decoding are ONE launch whose workgroups the dispatcher deals over different CUs (single blob: 12.4 -> 4.9 ms
      // together with the two-wave SHA-256).
      const bool small = n <= KZG_FUSED_PREP_MAX;
      if (small) {
        launch_challenge_and_decode(st, blobs, com, n, s->z, prf, n, stat + 2 * n, com, n, stat + n, s->aff, s->inf);
        (void)hipEventRecord(ev_join, st);
      } else {
        launch_challenge(ctx, st, blobs, com, n, s->z);
        (void)hipEventRecord(ev_fork, st);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        hipLaunchKernelGGL(k_g1_decompress, dim3(blocks_for(2 * n, 64)), dim3(64), 0, side, prf, n, stat + 2 * n, com, n, stat + n, s->aff, s->inf);
        (void)hipEventRecord(ev_join, side);
      }
      bool wide_groups = n < 4096;
      if (const char* e = getenv("KATETH_AMD_EVAL_GROUP")) wide_groups = atoi(e) != 16;  // tests force either shape
      if (!wide_groups)  // chip full: 16 lanes per blob (four blobs per wave), shorter merge tree
        hipLaunchKernelGGL(k_eval_frac<16>, dim3(blocks_for(n, 4)), dim3(64), 0, st, blobs, s->z, ctx->d_roots_brp, ctx->d_eval_tab, s->y, stat, n);
      else  // latency first: the whole wave on one blob
        hipLaunchKernelGGL(k_eval_frac<64>, dim3((unsigned)n), dim3(64), 0, st, blobs, s->z, ctx->d_roots_brp, ctx->d_eval_tab, s->y, stat, n);
      // the transcript hashes the input BYTES and (z, y): it does not wait for the decoded points
      hipLaunchKernelGGL(k_transcript_leaves, dim3(blocks_for(n, 256)), dim3(256), 0, st, com, prf, s->z, s->y, n, leaves);
      hipLaunchKernelGGL(k_transcript_nodes, dim3(blocks_for(groups, 64)), dim3(64), 0, st, leaves, n, nodes);
      (void)hipStreamWaitEvent(st, ev_join, 0);
      (void)hipEventDestroy(ev_fork);
      (void)hipEventDestroy(ev_join);
      if (hipGetLastError() != hipSuccess) {
        rc = fail(KZG_FAIL_HIP, "verify phase 1 launch failed");
        break;
      }
      if (hipMemcpyAsync(h_stat.data(), stat, 3 * n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipMemcpyAsync(h_nodes.data(), nodes, groups * 32, hipMemcpyDeviceToHost, st) != hipSuccess) {
        rc = fail(KZG_FAIL_HIP, "verify phase 1 readback failed");
        break;
      }
    }
    if (hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(KZG_FAIL_HIP, "verify phase 1 synchronize failed");
      break;
    }
  } while (0);
  if (rc) {
    kzg_verify_session_destroy(s);
    return rc;
  }
  tt.mark("gpu kernels + readback");
  scan_first_error(h_stat.data(), n, &err6[0], &err6[1]);
  scan_first_error(h_stat.data() + n, n, &err6[2], &err6[3]);
  scan_first_error(h_stat.data() + 2 * n, n, &err6[4], &err6[5]);
  // local transcript root = SHA-256 over the node digests (big-endian bytes)
  std::vector<uint8_t> nb(groups * 32);
  for (uint64_t k = 0; k < groups * 8; k++) store_be32(nb.data() + 4 * k, h_nodes[k]);
  sha256_bytes(out_root32, nb.data(), nb.size());
  tt.mark("status scan + root hash");
  *session = s;
  return 0;
}

extern "C" int32_t kzg_verify_phase2_dev(kzg_verify_session* s, const uint8_t* roots32, uint64_t world, uint64_t first_index, uint64_t n_total,
                                         uint8_t* out192) {
  if (!s || !roots32 || !out192 || world == 0) return fail(KZG_FAIL_ARGUMENT, "null argument");
  const kzg_ctx* ctx = s->ctx;
  TraceTimer tt("phase2");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = s->st;
  const uint64_t n = s->n;
  host::g1_host_affine A, B;
  A.inf = B.inf = true;
  bn_zero(A.x); bn_zero(A.y); bn_zero(B.x); bn_zero(B.y);
  if (n) {
    // r = hash_to_fr("RCKZGBATCH___V1_" || u128(4096) || u128(n_total) || roots)
    std::vector<uint8_t> msg(48 + 32 * world);
    memcpy(msg.data(), "RCKZGBATCH___V1_", 16);
    memset(msg.data() + 16, 0, 32);
    msg[30] = 0x10;  // 4096 as u128 big-endian
    for (int k = 0; k < 8; k++) msg[47 - k] = (uint8_t)(n_total >> (8 * k));
    memcpy(msg.data() + 48, roots32, 32 * world);
    uint8_t digest[32];
    sha256_bytes(digest, msg.data(), msg.size());
    fr_t r;
    fr_from_be_bytes_plain(r, digest);
    fr_reduce_256(r);
    to_mont<FrParams>(r, r);
    fr_t rpow2[64];
    rpow2[0] = r;
    for (int k = 1; k < 64; k++) fr_sqr(rpow2[k], rpow2[k - 1]);
    fr_t* d_rpow2 = s->rpow2;
    fr_t* d_ysum = s->ysum;
    const unsigned nblk = blocks_for(n, 256);
    int32_t rc = 0;
    g1_xyzz Ax, Bx;
    do {
      if (hipMemcpyAsync(d_rpow2, rpow2, sizeof(rpow2), hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(KZG_FAIL_HIP, "copy"); break; }
      hipLaunchKernelGGL(k_batch_scalars, dim3(nblk), dim3(256), 0, st, d_rpow2, s->z, s->y, n, first_index, s->scal + n, s->scal, d_ysum);
      hipLaunchKernelGGL(k_batch_ysum_finish, dim3(1), dim3(256), 0, st, d_ysum, nblk, s->scal + 2 * n);
      if (hipGetLastError() != hipSuccess) { rc = fail(KZG_FAIL_HIP, "verify phase 2 launch failed"); break; }
      // A = sum r_i * proof_i ; B = sum (r_i z_i) proof_i + sum r_i commitment_i - (sum r_i y_i) G
      tt.mark("seed + scalars enqueue");
      // the two lincombs are independent: A on the side stream, B on the caller's stream
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = fail(KZG_FAIL_HIP, "event create failed"); break; }
      (void)hipEventRecord(ev, st);
      (void)hipStreamWaitEvent(ctx->side_stream, ev, 0);
      (void)hipEventDestroy(ev);
      MsmVarJob ja, jb;
      rc = msm_var_launch(ja, s->aff, s->inf, s->scal + n, n, ctx->side_stream, s->msm_a);
      if (rc == 0) rc = msm_var_launch(jb, s->aff, s->inf, s->scal, 2 * n + 1, st, s->msm_b);
      int32_t rca = msm_var_finish(ja, Ax);
      int32_t rcb = msm_var_finish(jb, Bx);
      if (rc == 0) rc = rca ? rca : rcb;
      tt.mark("msm A || msm B (incl. host horner)");
    } while (0);
    (void)hipStreamSynchronize(st);
    if (rc) return rc;
    host_affine_from_xyzz(A, Ax);
    host_affine_from_xyzz(B, Bx);
    tt.mark("to affine");
  }
  host_affine_to_be96(out192, A);
  host_affine_to_be96(out192 + 96, B);
  return 0;
}

extern "C" int32_t kzg_verify_batch_finish(const kzg_ctx* ctx, const uint8_t* partials192, uint64_t world, int32_t* ok) {
  if (!ctx || !ok || (world && !partials192)) return fail(KZG_FAIL_ARGUMENT, "null argument");
  *ok = 0;
  g1_xyzz A, B;
  xyzz_set_inf(A);
  xyzz_set_inf(B);
  for (uint64_t k = 0; k < world; k++) {
    host::g1_host_affine a, b;
    if (!host_affine_from_be96(a, partials192 + 192 * k) || !host_affine_from_be96(b, partials192 + 192 * k + 96))
      return fail(KZG_FAIL_ARGUMENT, "malformed partial point");
    if (!a.inf) xyzz_madd(A, a.x, a.y);
    if (!b.inf) xyzz_madd(B, b.x, b.y);
  }
  host::g1_host_affine a, b;
  TraceTimer tt("finish");
  host_affine_from_xyzz(a, A);
  host_affine_from_xyzz(b, B);
  *ok = host::verify_pairings_fixed(*ctx->pairing, a, b) ? 1 : 0;
  tt.mark("pairing");
  return 0;
}

// first-error-wins order of the reference (src/kzg/setup.rs:259-271)
static int32_t first_error_code(const int32_t* err6) {
  if (err6[0] >= 0) return err6[1];
  if (err6[2] >= 0) return err6[3];
  if (err6[4] >= 0) return err6[5];
  return 0;
}

extern "C" int32_t kzg_verify_blob_proof_batch_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48,
                                                   uint64_t n, int32_t* ok, void* hip_stream) {
  if (!ctx || !ok) return fail(KZG_FAIL_ARGUMENT, "null argument");
  *ok = 0;
  if (n == 0) {  // reference quirk Q4: the spec answer for an empty batch is true
    *ok = 1;
    return 0;
  }
  uint8_t root[32];
  int32_t err6[6];
  kzg_verify_session* s = nullptr;
  int32_t rc = kzg_verify_phase1_dev(ctx, d_blobs, d_commitments48, d_proofs48, n, root, err6, &s, hip_stream);
  if (rc) return rc;
  const int32_t code = first_error_code(err6);
  if (code) {
    kzg_verify_session_destroy(s);
    return code;
  }
  uint8_t partial[192];
  rc = kzg_verify_phase2_dev(s, root, 1, 0, n, partial);
  kzg_verify_session_destroy(s);
  if (rc) return rc;
  return kzg_verify_batch_finish(ctx, partial, 1, ok);
}

extern "C" int32_t kzg_verify_blob_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48,
                                               uint64_t n, int32_t* ok) {
  if (!ctx || !ok || (n && (!blobs || !commitments48 || !proofs48))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  if (n == 0) {
    *ok = 1;
    return 0;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  uint8_t *d_blobs = nullptr, *d_c = nullptr, *d_p = nullptr;
  HIP_TRY(hipMalloc(&d_blobs, n * (size_t)KZG_BYTES_PER_BLOB));
  HIP_TRY(hipMalloc(&d_c, n * 48));
  HIP_TRY(hipMalloc(&d_p, n * 48));
  HIP_TRY(hipMemcpy(d_blobs, blobs, n * (size_t)KZG_BYTES_PER_BLOB, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_c, commitments48, n * 48, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_p, proofs48, n * 48, hipMemcpyHostToDevice));
  int32_t rc = kzg_verify_blob_proof_batch_dev(ctx, d_blobs, d_c, d_p, n, ok, nullptr);
  (void)hipFree(d_blobs);
  (void)hipFree(d_c);
  (void)hipFree(d_p);
  return rc;
}

// Setup::verify_blob_proof (src/kzg/setup.rs:208-221): a batch of one (the random
// coefficient is r^0 = 1, so this is exactly verify_proof_inner's equation)
extern "C" int32_t kzg_verify_blob_proof(const kzg_ctx* ctx, const uint8_t* blob, const uint8_t* commitment48, const uint8_t* proof48, int32_t* ok) {
  return kzg_verify_blob_proof_batch(ctx, blob, commitment48, proof48, 1, ok);
}

// Setup::verify_proof (src/kzg/setup.rs:96-113).  e(pi, [tau]_2 - z G2) == e(C - y G, G2)
// is checked in the equivalent fixed-G2 form  e(pi, [tau]_2) == e(C - y G + z pi, G2).
// Error order of the reference: proof, commitment, z, y.
extern "C" int32_t kzg_verify_proof(const kzg_ctx* ctx, const uint8_t* proof48, const uint8_t* commitment48, const uint8_t* z32, const uint8_t* y32,
                                    int32_t* ok) {
  if (!ctx || !proof48 || !commitment48 || !z32 || !y32 || !ok) return fail(KZG_FAIL_ARGUMENT, "null argument");
  *ok = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = nullptr;
  uint8_t* buf = nullptr;
  HIP_TRY(hipMalloc(&buf, 4096));
  uint8_t* d_in = buf;                    // proof48 || commitment48
  int32_t* d_stat = (int32_t*)(buf + 256);
  uint4* d_aff = (uint4*)(buf + 512);     // proof, commitment, generator
  uint8_t* d_inf = buf + 1024;
  fr_t* d_scal = (fr_t*)(buf + 1280);     // z, 1, -y
  int32_t rc = 0;
  int32_t h_stat[2] = {0, 0};
  g1_xyzz Bx, Ax;
  do {
    uint8_t in[96];
    memcpy(in, proof48, 48);
    memcpy(in + 48, commitment48, 48);
    if (hipMemcpy(d_in, in, 96, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_inf, 0, 3) != hipSuccess) { rc = fail(KZG_FAIL_HIP, "copy"); break; }
    hipLaunchKernelGGL(k_g1_decompress, dim3(1), dim3(64), 0, st, d_in, (uint64_t)2, d_stat, (const uint8_t*)nullptr, (uint64_t)0, (int32_t*)nullptr, d_aff, d_inf);
    if (hipMemcpy(h_stat, d_stat, 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(KZG_FAIL_HIP, "copy"); break; }
    if (h_stat[0]) { rc = h_stat[0]; break; }
    if (h_stat[1]) { rc = h_stat[1]; break; }
    fr_t z, y, one, negy;
    fr_from_be_bytes_plain(z, z32);
    fr_from_be_bytes_plain(y, y32);
    if (!fr_is_canonical(z) || !fr_is_canonical(y)) { rc = KZG_ERR_FF_NOT_IN_FIELD; break; }
    bn_zero(one);
    one.v[0] = 1;
    if (bn_is_zero(y)) negy = y; else bn_sub(negy, modulus<FrParams>(), y);
    fr_t sc[3] = {z, one, negy};
    if (hipMemcpy(d_scal, sc, sizeof(sc), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_aff + 12, ctx->d_gen_affine, 96, hipMemcpyDeviceToDevice) != hipSuccess) { rc = fail(KZG_FAIL_HIP, "copy"); break; }
    rc = msm_var(ctx, d_aff, d_inf, d_scal, 3, st, Bx);   // z*pi + C - y*G
    if (rc) break;
    rc = msm_var(ctx, d_aff, d_inf, d_scal + 1, 1, st, Ax);  // 1*pi
  } while (0);
  (void)hipFree(buf);
  if (rc) return rc;
  host::g1_host_affine a, b;
  host_affine_from_xyzz(a, Ax);
  host_affine_from_xyzz(b, Bx);
  *ok = host::verify_pairings_fixed(*ctx->pairing, a, b) ? 1 : 0;
  return 0;
}
