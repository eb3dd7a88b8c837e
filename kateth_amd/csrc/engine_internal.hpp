// Shared by the engine translation units (engine.hip, engine_proof.hip, engine_verify.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/kateth_amd.h"
// Host-side headers only: field / curve types and single-source math, the comb's geometry, SHA-256, the host pairing.  Every
// KERNEL header is included by exactly one translation unit, which exports host launchers for what the others need (round 5:
// the library used to carry each kernel once per engine*.hip that included its header):
//   engine.hip        setup_kernels.cuh, msm_comb.cuh, msm_fixed.cuh   -> msm_launch / msm_finish / msm_pipeline
//   engine_blob.hip   blob_kernels.cuh (hash, point decoding, scalars) -> launch_challenge*, launch_g1_decompress*, launch_fr_*
//   engine_proof.hip  poly_kernels.cuh                                 (its only user)
//   engine_verify.hip verify_kernels.cuh                               (its only user)
#include "comb_geom.hpp"
#include "g1.cuh"
#include "sha256.cuh"
#include "pairing.hpp"

using namespace kzg;

int32_t fail(int32_t code, const std::string& msg);
// The handler of every int32_t entry point's function-try-block: an exception on its way out of the library (std::bad_alloc from a
// std::vector, std::system_error from a thread start) becomes KZG_FAIL_HOST with its text as kzg_last_error(); nothing unwinds
// into the caller's C frames.
int32_t abi_exception() noexcept;
const std::string& last_error_text();
// kzg_last_error / kzg_last_error_code are thread-local: work done on a helper thread hands its error to the calling thread
struct ErrorSnapshot {
  std::string text;
  int32_t detail = 0;
};
ErrorSnapshot error_snapshot();
void error_publish(const ErrorSnapshot& e);

// Persistent helper threads of the process (engine.hip): a job goes to an idle worker, or to a new one while fewer than the
// cap exist; `false` = no worker to be had, the caller runs the job itself.  Workers never block on other jobs' results, so nested
// use (a group call whose members each split their ending over two threads) cannot deadlock.  ADVICE r04: run_on_helpers used to
// start a std::thread per job -- thread creation and a first hipSetDevice on the latency path of every group call and of every
// single-item verification.
bool helper_dispatch(std::function<void()> fn);

// Runs job(k) for k < count: job 0 on the calling thread, the others on pooled helper threads (a job no helper can take runs
// inline: nothing throws across the C boundary).  Returns the first non-zero code in k order, with THAT job's error text
// re-published on the calling thread.
template <class Job>
static int32_t run_on_helpers(uint32_t count, Job&& job) {
  if (count == 0) return 0;
  std::vector<int32_t> rc(count, 0);
  std::vector<ErrorSnapshot> err(count);
  std::vector<char> handed(count, 0);
  std::mutex mu;
  std::condition_variable cv;
  uint32_t pending = 0;
  for (uint32_t k = 1; k < count; k++) {
    {
      std::lock_guard<std::mutex> g(mu);
      pending++;
    }
    bool ok = false;
    try {
      ok = helper_dispatch([&, k]() {
        rc[k] = job(k);
        if (rc[k]) err[k] = error_snapshot();
        std::lock_guard<std::mutex> g(mu);  // notified under the lock: the waiter cannot destroy `cv` before this returns
        pending--;
        cv.notify_one();
      });
    } catch (...) {  // std::bad_alloc building the closure
    }
    if (ok) {
      handed[k] = 1;
    } else {
      std::lock_guard<std::mutex> g(mu);
      pending--;
    }
  }
  rc[0] = job(0);
  if (rc[0]) err[0] = error_snapshot();
  for (uint32_t k = 1; k < count; k++)
    if (!handed[k]) {
      rc[k] = job(k);
      if (rc[k]) err[k] = error_snapshot();
    }
  {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return pending == 0; });
  }
  for (uint32_t k = 0; k < count; k++)
    if (rc[k]) {
      error_publish(err[k]);
      return rc[k];
    }
  return 0;
}

#define HIP_TRY(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t _e = (expr);                                                                                \
    if (_e != hipSuccess) return fail(KZG_FAIL_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
  } while (0)

// Environment knobs are read ONCE, at kzg_ctx_create, into the context -- never on a call path.
struct EnvKnobs {
  bool trace = false;            // KATETH_AMD_TRACE
  uint64_t proof_chunk = 0;      // KATETH_AMD_PROOF_CHUNK (0 = default)
  int proof_overlap = -1;        // KATETH_AMD_PROOF_OVERLAP (-1 = default)
  int eval_group = 0;            // KATETH_AMD_EVAL_GROUP: 16 | 64 (0 = automatic)
  bool verify_serial = false;    // KATETH_AMD_VERIFY_SERIAL
  bool single_via_batch = false; // KATETH_AMD_SINGLE_VIA_BATCH: a single-item verification takes the batch machinery (the cross-check of the host lincomb)
  bool var_glv = false;          // KATETH_AMD_VAR_GLV=1: both lincombs of a batch of >= 32,768 items take their scalars GLV-split (built, measured and
                                 // NOT adopted in round 5: +0.2 ms per 65,536 triples, profiles/r05/verify_glv_rejected.json; kept as an independent cross-check)
  uint32_t var_seg = 1;          // 1 = on, share size chosen by the engine; >= 2: that many entries per lane (tuning sweeps); KATETH_AMD_VAR_SEG=0: the flat path's bucket sums with one thread per bucket (k_var_buckets_flat) instead of equal shares of
                                 // the sorted entry list per lane (k_var_buckets_seg): the cross-check of the balanced kernel
  bool var_msm_classic = false;  // KATETH_AMD_VAR_MSM=classic: c = 8 with per-bucket partials for every batch size (cross-check of the flat path)
  uint64_t verify_chunk = 0;     // KATETH_AMD_VERIFY_CHUNK: blobs per host-buffer staging chunk (0 = default)
  uint32_t verify_streams = 0;   // KATETH_AMD_VERIFY_STREAMS: compute streams the host-buffer verification rotates its chunks over (1..4; 0 = default)
  uint32_t comb_fair = 20;       // KATETH_AMD_COMB_FAIR=s: the MSM waves of a SIMD trade issue priority every 2^s cycles; 0 = hardware default (measurement aid)
  bool lat_table = true;         // KATETH_AMD_LAT_TABLE=0: no latency comb beside a class-22 table (measurement aid)
  bool comb_full_wave = false;   // KATETH_AMD_COMB_FULL_WAVE: never use the comb's two-blobs-per-wave mode (measurement aid)
  uint32_t msm_splits = 0;       // KATETH_AMD_MSM_SPLITS: force the (blob, split) decomposition of the fixed-base MSM (power of two <= 64; 0 = automatic)
  uint64_t challenge_split_max = 0;  // KATETH_AMD_CHALLENGE_SPLIT_MAX: largest batch hashed by the two-wave SHA-256 kernel (0 = default)
};
EnvKnobs read_env_knobs();

struct TraceTimer {  // KATETH_AMD_TRACE=1: host-side wall-clock marks on stderr
  bool on;
  std::chrono::steady_clock::time_point t0;
  const char* what;
  TraceTimer(bool enabled, const char* w) : on(enabled), t0(std::chrono::steady_clock::now()), what(w) {}
  void mark(const char* label) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[kateth_amd trace] %s: %s +%.3f ms\n", what, label, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

constexpr int KZG_STAGE_SLOTS = 4;    // x up to 4,096 blobs x 128 KiB = 2 GiB of staging at most
constexpr int KZG_STAGE_STREAMS = 4;  // per-chunk kernels rotate over these (a chunk's SHA-256 streams are latency-bound: several in flight)
#define KZG_SESSION_STREAM (reinterpret_cast<hipStream_t>(static_cast<intptr_t>(-1)))  // session_acquire: run on the session's own stream
enum ProfKind { PROF_MSM_FIXED = 0, PROF_CHALLENGE, PROF_EVAL, PROF_DECODE, PROF_POLY, PROF_VAR_MSM, PROF_REDUCE_COMPRESS, PROF_TRANSPOSE, PROF_KINDS };
static_assert(PROF_KINDS == KZG_PROF_KINDS, "include/kateth_amd.h and ProfKind disagree");
struct ProfEvent {
  int kind;
  hipEvent_t e0, e1;
};

struct kzg_verify_session;
struct kzg_ctx;
constexpr int KZG_WS_SLOTS = 3;
struct WsSlot {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t ev = nullptr;  // recorded after the last enqueued user of the slot; the next user's stream waits on it
  hipStream_t last_st = nullptr;  // the stream of that user: a call on the SAME stream is ordered behind it anyway and reuses the slot
  bool used = false;
};
// Extension point for the TEST-ONLY library (tests/window_msm/window_msm.hip = the product objects + one more translation
// unit): an alternative fixed-base MSM -- round 1's window-table kernels as independent cross-checks of the comb, and a
// time-stamping instance of the comb kernel.  The product library never sets g_msm_override_hook: kzg_ctx_create then has one
// path, and none of that code is in kateth_amd/csrc.
struct MsmOverride {
  const char* kernel_name;
  bool replaces_table;     // true: build() fills ctx->d_table / table_bytes / window_class instead of the comb build
  uint64_t adds_per_blob;  // reported by kzg_ctx_adds_per_blob when the table is replaced
  int32_t (*build)(kzg_ctx* ctx);
  // the MSM alone: 64 lane sums per (blob, split) unit into partials[unit * 64 + lane]; scratch = msm_scratch_bytes(ctx, n) bytes
  int32_t (*launch)(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits,
                    uint32_t lpb, void* scratch, hipStream_t st);
  void (*destroy)(kzg_ctx* ctx);
};
extern const MsmOverride* (*g_msm_override_hook)(kzg_ctx* ctx, uint32_t window_bits);  // nullptr in the product library

// The fixed-base tables a context computes with: the main comb, (class 22) the latency comb, their constant terms.
struct CombTables {
  CombGeom comb{}, comb_lat{};
  uint4 *d_table = nullptr, *d_table_lat = nullptr, *d_comb_k = nullptr, *d_comb_k_lat = nullptr;
  uint64_t table_bytes = 0;
  uint32_t window_class = 0;
};
struct TableChoice {  // a candidate (class, plane groups) of the automatic ladder and the free HBM it asks for
  uint32_t c, G;
  size_t need;
};

struct kzg_ctx {
  int device = 0;
  // GROUP context (kzg_config.ndev != 0): this context is member 0, `peers` are the members on the other listed devices
  // (owned, single-device contexts).  The host-buffer entry points shard over {this, peers...} (engine_multi.hip).
  std::vector<kzg_ctx*> peers;
  mutable std::atomic<uint32_t> rr{0};  // calls with fewer items than members start at a rotating member
  // KZG_CFG_BUILD_ASYNC: the chosen table is built by `build_thread` and swapped in under `lock`; the tables it replaces
  // stay allocated until kzg_ctx_destroy (launches already enqueued may still read them)
  std::thread build_thread;
  std::atomic<bool> build_cancel{false};
  mutable std::mutex build_mu;
  mutable std::condition_variable build_cv;
  bool build_running = false;  // guarded by build_mu
  int32_t build_rc = 0;
  ErrorSnapshot build_err;
  std::vector<CombTables> retired;
  uint4* d_table = nullptr;      // fixed-base comb table (msm_comb.cuh): comb_table_entries(comb) * 96 B
  bool use_comb = true;          // false only under a test override that replaces the table
  CombGeom comb{};
  const MsmOverride* msm_override = nullptr;  // test-only library, see above
  void* override_state = nullptr;
  // class 22 only: a second, small comb for latency (blocks of 8 points, 64 plane groups of 4 planes: 3 doublings per lane
  // instead of 31, 403 MB) used by calls of at most KZG_LAT_MAX_BLOBS blobs, where a lane's chain -- not the chip -- is the cost
  CombGeom comb_lat{};
  uint4* d_table_lat = nullptr;
  // the comb's constant term K = [(2^256-1)/2] * (sum of the setup points) as the STARTING VALUE of one lane per blob: that lane
  // doubles its accumulator H - 1 times, so the stored point is [c0 / 2^(H-1)] S (affine, table format: packed centred 30-bit digits of x * 2^390, 96 B -- fp30.cuh;
  // null = identity), one per table geometry
  uint4* d_comb_k = nullptr;
  uint4* d_comb_k_lat = nullptr;
  uint32_t window_class = 0;     // what kzg_ctx_window_bits reports
  uint4* d_bases_brp = nullptr;  // 4096 affine Lagrange points, BRP order
  fr_t* d_roots_brp = nullptr;   // 4096 roots of unity, Montgomery, BRP order
  uint32_t* d_eval_tab = nullptr;  // 256 hexes x twenty 9-limb slots in radix-2^29 limbs (layout: fr29.cuh; k_eval_frac, verify_kernels.cuh)
  uint4* d_gen_affine = nullptr; // G1 generator and its [z^2]-image (GLV cross-check path), affine, 2^392-Montgomery (2 x 96 B): a term of batch verification's second lincomb
  host::pairing_ctx* pairing = nullptr;  // host: Frobenius constants + Miller lines of G2 and [tau]_2
  uint64_t table_bytes = 0;
  uint32_t num_cus = 256;
  hipStream_t side_stream = nullptr;  // non-blocking stream for work that overlaps the caller's stream
  hipStream_t copy_stream = nullptr;  // non-blocking stream for the chunked host-to-device copies of the host-buffer entry points
  EnvKnobs knobs;  // read once at kzg_ctx_create
  // workspace (grown on demand, guarded by lock)
  mutable std::mutex lock;
  // KZG_WS_SLOTS workspaces for commitment / proof calls: a call takes the LOWEST slot whose previous user has completed or ran
  // on the call's own stream (a caller that runs one call at a time, or queues its calls on one stream, lives in slot 0, and only
  // slot 0 is ever allocated), else the slots in turn; its
  // stream waits for the previous user of ITS slot only, so calls enqueued on several streams run side by side (one call's
  // hash and quotient kernels in the shadow of the other's MSM) instead of queueing behind one shared buffer.  Each slot
  // grows by itself, to the largest call it has served.
  mutable WsSlot wss[KZG_WS_SLOTS];
  mutable uint32_t ws_next = 0;  // next slot in turn when every slot is busy
  mutable uint32_t ws_cur = 0;   // slot of the call being enqueued (between ws_begin and ws_end, under `lock`)
  mutable std::vector<hipEvent_t> proof_events;  // pooled fork/join events of the proof path's chunk pipeline (guarded by lock)
  // profiling (kzg_profile_begin/end): event pairs around the launches of the kernels named by ProfKind, each pair on the
  // stream its kernel runs on; own lock (the verify entry points do not take `lock`)
  mutable std::mutex prof_lock;
  mutable std::atomic<bool> profiling{false};  // read without prof_lock by ProfScope on every launch
  mutable std::vector<ProfEvent> prof_events;
  mutable size_t prof_used = 0;
  mutable unsigned long long* d_clock_probe = nullptr;  // kzg_clock_probe_launch / _read (guarded by prof_lock)
  mutable hipStream_t probe_stream = nullptr;
  // pooled verify sessions (device scratch + side stream + events), engine_verify.hip
  mutable std::mutex pool_lock;
  mutable std::vector<kzg_verify_session*> session_pool;
  // host-buffer pipelines (verification: engine_verify.hip; commitments: engine.hip; proofs: engine_proof.hip), guarded by
  // stage_lock, so that a steady-state host-buffer call allocates nothing: a staging arena of up to
  // KZG_STAGE_SLOTS chunk slots, a copy stream, rotating compute streams and their events; created on first use
  mutable std::mutex stage_lock;
  mutable uint8_t* stage = nullptr;
  mutable size_t stage_bytes = 0;
  mutable uint8_t* hostio = nullptr;  // pooled device buffers for the small inputs/outputs of the host-buffer commit/proof calls (48-96 B per item)
  mutable size_t hostio_bytes = 0;
  mutable bool stage_ready = false;
  mutable hipStream_t verify_stream = nullptr, stage_copy_stream = nullptr, stage_streams[KZG_STAGE_STREAMS] = {};
  mutable hipEvent_t stage_copied[KZG_STAGE_SLOTS] = {}, stage_done[KZG_STAGE_SLOTS] = {}, stage_join[KZG_STAGE_STREAMS] = {};
};
void session_pool_clear(const kzg_ctx* ctx);

// ---- single-device implementations behind the host-buffer entry points; the extern "C" wrappers hand GROUP contexts to
// engine_multi.hip, which shards a batch over the members and calls these per member ----
static inline bool is_group(const kzg_ctx* ctx) { return ctx && !ctx->peers.empty(); }
int32_t ctx_create_single(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, int device, kzg_ctx** out);  // engine.hip
int32_t commit_host(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, uint8_t* out_affine96, int32_t* status);    // engine.hip
int32_t proof_host(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* side, size_t side_bytes, bool side_is_commitment, uint64_t n, uint8_t* out48,
                   uint8_t* out_affine96, uint8_t* out_y32, int32_t* status);  // engine_proof.hip
int32_t verify_phase1_host(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, uint64_t n, uint8_t* out_root32,
                           int32_t* err6, kzg_verify_session** session);  // engine_verify.hip
int32_t verify_batch_host_single(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, uint64_t n, int32_t* ok);
int32_t verify_proof_single(const kzg_ctx* ctx, const uint8_t* proof48, const uint8_t* commitment48, const uint8_t* z32, const uint8_t* y32, int32_t* ok);
int32_t g1_decompress_single(const kzg_ctx* ctx, const uint8_t* in48, uint64_t n, uint8_t* out_affine96, int32_t* status);
int32_t evaluate_blobs_single(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_y32, int32_t* status);
// engine_multi.hip
int32_t group_create(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, kzg_ctx** out);
int32_t multi_commit(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, uint8_t* out_affine96, int32_t* status);
int32_t multi_proof(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* side, size_t side_bytes, bool side_is_commitment, uint64_t n, uint8_t* out48,
                    uint8_t* out_affine96, uint8_t* out_y32, int32_t* status);
int32_t multi_verify_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, uint64_t n, int32_t* ok);
int32_t multi_verify_proof(const kzg_ctx* ctx, const uint8_t* proof48, const uint8_t* commitment48, const uint8_t* z32, const uint8_t* y32, int32_t* ok);
int32_t multi_g1_decompress(const kzg_ctx* ctx, const uint8_t* in48, uint64_t n, uint8_t* out_affine96, int32_t* status);
int32_t multi_evaluate_blobs(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_y32, int32_t* status);
// one member's device-resident share of a group verification (kzg_verify_blob_proof_batch_group_dev): global range [first, first + count)
struct GroupDevShare {
  const kzg_ctx* member;
  const uint8_t *blobs, *commitments48, *proofs48;  // resident on member->device
  uint64_t first, count;
  hipStream_t st;
};
int32_t verify_group_dev(const kzg_ctx* ctx, const std::vector<GroupDevShare>& shares, uint64_t n_total, int32_t* ok);  // engine_verify.hip
int32_t stage_init(const kzg_ctx* ctx);                                             // caller holds stage_lock
int32_t stage_reserve(const kzg_ctx* ctx, size_t arena_bytes, size_t io_bytes);   // caller holds stage_lock
void stage_destroy(const kzg_ctx* ctx);

// workspace of the call being enqueued (caller holds ctx->lock from ws_begin to ws_end)
int32_t ws_begin(const kzg_ctx* ctx, hipStream_t st);    // takes the next slot; `st` waits for the slot's previous user
int32_t ws_wait(const kzg_ctx* ctx, hipStream_t st);     // a further stream of the same call waits for it too
int32_t ws_reserve(const kzg_ctx* ctx, size_t bytes, hipStream_t st);  // grows THIS slot only (waits for its previous user on the host first); if the
                                                                        // device has no room, falls back to a slot that is large enough already and makes `st` wait for it
int32_t ws_end(const kzg_ctx* ctx, hipStream_t st);      // records the slot's event on `st`
// ws_begin ... ws_end of one call; an error return in between still records the slot's event (kernels of the failed call that
// were already enqueued are then ordered before the slot's next user -- ADVICE r04)
struct WsCall {
  const kzg_ctx* ctx;
  hipStream_t st;
  bool open = false;
  WsCall(const kzg_ctx* c, hipStream_t s) : ctx(c), st(s) {}
  int32_t begin() {
    const int32_t rc = ws_begin(ctx, st);
    open = rc == 0;
    return rc;
  }
  int32_t end() {
    open = false;
    return ws_end(ctx, st);
  }
  ~WsCall() {
    if (open) (void)ws_end(ctx, st);
  }
};
static inline uint8_t* ws_ptr(const kzg_ctx* ctx) { return reinterpret_cast<uint8_t*>(ctx->wss[ctx->ws_cur].p); }
int32_t prof_next(const kzg_ctx* ctx, int kind, hipEvent_t* e0, hipEvent_t* e1);
// brackets the launches enqueued on `st` during its lifetime with an event pair (no-op unless profiling)
struct ProfScope {
  hipEvent_t e1 = nullptr;
  hipStream_t st;
  ProfScope(const kzg_ctx* ctx, int kind, hipStream_t s) : st(s) {
    hipEvent_t e0 = nullptr;
    if (ctx->profiling.load(std::memory_order_relaxed) && prof_next(ctx, kind, &e0, &e1) == 0 && e0) (void)hipEventRecord(e0, st);
  }
  ~ProfScope() {
    if (e1) (void)hipEventRecord(e1, st);
  }
};
uint32_t choose_splits(const kzg_ctx* ctx, uint64_t n);
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// Every translation unit carries its kernels in a code object of its own, which the HIP runtime loads at the unit's FIRST launch --
// behind whatever the runtime is doing then: with KZG_CFG_BUILD_ASYNC that is the background thread's 96- / 192-GiB allocation, and the
// first commitment after kzg_ctx_create waited 1.5-4.7 s for it (profiles/r05/first_use_regression.json).  kzg_ctx_create therefore
// touches one kernel of every unit (hipFuncGetAttributes) before it starts the background build.
void warm_code_object_blob();    // engine_blob.hip
void warm_code_object_proof();   // engine_proof.hip
void warm_code_object_verify();  // engine_verify.hip
// ---- launchers of engine_blob.hip (hash, point decoding, scalar parsing: blob_kernels.cuh) --------------------------------
// Fiat-Shamir challenges of n blobs (Blob::challenge, src/blob.rs:78-97) into z (plain limbs)
void launch_challenge(const kzg_ctx* ctx, hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z);
// Small batches: challenges of n blobs and decoding of n_a + n_b points in one launch
bool fused_prep_fits(const kzg_ctx* ctx, uint64_t n_blobs, uint64_t n_points);
void launch_challenge_and_decode(const kzg_ctx* ctx, hipStream_t st, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, fr_t* z, const uint8_t* in_a,
                                 uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b, int32_t* status_b, uint4* affine, uint8_t* inf);
// P1::decompress (src/bls.rs:505-531) of n_a + n_b points (two input arrays, one output array), all of them or the range [first, first + count)
void launch_g1_decompress(hipStream_t st, const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b, int32_t* status_b, uint4* affine,
                          uint8_t* inf);
void launch_g1_decompress_range(hipStream_t st, uint64_t first, uint64_t count, const uint8_t* in_a, uint64_t n_a, int32_t* status_a, const uint8_t* in_b, uint64_t n_b,
                                int32_t* status_b, uint4* affine, uint8_t* inf);
// Fr::from_be_slice (src/bls.rs:130-139) for n caller-supplied 32-byte values; plain limbs -> 32 big-endian bytes
void launch_fr_parse(hipStream_t st, const uint8_t* in32, uint64_t n, fr_t* out_plain, int32_t* status);
void launch_fr_store_be(hipStream_t st, const fr_t* plain, uint64_t n, const int32_t* status, uint8_t* out32);
void launch_synth_blobs(hipStream_t st, uint64_t seed, uint64_t first_index, uint64_t n, uint8_t* d_blobs);
constexpr uint64_t KZG_FUSED_PREP_MAX = 16384;  // the two-wave kernel's limit: 512 hash waves + 512 decode waves (verify), one wave per SIMD on 256 CUs

// The comb's half-wave mode: when two blobs per wave is the cheaper shape (engine.hip, msm_shape: from num_CUs x 16 = 4,096
// blobs per launch on, when that fills whole rounds), a blob takes 32 lanes (each lane owns twice the blocks for the same
// number of Horner doublings).  Units = waves of the MSM kernel = rows of 64 lane sums.
constexpr uint64_t KZG_LAT_MAX_BLOBS = 16;
// Units per blob on the latency comb (512 blocks x 4 planes per blob): 256 for up to 4 blobs (a lane chains 8 additions + 3
// doublings, and 1,024 waves are one per SIMD), 128 up to 8, 64 up to 16.  The main class-22 comb never splits more than 24
// ways (24 blocks per lane), so a split count >= 64 names the table.
static inline uint32_t lat_splits(uint64_t n) { return n <= 4 ? 256u : (n <= 8 ? 128u : 64u); }
static inline bool msm_uses_lat(const kzg_ctx* ctx, uint32_t splits) { return ctx->d_table_lat != nullptr && splits >= 64u; }
uint32_t msm_lanes_per_blob(const kzg_ctx* ctx, uint64_t n, uint32_t splits);  // engine.hip (msm_shape)
static inline uint64_t msm_units(uint64_t n, uint32_t splits, uint32_t lpb) { return lpb == 64 ? n * splits : (n + 1) / 2; }

// Scratch the fixed-base MSM needs besides the lane sums: the comb's bit-plane masks (the blob transposed, 128 KiB per blob).
static inline size_t msm_scratch_bytes(const kzg_ctx* ctx, uint64_t n) { return ctx->use_comb ? (size_t)n * KZG_BYTES_PER_BLOB : 0; }

// ---- launchers of engine.hip (fixed-base MSM: msm_comb.cuh, msm_fixed.cuh) ---------------------------------------------------
// The fixed-base MSM alone over `n` scalar vectors on the device: 64 lane sums per (blob, split) unit into
// partials[unit * 64 + lane].  `be_bytes`: the scalars are raw blob bytes (big-endian, canonicity checked into d_status) or
// plain little-endian limbs.  `scratch`: msm_scratch_bytes(ctx, n) bytes.  `scalars_consumed` (optional): recorded on `st` as
// soon as d_scalars is no longer read (after the bit-plane transposition; the MSM kernel reads the masks), so that a staging
// buffer can be refilled while the MSM runs.
int32_t msm_launch(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits, uint32_t lpb,
                   void* scratch, hipStream_t st, hipEvent_t scalars_consumed = nullptr);
// Lane sums of n blobs -> 48-byte encodings.  Two tree stages: the 64 lane sums of every (blob, split) unit, then the units
// of a blob.  `partials` must have room for n * splits unit sums after the n * splits * 64 lane sums.
int32_t msm_finish(const kzg_ctx* ctx, uint64_t n, uint8_t* d_out48, uint8_t* d_out_affine96, const int32_t* d_status, g1_xyzz* partials, g1_xyzz* sums,
                   uint32_t splits, uint32_t lpb, hipStream_t st);
// MSM + reduce + compress
int32_t msm_pipeline(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, uint8_t* d_out48, uint8_t* d_out_affine96, int32_t* d_status,
                     g1_xyzz* partials, g1_xyzz* sums, uint32_t splits, void* scratch, hipStream_t st);
