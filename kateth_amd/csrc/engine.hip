// kateth_amd engine: context, workspace and the C-ABI entry points of
// include/kateth_amd.h.  All per-blob arithmetic runs in the HIP kernels of the
// .cuh files next to this one; the host only orchestrates launches (and, for
// verification, runs the single two-pairing check per call -- pairing.hpp).
// There is NO CPU compute fallback: without a HIP device every entry point
// fails with KZG_FAIL_NO_DEVICE / KZG_FAIL_HIP.
#include <new>
#include <stdexcept>

#include "engine_internal.hpp"
#include "msm_comb.cuh"            // this translation unit owns the fixed-base MSM kernels,
#include "msm_reduce_kernels.cuh"  // the lane-sum trees / encoder, and their launchers
#include "setup_kernels.cuh"

#include <algorithm>
// The engine overlaps kernels on several HIP streams; the runtime multiplexes all streams of a process onto
// GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue run one after the other.  The variable is
// read when HIP initialises (the first API call of the PROCESS), so it belongs to the host program: the library does not touch
// the environment (rounds 3-4 set it from a load-time constructor: a process-wide side effect, not thread-safe, and silently
// void when HIP was already up -- VERDICT r04 #6).  kzg_recommended_env() names the setting; kzg_ctx_create says so once under
// KATETH_AMD_TRACE when the variable is missing or low.  16: a process that keeps three calls in flight has the caller's three
// streams beside a dozen of the engine's own (copy, staging, session streams); with 8 queues two of the three lanes shared a
// queue in bench.py's proof run (+5 % instead of +11 % over one call at a time), with 12, 16 and 24 none did.
extern "C" const char* kzg_recommended_env(void) { return "GPU_MAX_HW_QUEUES=16"; }
static void hw_queues_note(bool trace) {
  static std::atomic<bool> said{false};
  if (!trace || said.exchange(true)) return;
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  if (!e || atoi(e) < 8)
    fprintf(stderr, "[kateth_amd trace] GPU_MAX_HW_QUEUES is %s: calls kept in flight on several streams may share a hardware queue and run one after "
                    "the other; export %s before the process first touches HIP (the library does not change the environment)\n",
            e ? e : "unset (the runtime's default is 4)", kzg_recommended_env());
}

// ---- pooled helper threads (engine_internal.hpp: helper_dispatch / run_on_helpers) ----------------------------------------
namespace {
class HelperPool {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  std::vector<std::thread> threads;
  uint32_t idle = 0;
  bool stop = false;
  static constexpr uint32_t CAP = 96;  // group members (<= 64) + the two-thread endings of the members' calls

  void worker() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      idle++;
      cv.wait(lk, [&] { return stop || !q.empty(); });
      idle--;
      if (q.empty()) return;  // stop
      std::function<void()> fn = std::move(q.front());
      q.pop_front();
      lk.unlock();
      fn();
      lk.lock();
    }
  }

 public:
  bool dispatch(std::function<void()>&& fn) {
    std::lock_guard<std::mutex> g(mu);
    if (stop) return false;
    if (idle > q.size()) {  // an idle worker for every queued job and for this one
      q.push_back(std::move(fn));
      cv.notify_one();
      return true;
    }
    if (threads.size() >= CAP) return false;
    try {
      threads.emplace_back([this] { worker(); });
    } catch (...) {  // std::system_error: no thread to be had
      return false;
    }
    q.push_back(std::move(fn));
    cv.notify_one();
    return true;
  }
  ~HelperPool() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    for (std::thread& t : threads)
      if (t.joinable()) t.join();
  }
};
HelperPool g_helpers;
}  // namespace
bool helper_dispatch(std::function<void()> fn) { return g_helpers.dispatch(std::move(fn)); }

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;
static thread_local int32_t g_last_detail = 0;
extern "C" const char* kzg_last_error(void) { return g_last_error.c_str(); }
extern "C" int32_t kzg_last_error_code(void) { return g_last_detail; }

int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  g_last_detail = 0;
  // the runtime keeps the code of a failed call (an out-of-memory allocation, say) until somebody asks for it: reported here, it
  // must not be found again by the launch check of this thread's NEXT, unrelated call
  if (code == KZG_FAIL_HIP) (void)hipGetLastError();
  return code;
}
int32_t abi_exception() noexcept {
  const char* what = "unexpected C++ exception";
  char text[160];
  try {
    throw;
  } catch (const std::bad_alloc&) {
    what = "out of host memory (std::bad_alloc)";
  } catch (const std::exception& e) {
    snprintf(text, sizeof text, "%s", e.what());
    what = text;
  } catch (...) {
  }
  try {
    g_last_error = std::string("host failure: ") + what;
  } catch (...) {  // no memory for the message either: keep whatever text is there
  }
  g_last_detail = 0;
  return KZG_FAIL_HOST;
}
extern "C" int32_t kzg_selftest_exception_guard(int32_t kind) try {
  if (kind == 0) throw std::bad_alloc();
  if (kind == 1) throw std::runtime_error("selftest: a runtime_error inside an entry point");
  if (kind == 2) throw 42;
  return 0;
} catch (...) {
  return abi_exception();
}
// a failure caused by a rejected input: `detail` is the KZG_ERR_* code of that input
static int32_t fail_detail(int32_t code, int32_t detail, const std::string& msg) {
  g_last_error = msg;
  g_last_detail = detail;
  return code;
}
const std::string& last_error_text() { return g_last_error; }
ErrorSnapshot error_snapshot() { return ErrorSnapshot{g_last_error, g_last_detail}; }
void error_publish(const ErrorSnapshot& e) {
  g_last_error = e.text;
  g_last_detail = e.detail;
}


// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------


static int32_t ws_grow(WsSlot& w, size_t bytes) {
  if (w.bytes >= bytes) return 0;
  size_t want = bytes;
  if (w.p) {
    if (w.ev) HIP_TRY(hipEventSynchronize(w.ev));  // nothing enqueued may still use the old buffer
    HIP_TRY(hipFree(w.p));
    w.p = nullptr;
    w.bytes = 0;
    want += bytes / 8;  // a slot that grows again is likely to see still larger calls: slack on RE-growth only
  }
  if (hipMalloc(&w.p, want) != hipSuccess) {
    (void)hipGetLastError();
    w.p = nullptr;
    if (want == bytes || hipMalloc(&w.p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      w.p = nullptr;
      return fail(KZG_FAIL_HIP, "workspace allocation (hipMalloc) of " + std::to_string(bytes >> 20) + " MiB failed: out of device memory");
    }
    want = bytes;
  }
  w.bytes = want;
  return 0;
}
// The slot of the call being enqueued grows by itself (growing all three together cost a caller that never keeps
// calls in flight 3 x the workspace, e.g. 14 instead of 5 GiB for proofs in chunks of 16,384 blobs).  If the device has no
// room for a second or third slot -- a class-16 context created at the 21-GiB rung of the ladder serves ONE 7.5-GiB proof
// workspace -- the call falls back to a slot that is large enough already and queues behind its user instead of failing.
int32_t ws_reserve(const kzg_ctx* ctx, size_t bytes, hipStream_t st) {
  WsSlot& w = ctx->wss[ctx->ws_cur];
  if (w.bytes >= bytes) return 0;
  const int32_t rc = ws_grow(w, bytes);
  if (rc == 0) return 0;
  for (uint32_t k = 0; k < (uint32_t)KZG_WS_SLOTS; k++)
    if (k != ctx->ws_cur && ctx->wss[k].bytes >= bytes) {
      ctx->ws_cur = k;
      return ws_wait(ctx, st);
    }
  return rc;
}

// A call takes the lowest slot whose previous user has completed or was enqueued on the call's own stream (stream order already
// puts this call behind it: a second slot would buy nothing and cost its allocation); when every slot is busy with other streams'
// calls, the slots in turn.  A slot's users on different streams are ordered by its event.
int32_t ws_begin(const kzg_ctx* ctx, hipStream_t st) {
  uint32_t pick = (uint32_t)KZG_WS_SLOTS;
  for (uint32_t k = 0; k < (uint32_t)KZG_WS_SLOTS && pick == (uint32_t)KZG_WS_SLOTS; k++) {
    const WsSlot& w = ctx->wss[k];
    if (!w.ev || !w.used || w.last_st == st) {  // never used, or its last user is ahead of this call on the same stream
      pick = k;
    } else {
      const hipError_t q = hipEventQuery(w.ev);
      if (q == hipSuccess) pick = k;
      else (void)hipGetLastError();  // hipErrorNotReady is not an error of this call
    }
  }
  if (pick == (uint32_t)KZG_WS_SLOTS) {
    pick = ctx->ws_next % (uint32_t)KZG_WS_SLOTS;
    ctx->ws_next = (pick + 1u) % (uint32_t)KZG_WS_SLOTS;
  }
  ctx->ws_cur = pick;
  return ws_wait(ctx, st);
}
int32_t ws_wait(const kzg_ctx* ctx, hipStream_t st) {
  const WsSlot& w = ctx->wss[ctx->ws_cur];
  if (w.ev) HIP_TRY(hipStreamWaitEvent(st, w.ev, 0));
  return 0;
}
int32_t ws_end(const kzg_ctx* ctx, hipStream_t st) {
  WsSlot& w = ctx->wss[ctx->ws_cur];
  if (!w.ev) HIP_TRY(hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(w.ev, st));
  w.last_st = st;
  w.used = true;
  return 0;
}


static void tables_free(CombTables& t);
const MsmOverride* (*g_msm_override_hook)(kzg_ctx* ctx, uint32_t window_bits) = nullptr;  // set only by the test-only library (tests/window_msm)

// Shape of a fixed-base MSM launch over n blobs: (blob, split) units of 64 lanes, or -- splits = 1, lpb = 32 -- two blobs per
// wave.  All waves of a launch are equally long, the chip holds S = num_CUs x 8 of them (two per SIMD), and they are dealt
// out as slots free up, so a launch takes ceil(waves / S) rounds of one wave's length: A / v additions (A = planes per lane x
// blocks per lane) + H - 1 Horner doublings (0.75 of an addition each), repeated by every split.  2,050 blobs as 2,050 waves
// would run a second round for two waves; 6 splits make it 7 rounds of 151 instead of 2 of 791.  The cheapest shape wins; ties
// go to fewer units (less lane-sum tree work, charged as 10 additions per round).
struct MsmShapeCost {
  uint32_t splits;
  uint32_t lpb;
  double cost;
};
static MsmShapeCost msm_shape(const kzg_ctx* ctx, uint64_t n) {
  const double S = (double)ctx->num_cus * 8.0;
  const uint32_t per_lane = (64u * ctx->comb.nb) / ctx->comb.lpg;  // blocks a lane owns at 64 lanes per blob, one split
  const double A = (double)ctx->comb.H * per_lane, D = 0.75 * (ctx->comb.H - 1u);
  auto rounds = [&](double waves) { return waves <= S ? 1.0 : (double)(uint64_t)((waves + S - 1) / S); };
  MsmShapeCost best{1, 64, 0};
  bool have = false;
  for (uint32_t v = 1; v <= 64 && v <= per_lane; v++) {
    if (per_lane % v != 0) continue;  // a lane owns a whole number of blocks; k_msm_reduce_splits sums <= 64 units
    const double c = rounds((double)n * v) * (A / v + D + 10.0);
    if (!have || c < best.cost) {
      best = MsmShapeCost{v, 64, c};
      have = true;
    }
  }
  const bool half_ok = !ctx->knobs.comb_full_wave && ctx->comb.G <= 32 && (64u * ctx->comb.nb) % (32u / ctx->comb.G) == 0;
  if (half_ok && n >= 2) {
    const double c = rounds((double)((n + 1) / 2)) * (2.0 * A + D + 10.0);
    if (c < best.cost) best = MsmShapeCost{1, 32, c};
  }
  return best;
}

uint32_t choose_splits(const kzg_ctx* ctx, uint64_t n) {
  if (ctx->use_comb) {
    if (ctx->d_table_lat && n <= KZG_LAT_MAX_BLOBS && !ctx->knobs.msm_splits) return lat_splits(n);  // the latency comb: 2 to 8 blocks x 4 planes per lane
    const uint32_t per_lane = (64u * ctx->comb.nb) / ctx->comb.lpg;
    if (ctx->knobs.msm_splits && ctx->knobs.msm_splits <= 64 && per_lane % ctx->knobs.msm_splits == 0) return ctx->knobs.msm_splits;
    return msm_shape(ctx, n).splits;
  }
  // test-only window-table override: any power of two, enough units for 8 waves per CU
  if (ctx->knobs.msm_splits) return ctx->knobs.msm_splits;
  const uint64_t target = (uint64_t)ctx->num_cus * 8;
  uint32_t s = 1;
  while (s < 64 && n * s < target) s <<= 1;
  return s;
}
uint32_t msm_lanes_per_blob(const kzg_ctx* ctx, uint64_t n, uint32_t splits) {
  if (!ctx->use_comb || splits != 1 || ctx->knobs.msm_splits) return 64;
  if (msm_uses_lat(ctx, splits)) return 64;
  return msm_shape(ctx, n).lpb;
}

extern "C" uint64_t kzg_ctx_adds_per_blob(const kzg_ctx* ctx) {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return ctx->use_comb ? (uint64_t)256u * 64u * ctx->comb.nb : ctx->msm_override->adds_per_blob;
}

extern "C" int32_t kzg_profile_begin(const kzg_ctx* ctx) try {
  if (!ctx) return fail(KZG_FAIL_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> guard(ctx->prof_lock);
  ctx->prof_used = 0;
  ctx->profiling.store(true);
  return 0;
} catch (...) {
  return abi_exception();
}

static const char* const PROF_NAMES[PROF_KINDS] = {"k_msm_comb30", "k_challenge*", "k_eval_frac", "k_g1_decompress", "k_poly",
                                                    "k_var_* (two lincombs)", "k_msm_reduce* + k_g1_compress", "k_comb_transpose"};

extern "C" const char* kzg_ctx_msm_kernel_name(const kzg_ctx* ctx) { return (ctx && ctx->msm_override) ? ctx->msm_override->kernel_name : "k_msm_comb30"; }
extern "C" int32_t kzg_ctx_plane_groups(const kzg_ctx* ctx) try {
  if (!ctx || !ctx->use_comb) return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return (int32_t)ctx->comb.G;
} catch (...) {
  return abi_exception();
}
extern "C" const char* kzg_profile_kind_name(int32_t kind) { return (kind >= 0 && kind < PROF_KINDS) ? PROF_NAMES[kind] : ""; }

extern "C" int32_t kzg_profile_end_kinds(const kzg_ctx* ctx, double* ms_out, uint64_t* launches) try {
  if (!ctx || !ms_out || !launches) return fail(KZG_FAIL_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> guard(ctx->prof_lock);
  HIP_TRY(hipSetDevice(ctx->device));
  for (int k = 0; k < PROF_KINDS; k++) {
    ms_out[k] = 0;
    launches[k] = 0;
  }
  // Profiling must not overlap calls that are still being ENQUEUED on other threads (a pair whose second event is not yet
  // recorded cannot be read): such pairs are skipped, never fatal, and the interval always ends.
  ctx->profiling.store(false);
  for (size_t i = 0; i < ctx->prof_used; i++) {
    float ms = 0;
    const ProfEvent& pe = ctx->prof_events[i];
    if (hipEventSynchronize(pe.e1) != hipSuccess || hipEventElapsedTime(&ms, pe.e0, pe.e1) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    ms_out[pe.kind] += ms;
    launches[pe.kind]++;
  }
  ctx->prof_used = 0;
  return 0;
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_profile_end(const kzg_ctx* ctx, double* msm_ms_total, uint64_t* msm_launches) try {
  if (!ctx || !msm_ms_total || !msm_launches) return fail(KZG_FAIL_ARGUMENT, "null argument");
  double ms[PROF_KINDS];
  uint64_t cnt[PROF_KINDS];
  int32_t rc = kzg_profile_end_kinds(ctx, ms, cnt);
  if (rc) return rc;
  *msm_ms_total = ms[PROF_MSM_FIXED];
  *msm_launches = cnt[PROF_MSM_FIXED];
  return 0;
} catch (...) {
  return abi_exception();
}

// returns the event pair to record around the next launch of class `kind` (or nullptrs)
int32_t prof_next(const kzg_ctx* ctx, int kind, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  std::lock_guard<std::mutex> guard(ctx->prof_lock);
  if (!ctx->profiling.load()) return 0;
  if (ctx->prof_used == ctx->prof_events.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    ctx->prof_events.push_back(ProfEvent{kind, a, b});
  }
  ProfEvent& pe = ctx->prof_events[ctx->prof_used++];
  pe.kind = kind;
  *e0 = pe.e0;
  *e1 = pe.e1;
  return 0;
}

EnvKnobs read_env_knobs() {
  {
    EnvKnobs k;
    k.trace = getenv("KATETH_AMD_TRACE") != nullptr;
    if (const char* e = getenv("KATETH_AMD_HOST_FP")) host_fp_force_portable() = std::string(e) == "portable";  // measurement aid: the host's Fp products without mulx / adx
    if (const char* e = getenv("KATETH_AMD_PROOF_CHUNK")) k.proof_chunk = (uint64_t)atoll(e) > 0 ? (uint64_t)atoll(e) : 0;
    if (const char* e = getenv("KATETH_AMD_PROOF_OVERLAP")) k.proof_overlap = atoi(e) != 0;
    if (const char* e = getenv("KATETH_AMD_EVAL_GROUP")) k.eval_group = atoi(e);
    k.verify_serial = getenv("KATETH_AMD_VERIFY_SERIAL") != nullptr;
    k.single_via_batch = getenv("KATETH_AMD_SINGLE_VIA_BATCH") != nullptr;
    if (const char* e = getenv("KATETH_AMD_VAR_MSM")) k.var_msm_classic = std::string(e) == "classic";
    if (const char* e = getenv("KATETH_AMD_VAR_GLV")) k.var_glv = atoi(e) != 0;
    if (const char* e = getenv("KATETH_AMD_VAR_SEG")) k.var_seg = (uint32_t)std::max(0, atoi(e));
    if (const char* e = getenv("KATETH_AMD_VERIFY_STREAMS")) k.verify_streams = (uint32_t)atoi(e) <= (uint32_t)KZG_STAGE_STREAMS ? (uint32_t)atoi(e) : 0u;
    if (const char* e = getenv("KATETH_AMD_VERIFY_CHUNK")) k.verify_chunk = (uint64_t)atoll(e) > 0 ? (uint64_t)atoll(e) : 0;
    k.comb_full_wave = getenv("KATETH_AMD_COMB_FULL_WAVE") != nullptr;
    if (const char* e = getenv("KATETH_AMD_LAT_TABLE")) k.lat_table = atoi(e) != 0;
    if (const char* e = getenv("KATETH_AMD_COMB_FAIR")) k.comb_fair = (uint32_t)atoi(e) < 40u ? (uint32_t)atoi(e) : 0u;
    if (const char* e = getenv("KATETH_AMD_MSM_SPLITS")) {
      const int v = atoi(e);
      if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) k.msm_splits = (uint32_t)v;
    }
    if (const char* e = getenv("KATETH_AMD_CHALLENGE_SPLIT_MAX")) k.challenge_split_max = (uint64_t)atoll(e) > 0 ? (uint64_t)atoll(e) : 0;
    return k;
  }
}

// the table fields may be swapped by the background build (KZG_CFG_BUILD_ASYNC): read under the lock
extern "C" int32_t kzg_ctx_window_bits(const kzg_ctx* ctx) try {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return (int32_t)ctx->window_class;
} catch (...) {
  return abi_exception();
}
extern "C" uint64_t kzg_ctx_table_bytes(const kzg_ctx* ctx) {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return ctx->table_bytes;
}
extern "C" uint64_t kzg_ctx_workspace_bytes(const kzg_ctx* ctx, uint32_t slot) {
  if (!ctx || slot >= (uint32_t)KZG_WS_SLOTS) return 0;
  std::lock_guard<std::mutex> guard(ctx->lock);
  return ctx->wss[slot].bytes;
}
extern "C" uint32_t kzg_ctx_members(const kzg_ctx* ctx) { return ctx ? 1u + (uint32_t)ctx->peers.size() : 0u; }
extern "C" const kzg_ctx* kzg_ctx_member(const kzg_ctx* ctx, uint32_t k) {
  if (!ctx || k > ctx->peers.size()) return nullptr;
  return k == 0 ? ctx : ctx->peers[k - 1];
}
extern "C" int32_t kzg_ctx_member_device(const kzg_ctx* ctx, uint32_t k) try {
  const kzg_ctx* m = kzg_ctx_member(ctx, k);
  return m ? m->device : -1;
} catch (...) {
  return abi_exception();
}

extern "C" void kzg_ctx_destroy(kzg_ctx* ctx) {
  if (!ctx) return;
  for (kzg_ctx* p : ctx->peers) kzg_ctx_destroy(p);
  ctx->peers.clear();
  ctx->build_cancel.store(true);
  if (ctx->build_thread.joinable()) ctx->build_thread.join();
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (CombTables& t : ctx->retired) tables_free(t);
  if (ctx->d_table) (void)hipFree(ctx->d_table);
  if (ctx->d_table_lat) (void)hipFree(ctx->d_table_lat);
  if (ctx->d_bases_brp) (void)hipFree(ctx->d_bases_brp);
  if (ctx->d_roots_brp) (void)hipFree(ctx->d_roots_brp);
  if (ctx->d_eval_tab) (void)hipFree(ctx->d_eval_tab);
  if (ctx->d_gen_affine) (void)hipFree(ctx->d_gen_affine);
  if (ctx->d_comb_k) (void)hipFree(ctx->d_comb_k);
  if (ctx->d_comb_k_lat) (void)hipFree(ctx->d_comb_k_lat);
  if (ctx->msm_override && ctx->msm_override->destroy) ctx->msm_override->destroy(ctx);
  delete ctx->pairing;
  for (WsSlot& w : ctx->wss) {
    if (w.p) (void)hipFree(w.p);
    if (w.ev) (void)hipEventDestroy(w.ev);
  }
  if (ctx->d_clock_probe) (void)hipFree(ctx->d_clock_probe);
  if (ctx->probe_stream) (void)hipStreamDestroy(ctx->probe_stream);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (auto e : ctx->proof_events) (void)hipEventDestroy(e);
  for (auto& pe : ctx->prof_events) {
    (void)hipEventDestroy(pe.e0);
    (void)hipEventDestroy(pe.e1);
  }
  session_pool_clear(ctx);
  stage_destroy(ctx);
  delete ctx;
}


// temporary device allocations of a build: freed on every exit path
struct ScratchAllocs {
  std::vector<void*> ptrs;
  template <class T>
  hipError_t alloc(T** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
  ~ScratchAllocs() {
    for (void* p : ptrs) (void)hipFree(p);
  }
};

static void tables_free(CombTables& t) {
  if (t.d_table) (void)hipFree(t.d_table);
  if (t.d_table_lat) (void)hipFree(t.d_table_lat);
  if (t.d_comb_k) (void)hipFree(t.d_comb_k);
  if (t.d_comb_k_lat) (void)hipFree(t.d_comb_k_lat);
  t = CombTables{};
}
// Makes `t` the tables the context computes with.  Commitment and proof calls read them under ctx->lock; the tables they
// replace are kept until kzg_ctx_destroy (launches already enqueued hold their addresses).
static void tables_install(kzg_ctx* ctx, const CombTables& t) {
  std::lock_guard<std::mutex> guard(ctx->lock);
  if (ctx->d_table || ctx->d_table_lat || ctx->d_comb_k || ctx->d_comb_k_lat) {
    CombTables old;
    old.d_table = ctx->d_table;
    old.d_table_lat = ctx->d_table_lat;
    old.d_comb_k = ctx->d_comb_k;
    old.d_comb_k_lat = ctx->d_comb_k_lat;
    ctx->retired.push_back(old);
  }
  ctx->comb = t.comb;
  ctx->comb_lat = t.comb_lat;
  ctx->d_table = t.d_table;
  ctx->d_table_lat = t.d_table_lat;
  ctx->d_comb_k = t.d_comb_k;
  ctx->d_comb_k_lat = t.d_comb_k_lat;
  ctx->table_bytes = t.table_bytes;
  ctx->window_class = t.window_class;
}

// ---- comb table (msm_comb.cuh): G groups x 64 chunks x ep64 subset sums, built a few chunks at a time through an XYZZ
// staging buffer and the batch normaliser of the window table.  Everything runs on `st` (the background build of
// KZG_CFG_BUILD_ASYNC has a stream of its own and never touches the null stream); `cancel` is polled between passes. ----
static int32_t comb_build_table(const kzg_ctx* ctx, const CombGeom& cg, uint4** out_table, ScratchAllocs& scratch, TraceTimer& tt, hipStream_t st,
                                const std::atomic<bool>* cancel) {
  uint4* d_table = nullptr;
  HIP_TRY(hipMalloc(&d_table, comb_table_entries(cg) * 96));
  *out_table = d_table;  // owned by the caller's CombTables from here on
  tt.mark("table allocation");
  uint4 *d_B = nullptr, *d_D = nullptr;
  HIP_TRY(scratch.alloc(&d_B, (size_t)cg.G * 4096 * 96));
  HIP_TRY(scratch.alloc(&d_D, (size_t)cg.G * 4096 * 96));
  hipLaunchKernelGGL(k_comb_bases, dim3(64), dim3(64), 0, st, ctx->d_bases_brp, cg, d_B, d_D);
  HIP_TRY(hipGetLastError());
  uint32_t min_t = 64;
  for (uint32_t r = 0; r < cg.nb; r++) min_t = comb_tbits(cg.nb, r) < min_t ? comb_tbits(cg.nb, r) : min_t;
  const uint32_t sl = (min_t - 1 < 9) ? min_t - 1 : 9;  // segment = 2^sl entries per thread
  uint32_t nq = 64;                                       // chunks per pass: staging buffer <= ~13 GB
  while (nq > 1 && (uint64_t)nq * cg.ep64 * sizeof(g1_xyzz) > (13ull << 30)) nq >>= 1;
  g1_xyzz* d_tmp = nullptr;
  HIP_TRY(scratch.alloc(&d_tmp, (size_t)nq * cg.ep64 * sizeof(g1_xyzz)));
  uint32_t* d_inf_seen = nullptr;
  HIP_TRY(scratch.alloc(&d_inf_seen, sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(d_inf_seen, 0, sizeof(uint32_t), st));
  for (uint32_t grp = 0; grp < cg.G; grp++) {
    for (uint32_t q0 = 0; q0 < 64; q0 += nq) {
      if (cancel && cancel->load()) {
        (void)hipStreamSynchronize(st);
        return fail(KZG_FAIL_ARGUMENT, "table build cancelled (context destroyed)");
      }
      const uint64_t threads = (uint64_t)nq * (cg.ep64 >> sl);
      hipLaunchKernelGGL(k_comb_chain, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, d_B, d_D, grp, q0, nq, cg, sl, d_tmp);
      constexpr int KN = 8;
      const uint64_t count = (uint64_t)nq * cg.ep64;
      const uint64_t nthreads = (count + KN - 1) / KN;
      hipLaunchKernelGGL(k_table_normalize<KN>, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, st, d_tmp, count, d_table,
                         (uint64_t)grp * cg.epg + (uint64_t)q0 * cg.ep64, TABLE_FMT_PACKED30, d_inf_seen);
      HIP_TRY(hipGetLastError());
      if (cancel) HIP_TRY(hipStreamSynchronize(st));  // background build: a pass at a time, so that a cancel is seen within one pass
    }
  }
  uint32_t inf_seen = 0;
  HIP_TRY(hipMemcpyAsync(&inf_seen, d_inf_seen, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  tt.mark("comb table build kernels");
  // A subset sum sum_p +-L_p over a block of consecutive setup points is the identity (e.g. repeated or opposite points):
  // the table cannot hold it (affine entries).  The reference's P1::lincomb would accept such a setup; the ceremony file and
  // any setup of independent points cannot produce one.  Rejected loudly rather than committed to wrongly.
  if (inf_seen)
    return fail(KZG_FAIL_SETUP_UNSUPPORTED, "degenerate setup: a +-1 combination of consecutive g1_lagrange points is the point at infinity "
                                   "(repeated / opposite points); the fixed-base comb table cannot represent it");
  return 0;
}

// class -> blocks per 64 points: 22: 22 + 21 + 21; 16..21: 4 x 16; 8..15: 8 x 8; 4..7: 16 x 4 (the small classes keep test contexts cheap)
static inline uint32_t class_blocks(uint32_t c) { return c >= 22 ? 3u : (c >= 16 ? 4u : (c >= 8 ? 8u : 16u)); }

// Tables of class c with G plane groups over the context's (already decoded) setup points.  On failure nothing is left allocated.
static int32_t comb_build_tables(const kzg_ctx* ctx, uint32_t c, uint32_t G, CombTables& t, hipStream_t st, const std::atomic<bool>* cancel) {
  TraceTimer tt(ctx->knobs.trace, cancel ? "comb_build (background)" : "comb_build");
  ScratchAllocs scratch;
  t = CombTables{};
  const uint32_t nb = class_blocks(c);
  t.comb = comb_make_geom(nb, G);
  t.comb.fair = ctx->knobs.comb_fair;
  t.window_class = nb == 3 ? 22u : 64u / nb;
  t.table_bytes = comb_table_entries(t.comb) * 96;
  int32_t rc = comb_build_table(ctx, t.comb, &t.d_table, scratch, tt, st, cancel);
  if (rc == 0 && nb == 3 && ctx->knobs.lat_table) {  // class 22: the latency comb beside it
    t.comb_lat = comb_make_geom(8, 64);
    t.comb_lat.fair = t.comb.fair;
    rc = comb_build_table(ctx, t.comb_lat, &t.d_table_lat, scratch, tt, st, cancel);
  }
  // K = [c0] S with S = sum of the setup points, summed on the device.  For a Lagrange basis S is the G1 generator (the
  // basis sums to one), but Setup::load_json (src/kzg/setup.rs:46-82) accepts any in-group points and P1::lincomb is
  // right for all of them, so nothing here assumes it.  The ladder runs on the host (255 doublings + additions, once).
  auto constant_terms = [&]() -> int32_t {
    uint4* d_sum = nullptr;
    uint32_t* d_sum_inf = nullptr;
    HIP_TRY(scratch.alloc(&d_sum, 96));
    HIP_TRY(scratch.alloc(&d_sum_inf, sizeof(uint32_t)));
    hipLaunchKernelGGL(k_setup_sum_bases, dim3(1), dim3(64), 0, st, ctx->d_bases_brp, d_sum, d_sum_inf);
    HIP_TRY(hipGetLastError());
    uint32_t h[24], sum_inf = 0;
    HIP_TRY(hipMemcpyAsync(h, d_sum, 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&sum_inf, d_sum_inf, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sum_inf) return 0;  // S = O: K = O, no lane starts from it (d_comb_k stays null)
    fp_t x, y;
    for (int q = 0; q < 12; q++) {
      x.v[q] = h[q];
      y.v[q] = h[12 + q];
    }
    // One lane per blob STARTS from the constant term (k_msm_comb30).  That lane doubles its accumulator H - 1 times on
    // its way down the planes, so it is given [c0 / 2^(H-1)] S: one point per table geometry (main comb, latency comb).
    auto constant_for = [&](uint32_t H, uint4** d_out) -> int32_t {
      const uint32_t c0p[8] = KZG_FR_COMB_C0_PLAIN;
      fr_t c0, two, pw, inv, k;
      for (int q = 0; q < 8; q++) c0.v[q] = c0p[q];
      to_mont<FrParams>(c0, c0);
      two = fr_one();
      add_mod<FrParams>(two, two, two);
      pw = fr_one();
      for (uint32_t i = 0; i + 1 < H; i++) fr_mul(pw, pw, two);  // 2^(H-1)
      fr_inv(inv, pw);
      fr_mul(k, c0, inv);
      from_mont<FrParams>(k, k);  // plain scalar c0 / 2^(H-1) mod r
      g1_xyzz acc;
      xyzz_set_inf(acc);
      for (int bit = 255; bit >= 0; bit--) {
        xyzz_dbl(acc);
        if ((k.v[bit >> 5] >> (bit & 31)) & 1u) xyzz_madd(acc, x, y);
      }
      fp_t kx, ky;
      if (!xyzz_to_affine(kx, ky, acc)) return 0;  // the identity: nothing to start from
      uint32_t hk[24];  // the table's format (fp30.cuh: packed centred 30-bit digits of x * 2^390): k_msm_comb30 loads K like an entry
      fp_to_packed30(hk, kx);
      fp_to_packed30(hk + 12, ky);
      HIP_TRY(hipMalloc(d_out, 96));
      HIP_TRY(hipMemcpyAsync(*d_out, hk, 96, hipMemcpyHostToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));  // hk is a stack buffer
      return 0;
    };
    int32_t rck = constant_for(t.comb.H, &t.d_comb_k);
    if (rck == 0 && t.d_table_lat) rck = constant_for(t.comb_lat.H, &t.d_comb_k_lat);
    return rck;
  };
  if (rc == 0) rc = constant_terms();
  if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = fail(KZG_FAIL_HIP, "table build: synchronize failed");
  tt.mark("comb constant term");
  if (rc) {
    const ErrorSnapshot keep = error_snapshot();
    (void)hipStreamSynchronize(st);
    tables_free(t);
    error_publish(keep);
  }
  return rc;
}

static int32_t ctx_build(kzg_ctx* ctx, const uint8_t* g1_lagrange, const uint8_t* g2_monomial, uint32_t c, uint32_t G) {
  TraceTimer tt(ctx->knobs.trace, "ctx_build");
  ScratchAllocs scratch;
  hipStream_t st = nullptr;
  // ---- G2 monomial points (host): P2::decompress of all 65 (src/kzg/setup.rs:67-72) ----
  {
    host::g2_affine tau{};
    for (int i = 0; i < KZG_SETUP_G2_POINTS; i++) {
      host::g2_affine q;
      int32_t stq = host::g2_decompress(q, g2_monomial + 96 * i);
      if (stq != 0) return fail_detail(KZG_FAIL_SETUP_G2, stq, "g2_monomial[" + std::to_string(i) + "] rejected, code " + std::to_string(stq));
      if (i == 1) tau = q;
    }
    ctx->pairing = new host::pairing_ctx();
    ctx->pairing->fc = host::make_frob_consts();
    host::g2_affine gen;
    {
      const uint32_t x0[12] = KZG_FP_G2X0_MONT, x1[12] = KZG_FP_G2X1_MONT, y0[12] = KZG_FP_G2Y0_MONT, y1[12] = KZG_FP_G2Y1_MONT;
      for (int q = 0; q < 12; q++) {
        gen.x.c0.v[q] = x0[q];
        gen.x.c1.v[q] = x1[q];
        gen.y.c0.v[q] = y0[q];
        gen.y.c1.v[q] = y1[q];
      }
      gen.inf = false;
    }
    ctx->pairing->lines_g2 = host::precompute_lines(gen);
    ctx->pairing->lines_tau = host::precompute_lines(tau);
  }
  tt.mark("G2 decode + Miller lines (host)");
  // ---- G1 generator (BLS12_381_G1, src/bls.rs:391) ----
  {
    const uint32_t gx[12] = KZG_FP_G1X_R392, gy[12] = KZG_FP_G1Y_R392;  // operand of the variable-base MSM: 2^392 domain
    uint32_t h[24];
    for (int q = 0; q < 12; q++) {
      h[q] = gx[q];
      h[12 + q] = gy[q];
    }
    const uint32_t px[12] = KZG_FP_G1PHIX_R392, py[12] = KZG_FP_G1PHIY_R392;  // [z^2]G = (beta Gx, -Gy): the generator term's second GLV point
    uint32_t h2[48];
    for (int q = 0; q < 24; q++) h2[q] = h[q];
    for (int q = 0; q < 12; q++) {
      h2[24 + q] = px[q];
      h2[36 + q] = py[q];
    }
    HIP_TRY(hipMalloc(&ctx->d_gen_affine, 192));
    HIP_TRY(hipMemcpy(ctx->d_gen_affine, h2, 192, hipMemcpyHostToDevice));
  }
  // ---- G1 Lagrange points: decompress, subgroup check, BRP -----------------
  uint8_t* d_in = nullptr;
  int32_t* d_status = nullptr;
  HIP_TRY(scratch.alloc(&d_in, 4096 * 48));
  HIP_TRY(scratch.alloc(&d_status, 4096 * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(d_in, g1_lagrange, 4096 * 48, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&ctx->d_bases_brp, 4096 * 96));
  hipLaunchKernelGGL(k_setup_g1, dim3(64), dim3(64), 0, st, d_in, ctx->d_bases_brp, d_status);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> h_status(4096);
  HIP_TRY(hipMemcpy(h_status.data(), d_status, 4096 * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < 4096; i++)
    if (h_status[i] != 0)
      return fail_detail(KZG_FAIL_SETUP_G1, h_status[i] == 100 ? 0 : h_status[i], "g1_lagrange[" + std::to_string(i) + "] rejected, code " + std::to_string(h_status[i]) +
                                         (h_status[i] == 100 ? " (point at infinity is not supported as a setup base)" : ""));
  tt.mark("G1 decode");
  // ---- roots of unity -------------------------------------------------------
  HIP_TRY(hipMalloc(&ctx->d_roots_brp, 4096 * sizeof(fr_t)));
  hipLaunchKernelGGL(k_setup_roots, dim3(64), dim3(64), 0, st, ctx->d_roots_brp);
  HIP_TRY(hipMalloc(&ctx->d_eval_tab, (size_t)EVAL_TAB_HEXES * EVAL_TAB_DWORDS * sizeof(uint32_t)));
  hipLaunchKernelGGL(k_setup_eval_tab, dim3(EVAL_TAB_HEXES / 64), dim3(64), 0, st, ctx->d_roots_brp, ctx->d_eval_tab);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  tt.mark("roots + evaluation table");
  if (ctx->msm_override && ctx->msm_override->build) {  // test-only library (tests/window_msm)
    int32_t rco = ctx->msm_override->build(ctx);
    if (rco) return rco;
  }
  if (ctx->use_comb) {
    CombTables t;
    int32_t rcc = comb_build_tables(ctx, c, G, t, st, nullptr);
    if (rcc) return rcc;
    tables_install(ctx, t);
  }
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

// one attempt at a context with table class c (>= 4) and G plane groups
static int32_t ctx_create_with(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, int device, uint32_t c, uint32_t G, kzg_ctx** out) {
  kzg_ctx* ctx = new (std::nothrow) kzg_ctx();
  if (!ctx) return fail(KZG_FAIL_ARGUMENT, "out of host memory");
  ctx->device = device;
  const uint32_t nb = class_blocks(c);
  ctx->knobs = read_env_knobs();
  ctx->comb = comb_make_geom(nb, G);
  ctx->comb.fair = ctx->knobs.comb_fair;
  ctx->window_class = nb == 3 ? 22u : 64u / nb;
  if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return fail(KZG_FAIL_HIP, "hipStreamCreate failed");
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = (uint32_t)prop.multiProcessorCount;
  if (g_msm_override_hook) ctx->msm_override = g_msm_override_hook(ctx, c);  // test-only library (tests/window_msm), never the product
  int32_t rc = ctx_build(ctx, g1_lagrange, g2_monomial, c, G);
  if (rc != 0) {
    const ErrorSnapshot keep = error_snapshot();
    kzg_ctx_destroy(ctx);
    (void)hipGetLastError();  // a failed allocation leaves its code behind: the NEXT call's launch check must not trip over it
    error_publish(keep);
    return rc;
  }
  // every unit's code object in, now -- with KZG_CFG_BUILD_ASYNC BEFORE the background thread's allocation holds the runtime
  // (engine_internal.hpp); this unit's own was loaded by the setup kernels above
  warm_code_object_blob();
  warm_code_object_proof();
  warm_code_object_verify();
  *out = ctx;
  return 0;
}

// KZG_CFG_BUILD_ASYNC: the chosen table, built beside the caller's first calls (which run on the first-use table) and
// swapped in under ctx->lock.  An automatic choice whose allocation fails steps down the ladder; if nothing larger than the
// first-use table can be built the context simply stays on it (kzg_ctx_wait_ready reports the code).
static void build_thread_main(kzg_ctx* ctx, std::vector<TableChoice> ladder, bool automatic) {
  int32_t rc = 0;
  hipStream_t st = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
    rc = fail(KZG_FAIL_HIP, "background table build: stream creation failed");
  if (rc == 0) {
    for (const TableChoice& ch : ladder) {
      if (class_blocks(ch.c) == ctx->comb.nb && ch.G == ctx->comb.G) {  // what already serves: nothing larger could be built
        break;
      }
      CombTables t;
      rc = comb_build_tables(ctx, ch.c, ch.G, t, st, &ctx->build_cancel);
      if (rc == 0) {
        tables_install(ctx, t);
        break;
      }
      if (!automatic || rc != KZG_FAIL_HIP || ctx->build_cancel.load()) break;
      (void)hipGetLastError();  // an allocation failed: clear it and step down
    }
  }
  if (st) (void)hipStreamDestroy(st);
  if (rc) (void)hipGetLastError();  // (the last-error slot is per host thread; cleared all the same)
  {
    std::lock_guard<std::mutex> guard(ctx->build_mu);
    ctx->build_rc = rc;
    if (rc) ctx->build_err = error_snapshot();
    ctx->build_running = false;
  }
  ctx->build_cv.notify_all();
}

extern "C" int32_t kzg_ctx_ready(const kzg_ctx* ctx) try {
  if (!ctx) return 0;
  {
    std::lock_guard<std::mutex> guard(ctx->build_mu);
    if (ctx->build_running) return 0;
  }
  for (const kzg_ctx* p : ctx->peers)
    if (!kzg_ctx_ready(p)) return 0;
  return 1;
} catch (...) {
  return abi_exception();
}
extern "C" int32_t kzg_ctx_wait_ready(const kzg_ctx* ctx) try {
  if (!ctx) return fail(KZG_FAIL_ARGUMENT, "null argument");
  int32_t rc = 0;
  {
    std::unique_lock<std::mutex> guard(ctx->build_mu);
    ctx->build_cv.wait(guard, [&] { return !ctx->build_running; });
    rc = ctx->build_rc;
    if (rc) error_publish(ctx->build_err);
  }
  for (const kzg_ctx* p : ctx->peers) {
    const int32_t rp = kzg_ctx_wait_ready(p);
    if (rc == 0) rc = rp;
  }
  return rc;
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_device_count(void) try {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(KZG_FAIL_NO_DEVICE, "no HIP device visible: the kateth_amd engine has no CPU fallback");
  return ndev;
} catch (...) {
  return abi_exception();
}

// A single-device context on `device` (cfg's device / devices / ndev are not read here).
int32_t ctx_create_single(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, int device, kzg_ctx** out) {
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  // Table class (index bits per lookup = points per block of the comb).  window_bits = 0 -- what Setup::load_json's drop-in
  // passes, INTEGRATION.md section 5 -- takes the fastest class that is within the caller's budget and that the device has
  // room for RIGHT NOW (include/kateth_amd.h, kzg_config):
  //   class 22 (blocks of 22 + 21 + 21 points, 49,152 additions per blob), G = 8 plane groups = 192 GiB  if >= 232 GiB are free
  //   class 22, G = 4 = 96 GiB (63 instead of 31 Horner doublings per lane: -2 %)                         if >= 136 GiB
  //   class 16 (blocks of 16, G = 16, 65,536 additions per blob: -25 %) = 12.9 GB                         if >=  21 GiB
  //   class 8  (blocks of 8, G = 16) = 100 MB                                                             otherwise
  // (the margins cover the build's 13 GB of staging, the 0.4-GB latency comb, the caller's blobs and the call workspace: ONE slot --
  // at most 7.5 GiB, a proof call in chunks of 16,384 blobs -- for a caller that runs one call at a time; callers that keep calls
  // in flight get up to three slots while the device has room and queue behind a large-enough slot when it has not: ws_reserve).
  // The default budget is 100 GiB: an unconfigured context stops at the 96-GiB table and leaves a 288-GB part two thirds free;
  // KZG_CFG_TABLE_MAX or an explicit table_budget_bytes moves the cap.  If the allocation of an automatically
  // chosen table fails all the same (fragmentation, another process), the next smaller choice is tried.  Precedence: a
  // non-zero field of kzg_config beats the environment (KATETH_AMD_WINDOW_BITS / KATETH_AMD_COMB_GROUPS), which beats the
  // automatic choice; an explicit class is honoured as given (and fails if it cannot be built).
  constexpr size_t GiB = (size_t)1 << 30;
  constexpr size_t GROUP22_BYTES = (size_t)64 * ((size_t)1 << 22) * 96;  // one plane group of class 22: 24 GiB
  constexpr size_t CLASS16_BYTES = (size_t)16 * 64 * ((size_t)4 << 15) * 96;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  const int32_t flags = cfg ? cfg->flags : 0;
  if (flags & ~(KZG_CFG_TABLE_MAX | KZG_CFG_BUILD_ASYNC)) return fail(KZG_FAIL_ARGUMENT, "unknown kzg_config.flags bit");
  if (cfg && cfg->reserved) return fail(KZG_FAIL_ARGUMENT, "kzg_config.reserved must be 0");
  uint32_t want = cfg ? (uint32_t)cfg->window_bits : 0u;
  if (want == 0)
    if (const char* e = getenv("KATETH_AMD_WINDOW_BITS")) want = (uint32_t)atoi(e);
  if (want != 0 && (want < 4 || want > 22)) return fail(KZG_FAIL_ARGUMENT, "window_bits must be 0 (automatic) or in [4,22]");
  uint32_t G_fixed = (cfg && cfg->plane_groups) ? (uint32_t)cfg->plane_groups : 0u;
  if (G_fixed == 0)
    if (const char* e = getenv("KATETH_AMD_COMB_GROUPS")) G_fixed = (uint32_t)atoi(e);
  if (G_fixed && !(G_fixed == 1 || G_fixed == 2 || G_fixed == 4 || G_fixed == 8 || G_fixed == 16))
    return fail(KZG_FAIL_ARGUMENT, "plane groups must be 1, 2, 4, 8 or 16");
  size_t budget = cfg ? (size_t)cfg->table_budget_bytes : 0;
  if (budget == 0 && !(flags & KZG_CFG_TABLE_MAX))
    if (const char* e = getenv("KATETH_AMD_TABLE_BUDGET_GIB"))  // an operator's cap on an unconfigured drop-in (the same precedence rule: a field beats the variable)
      if (atof(e) > 0) budget = (size_t)(atof(e) * (double)GiB);
  if (budget == 0) budget = (flags & KZG_CFG_TABLE_MAX) ? ~(size_t)0 : 100 * GiB;
  // candidates (class, plane groups), fastest first
  const TableChoice full[4] = {{22, 8, 8 * GROUP22_BYTES + 40 * GiB}, {22, 4, 4 * GROUP22_BYTES + 40 * GiB}, {16, 16, 21 * GiB}, {8, 16, 0}};
  const size_t table_of[4] = {8 * GROUP22_BYTES, 4 * GROUP22_BYTES, CLASS16_BYTES, 0};
  std::vector<TableChoice> ladder;
  const bool automatic = want == 0;
  if (!automatic) {
    uint32_t G = want >= 22 ? (free_b >= full[0].need ? 8u : 4u) : 16u;  // the budget bounds the AUTOMATIC choice only
    if (G_fixed) G = G_fixed;
    ladder.push_back(TableChoice{want, G, 0});
  } else {
    for (int k = 0; k < 4; k++)
      if (free_b >= full[k].need && budget >= table_of[k]) ladder.push_back(TableChoice{full[k].c, G_fixed ? G_fixed : full[k].G, full[k].need});
  }
  if (ladder.empty()) ladder.push_back(TableChoice{8, G_fixed ? G_fixed : 16u, 0});
  // KZG_CFG_BUILD_ASYNC: stand up on the 100-MB class-8 table (the reference's load_json is 4,096 + 65 decompressions,
  // src/kzg/setup.rs:59-72 -- the context is usable after about as much work), build the chosen table beside the first calls
  if ((flags & KZG_CFG_BUILD_ASYNC) && !g_msm_override_hook && class_blocks(ladder[0].c) < 8u) {
    kzg_ctx* ctx = nullptr;
    hw_queues_note(getenv("KATETH_AMD_TRACE") != nullptr);
    if (getenv("KATETH_AMD_TRACE"))
      fprintf(stderr, "[kateth_amd trace] kzg_ctx_create device %d: first-use table class 8; class %u, %u plane groups follows in the background\n", device,
              ladder[0].c, ladder[0].G);
    int32_t rc = ctx_create_with(g1_lagrange, g2_monomial, device, 8, 16, &ctx);
    if (rc) return rc;
    {
      std::lock_guard<std::mutex> guard(ctx->build_mu);
      ctx->build_running = true;
    }
    try {
      ctx->build_thread = std::thread(build_thread_main, ctx, ladder, automatic);
    } catch (...) {  // no thread to be had: build in the caller's time after all
      build_thread_main(ctx, ladder, automatic);
    }
    *out = ctx;
    return 0;
  }
  const bool trace = getenv("KATETH_AMD_TRACE") != nullptr;
  hw_queues_note(trace);
  auto say = [&](const char* how, const TableChoice& ch) {  // the choice, once per context, for whoever asks (KATETH_AMD_TRACE=1)
    if (trace)
      fprintf(stderr, "[kateth_amd trace] kzg_ctx_create device %d: table class %u, %u plane groups (%s; %.1f GiB free, budget %s)\n", device, ch.c, ch.G, how,
              (double)free_b / (double)GiB, budget == ~(size_t)0 ? "none" : (std::to_string((double)budget / (double)GiB) + " GiB").c_str());
  };
  int32_t rc = fail(KZG_FAIL_HIP, "no table class fits");
  for (const TableChoice& ch : ladder) {
    say(automatic ? "automatic" : "as configured", ch);
    rc = ctx_create_with(g1_lagrange, g2_monomial, device, ch.c, ch.G, out);
    if (!automatic || rc != KZG_FAIL_HIP) return rc;  // built, or rejected for a reason a smaller table would not cure (bad setup point, ...)
    (void)hipGetLastError();                          // an allocation failed: clear it and step down
  }
  return rc;
}

// a caller compiled against another revision of the header passes another size: refused, never read past
static int32_t cfg_check(const kzg_config* cfg) {
  if (cfg && cfg->struct_size != (uint32_t)sizeof(kzg_config))
    return fail(KZG_FAIL_ARGUMENT, "kzg_config.struct_size is " + std::to_string(cfg->struct_size) + ", this library's kzg_config has " +
                                       std::to_string(sizeof(kzg_config)) + " bytes: initialise with KZG_CONFIG_INIT from the matching include/kateth_amd.h");
  return 0;
}

extern "C" int32_t kzg_ctx_create(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, kzg_ctx** out) try {
  if (!g1_lagrange || !g2_monomial || !out) return fail(KZG_FAIL_ARGUMENT, "null argument");
  *out = nullptr;
  if (int32_t rcc = cfg_check(cfg)) return rcc;
  const int32_t ndev = kzg_device_count();
  if (ndev < 0) return ndev;
  if (cfg && cfg->ndev != 0) return group_create(g1_lagrange, g2_monomial, cfg, out);  // engine_multi.hip
  const int device = cfg ? cfg->device : 0;
  if (device < 0 || device >= ndev) return fail(KZG_FAIL_ARGUMENT, "device ordinal out of range");
  return ctx_create_single(g1_lagrange, g2_monomial, cfg, device, out);
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_ctx_create_multi(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const int32_t* devices, uint32_t ndev,
                                        const kzg_config* cfg, kzg_ctx** out) try {
  kzg_config c = KZG_CONFIG_INIT;
  if (int32_t rcc = cfg_check(cfg)) return rcc;
  if (cfg) c = *cfg;
  c.devices = devices;
  c.ndev = ndev;
  if (ndev == 0) return fail(KZG_FAIL_ARGUMENT, "kzg_ctx_create_multi: ndev must not be 0 (KZG_ALL_DEVICES = every visible device)");
  return kzg_ctx_create(g1_lagrange, g2_monomial, &c, out);
} catch (...) {
  return abi_exception();
}

// ---------------------------------------------------------------------------
// fixed-base MSM launchers (declared in engine_internal.hpp; the kernels are compiled in this translation unit only)
// ---------------------------------------------------------------------------
// The fixed-base MSM alone over `n` scalar vectors on the device: 64 lane sums per (blob, split) unit into
// partials[unit * 64 + lane].  `scratch`: msm_scratch_bytes(ctx, n) bytes.  `scalars_consumed` (optional): recorded on `st` as
// soon as d_scalars is no longer read (after the bit-plane transposition; the MSM kernel reads the masks), so that a staging
// buffer can be refilled while the MSM runs.
int32_t msm_launch(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits, uint32_t lpb,
                   void* scratch, hipStream_t st, hipEvent_t scalars_consumed) {
  if (ctx->msm_override) {
    int32_t rco = ctx->msm_override->launch(ctx, be_bytes, d_scalars, n, d_status, partials, splits, lpb, scratch, st);
    if (rco == 0 && scalars_consumed) HIP_TRY(hipEventRecord(scalars_consumed, st));
    return rco;
  }
  uint64_t* masks = reinterpret_cast<uint64_t*>(scratch);
  {
    ProfScope ps(ctx, PROF_TRANSPOSE, st);
    if (be_bytes)
      hipLaunchKernelGGL((k_comb_transpose<true>), dim3((unsigned)(n * 8)), dim3(512), 0, st, d_scalars, n, masks, d_status);
    else
      hipLaunchKernelGGL((k_comb_transpose<false>), dim3((unsigned)(n * 8)), dim3(512), 0, st, d_scalars, n, masks, d_status);
  }
  if (scalars_consumed) HIP_TRY(hipEventRecord(scalars_consumed, st));
  ProfScope ps(ctx, PROF_MSM_FIXED, st);
  const bool lat = msm_uses_lat(ctx, splits);
  const uint4* table = lat ? ctx->d_table_lat : ctx->d_table;
  const CombGeom geom = lat ? ctx->comb_lat : ctx->comb;
  hipLaunchKernelGGL(k_msm_comb30<false>, dim3((unsigned)msm_units(n, splits, lpb)), dim3(64), 0, st, masks, n, splits, lpb, table, geom, partials,
                     (const uint4*)(lat ? ctx->d_comb_k_lat : ctx->d_comb_k), (uint64_t*)nullptr);
  HIP_TRY(hipGetLastError());
  return 0;
}
// Lane sums of n blobs -> 48-byte encodings.  Two tree stages: the 64 lane sums of every (blob, split) unit, then the units
// of a blob -- 6 + log2(splits) levels of latency instead of the splits + 5 a sequential walk over the splits costs (a
// single blob uses up to 256 units on the latency comb).  `partials` must have room for n * splits unit sums after the n * splits * 64 lane sums.
int32_t msm_finish(const kzg_ctx* ctx, uint64_t n, uint8_t* d_out48, uint8_t* d_out_affine96, const int32_t* d_status, g1_xyzz* partials,
                                 g1_xyzz* sums, uint32_t splits, uint32_t lpb, hipStream_t st) {
  ProfScope ps(ctx, PROF_REDUCE_COMPRESS, st);
  g1_xyzz* unit_sums = (splits == 1) ? sums : partials + (size_t)n * splits * 64;
  const uint64_t units = msm_units(n, splits, lpb);
  if (lpb == 32)  // half-wave mode: four blobs per reducing wave
    hipLaunchKernelGGL(k_msm_reduce_half4, dim3((unsigned)((units + 1) / 2)), dim3(64), 0, st, partials, units, unit_sums, n);
  else
    hipLaunchKernelGGL(k_msm_reduce, dim3((unsigned)units), dim3(64), 0, st, partials, units, lpb, unit_sums, units);
  if (splits > 64) {  // latency shape (a few blobs over up to 256 units each): tree + constant term + encoding in one launch
    hipLaunchKernelGGL((k_msm_reduce_splits<true>), dim3((unsigned)n), dim3(64), 0, st, unit_sums, splits, n, sums, d_status, d_out48, d_out_affine96);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (splits > 1)
    hipLaunchKernelGGL((k_msm_reduce_splits<false>), dim3((unsigned)n), dim3(64), 0, st, unit_sums, splits, n, sums, (const int32_t*)nullptr,
                       (uint8_t*)nullptr, (uint8_t*)nullptr);
  hipLaunchKernelGGL(k_g1_compress, dim3(blocks_for(n, 64)), dim3(64), 0, st, sums, n, d_status, d_out48, d_out_affine96);
  HIP_TRY(hipGetLastError());
  return 0;
}
// MSM + reduce + compress
int32_t msm_pipeline(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, uint8_t* d_out48, uint8_t* d_out_affine96, int32_t* d_status,
                            g1_xyzz* partials, g1_xyzz* sums, uint32_t splits, void* scratch, hipStream_t st) {
  const uint32_t lpb = msm_lanes_per_blob(ctx, n, splits);
  int32_t rc = msm_launch(ctx, be_bytes, d_scalars, n, d_status, partials, splits, lpb, scratch, st);
  if (rc) return rc;
  return msm_finish(ctx, n, d_out48, d_out_affine96, d_status, partials, sums, splits, lpb, st);
}

// ---------------------------------------------------------------------------
// blob_to_kzg_commitment
// ---------------------------------------------------------------------------
static int32_t commit_dev_locked(const kzg_ctx* ctx, const void* d_blobs, uint64_t n, void* d_out48, void* d_out_affine96, int32_t* d_status,
                                 hipStream_t st) {
  if (n == 0) return 0;
  const uint64_t chunk_max = 16384;  // bounds the lane-partial scratch (12 KiB per blob)
  const uint64_t cn = n < chunk_max ? n : chunk_max;
  const uint32_t splits = choose_splits(ctx, cn);
  const size_t partial_bytes = align_up((size_t)cn * splits * 65 * sizeof(g1_xyzz), 256);  // 64 lane sums + 1 unit sum per (blob, split)
  const size_t sums_bytes = align_up((size_t)cn * sizeof(g1_xyzz), 256);
  const size_t need = partial_bytes + sums_bytes + msm_scratch_bytes(ctx, cn);
  int32_t rc = ws_reserve(ctx, need, st);
  if (rc) return rc;
  g1_xyzz* partials = reinterpret_cast<g1_xyzz*>(ws_ptr(ctx));
  g1_xyzz* sums = reinterpret_cast<g1_xyzz*>(ws_ptr(ctx) + partial_bytes);
  void* msm_scratch = ws_ptr(ctx) + partial_bytes + sums_bytes;
  HIP_TRY(hipMemsetAsync(d_status, 0, n * sizeof(int32_t), st));
  for (uint64_t base = 0; base < n; base += cn) {
    const uint64_t m = (n - base < cn) ? (n - base) : cn;
    rc = msm_pipeline(ctx, true, reinterpret_cast<const uint8_t*>(d_blobs) + base * (uint64_t)KZG_BYTES_PER_BLOB, m,
                            d_out48 ? reinterpret_cast<uint8_t*>(d_out48) + base * 48 : nullptr,
                            d_out_affine96 ? reinterpret_cast<uint8_t*>(d_out_affine96) + base * 96 : nullptr, d_status + base, partials, sums,
                            splits, msm_scratch, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int32_t kzg_blob_to_commitment_batch_dev(const kzg_ctx* ctx, const void* d_blobs, uint64_t n, void* d_out48, void* d_status,
                                                    void* hip_stream) try {
  if (!ctx || (n && (!d_blobs || !d_out48 || !d_status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> guard(ctx->lock);
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  WsCall ws(ctx, st);
  int32_t rc = ws.begin();
  if (rc == 0) rc = commit_dev_locked(ctx, d_blobs, n, d_out48, nullptr, reinterpret_cast<int32_t*>(d_status), st);
  if (rc == 0) rc = ws.end();
  return rc;
} catch (...) {
  return abi_exception();
}

// Host-buffer entry point.  The blobs cross PCIe in chunks through two device staging buffers: while the MSM of chunk k runs
// on one compute stream, chunk k+1 is copied in on the copy stream and then runs on the OTHER compute stream, so the
// transfer (56-57 GB/s: 9.4 ms per 4,096 blobs) hides behind the MSM and only the first chunk's copy is exposed.
//
// Chunk plan: a RAMP.  The exposed copy should be short, but a small chunk is an inefficient MSM launch (it needs split
// units to fill the chip, and every split repeats a lane's Horner doublings: 31 per 192 additions at 512 blobs, per 1,536 in
// half-wave mode), and a chunk's copy must finish inside the previous chunk's MSM (a blob copies in 2.3 us and commits in
// 7.3 us: the next chunk may be up to 3x as large).  So chunks grow 512, 512, 1,024, 2,048, then 4,096 (half-wave mode) --
// every one a launch that fills the chip once or an exact number of times, each with the splits choose_splits gives its size.
// Every chunk is its own group (its lane-sum trees and encoding follow its MSM on its stream, beside the next chunk's MSM).
// The staging buffer of a slot is free again as soon as the chunk's bit-plane transposition has read it.
// Measured at n = 4,096 on one box: 4 equal chunks 36.0 ms (profiles/r03/host_commit_timeline_equal_chunks.txt), ramp: see
// DESIGN.md section 7.
static std::vector<uint64_t> commit_host_plan(uint64_t n) {
  std::vector<uint64_t> plan;
  if (n <= 768) {
    plan.push_back(n);
    return plan;
  }
  // the ramp as far as it fits, then half-wave chunks of 4,096; a remainder below 256 blobs joins the first chunk (three
  // splits still run 682 blobs in one round), a longer one is the last chunk with a shape of its own
  const uint64_t ramp[4] = {512, 512, 1024, 2048};
  uint64_t rest = n;
  for (uint64_t r : ramp) {
    if (rest < r) break;
    plan.push_back(r);
    rest -= r;
  }
  if (plan.size() == 4)
    for (; rest >= 4096; rest -= 4096) plan.push_back(4096);
  if (rest >= 256)
    plan.push_back(rest);
  else
    plan[0] += rest;
  return plan;
}

int32_t commit_host(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, uint8_t* out_affine96, int32_t* status) {
  if (!ctx || (n && (!blobs || (!out48 && !out_affine96) || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(ctx->device));
  const std::vector<uint64_t> plan = commit_host_plan(n);
  uint64_t max_chunk = 0;
  for (uint64_t m : plan) max_chunk = m > max_chunk ? m : max_chunk;
  // Pooled resources: the staging arena, the result buffers, the streams and the events belong to the context
  // (stage_lock), so a steady-state call allocates nothing; the streams are ordered by events only -- the host never waits
  // inside the loop.
  std::lock_guard<std::mutex> stage_guard(ctx->stage_lock);
  int32_t rc = stage_init(ctx);
  if (rc) return rc;
  const uint64_t nslots = plan.size() > 1 ? 2 : 1;
  const size_t slot_bytes = (size_t)max_chunk * KZG_BYTES_PER_BLOB;
  const size_t out_bytes = align_up((size_t)n * (out48 ? 48 : 96), 256);
  rc = stage_reserve(ctx, nslots * slot_bytes, out_bytes + (size_t)n * sizeof(int32_t));
  if (rc) return rc;
  uint8_t* d_res = ctx->hostio;
  int32_t* d_status = reinterpret_cast<int32_t*>(ctx->hostio + out_bytes);
  // A call of ONE chunk (single blobs -- the drop-in's common case -- up to a few thousand) rides on one stream, copy included:
  // every event between two streams is a packet on one hardware queue waiting for a signal from another (~0.1 ms each with a
  // queue per stream), and there is nothing to overlap.
  const bool one_chunk = plan.size() == 1;
  hipStream_t comp[2] = {ctx->stage_streams[0], ctx->stage_streams[1]};
  hipStream_t copy_st = one_chunk ? comp[0] : ctx->stage_copy_stream;
  std::lock_guard<std::mutex> guard(ctx->lock);  // the workspace: per slot the lane sums, the sums and the bit-plane masks of a chunk
  WsCall ws(ctx, comp[0]);
  uint64_t max_units = 1;  // launch shapes follow the table in use, which only changes under this lock
  for (uint64_t m : plan) {
    const uint32_t sp = choose_splits(ctx, m);
    const uint64_t units = msm_units(m, sp, msm_lanes_per_blob(ctx, m, sp));
    max_units = units > max_units ? units : max_units;
  }
  do {
    const size_t partial_bytes = align_up((size_t)max_units * 65 * sizeof(g1_xyzz), 256);
    const size_t sums_bytes = align_up((size_t)max_chunk * sizeof(g1_xyzz), 256);
    const size_t scratch_bytes = align_up(msm_scratch_bytes(ctx, max_chunk), 256);
    const size_t per_slot = partial_bytes + sums_bytes + scratch_bytes;
    rc = ws.begin();
    if (rc == 0) rc = ws_reserve(ctx, nslots * per_slot, comp[0]);
    if (rc == 0 && !one_chunk) rc = ws_wait(ctx, comp[1]);
    if (rc) break;
    if (hipMemsetAsync(d_status, 0, n * sizeof(int32_t), comp[0]) != hipSuccess ||
        (!one_chunk && (hipEventRecord(ctx->stage_join[0], comp[0]) != hipSuccess || hipStreamWaitEvent(comp[1], ctx->stage_join[0], 0) != hipSuccess))) {
      rc = fail(KZG_FAIL_HIP, "memset failed");
      break;
    }
    uint64_t base = 0;
    for (size_t k = 0; k < plan.size() && rc == 0; base += plan[k], k++) {
      const int slot = (int)(k & 1);  // staging slot = workspace slot = compute stream
      const uint64_t m = plan[k];
      const uint32_t splits = choose_splits(ctx, m);
      const uint32_t lpb = msm_lanes_per_blob(ctx, m, splits);
      uint8_t* wslot = ws_ptr(ctx) + (size_t)slot * per_slot;
      g1_xyzz* partials = reinterpret_cast<g1_xyzz*>(wslot);
      g1_xyzz* sums = reinterpret_cast<g1_xyzz*>(wslot + partial_bytes);
      // chunk k-2 (same slot) must have been transposed out of the staging buffer before it is overwritten
      if (k >= 2 && hipStreamWaitEvent(copy_st, ctx->stage_done[slot], 0) != hipSuccess) {
        rc = fail(KZG_FAIL_HIP, "stream wait failed");
        break;
      }
      if (hipMemcpyAsync(ctx->stage + (size_t)slot * slot_bytes, blobs + base * (size_t)KZG_BYTES_PER_BLOB, m * (size_t)KZG_BYTES_PER_BLOB,
                         hipMemcpyHostToDevice, copy_st) != hipSuccess ||
          (!one_chunk && (hipEventRecord(ctx->stage_copied[slot], copy_st) != hipSuccess ||
                          hipStreamWaitEvent(comp[slot], ctx->stage_copied[slot], 0) != hipSuccess))) {
        rc = fail(KZG_FAIL_HIP, "host-to-device copy failed");
        break;
      }
      rc = msm_launch(ctx, true, ctx->stage + (size_t)slot * slot_bytes, m, d_status + base, partials, splits, lpb, wslot + partial_bytes + sums_bytes,
                            comp[slot], ctx->stage_done[slot]);
      if (rc == 0)
        rc = msm_finish(ctx, m, out48 ? d_res + base * 48 : nullptr, out_affine96 ? d_res + base * 96 : nullptr, d_status + base, partials, sums, splits, lpb,
                        comp[slot]);
    }
    if (rc) break;
    if (!one_chunk && (hipEventRecord(ctx->stage_join[1], comp[1]) != hipSuccess || hipStreamWaitEvent(comp[0], ctx->stage_join[1], 0) != hipSuccess)) {
      rc = fail(KZG_FAIL_HIP, "stream join failed");
      break;
    }
    rc = ws.end();
    if (rc) break;
    if (hipMemcpyAsync(out48 ? out48 : out_affine96, d_res, (size_t)n * (out48 ? 48 : 96), hipMemcpyDeviceToHost, comp[0]) != hipSuccess ||
        hipMemcpyAsync(status, d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, comp[0]) != hipSuccess ||
        hipStreamSynchronize(comp[0]) != hipSuccess)
      rc = fail(KZG_FAIL_HIP, "device-to-host copy failed");
  } while (0);
  if (rc) (void)hipDeviceSynchronize();
  return rc;
}

extern "C" int32_t kzg_blob_to_commitment_batch(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, int32_t* status) try {
  if (n && !out48) return fail(KZG_FAIL_ARGUMENT, "null argument");
  if (is_group(ctx)) return multi_commit(ctx, blobs, n, out48, nullptr, status);
  return commit_host(ctx, blobs, n, out48, nullptr, status);
} catch (...) {
  return abi_exception();
}
extern "C" int32_t kzg_blob_to_commitment_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out_affine96, int32_t* status) try {
  if (n && !out_affine96) return fail(KZG_FAIL_ARGUMENT, "null argument");
  if (is_group(ctx)) return multi_commit(ctx, blobs, n, nullptr, out_affine96, status);
  return commit_host(ctx, blobs, n, nullptr, out_affine96, status);
} catch (...) {
  return abi_exception();
}

// ---------------------------------------------------------------------------
// synthetic blobs + micro-benchmarks
// ---------------------------------------------------------------------------
extern "C" int32_t kzg_synth_blobs_dev(const kzg_ctx* ctx, uint64_t seed, uint64_t first_index, uint64_t n, void* d_blobs, void* hip_stream) try {
  if (!ctx || (n && !d_blobs)) return fail(KZG_FAIL_ARGUMENT, "null argument");
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(ctx->device));
  launch_synth_blobs(reinterpret_cast<hipStream_t>(hip_stream), seed, first_index, n, reinterpret_cast<uint8_t*>(d_blobs));
  HIP_TRY(hipGetLastError());
  return 0;
} catch (...) {
  return abi_exception();
}

// a chain of dependent products with the multiply of the radix-2^28 MSM kernel (fp28.cuh)
__global__ __launch_bounds__(256) void k_microbench_fp28_mul(uint32_t* out, uint64_t iters) {
  fp28 a, b;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int q = 0; q < F28_N; q++) {
    a.l[q] = f28_one_limb(q) ^ (t & 0xffu);
    b.l[q] = f28_r384_limb(q);
  }
#pragma unroll 1
  for (uint64_t it = 0; it < iters; it++) {
    fp28 r;
    f28_mul(r, a, b);
    a = b;
    b = r;
  }
  uint32_t x = 0;
#pragma unroll
  for (int q = 0; q < F28_N; q++) x ^= b.l[q];
  out[t] = x;
}

extern "C" int32_t kzg_microbench_fp_mul(const kzg_ctx* ctx, uint64_t lanes, uint64_t iters, float* ms) try {
  if (!ctx || !ms || lanes == 0) return fail(KZG_FAIL_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  lanes = align_up(lanes, 256);
  uint32_t* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, lanes * sizeof(uint32_t)));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  auto kern = k_microbench_fp28_mul;  // the multiply the MSM kernel is built on
  hipLaunchKernelGGL(kern, dim3((unsigned)(lanes / 256)), dim3(256), 0, nullptr, d_out, (uint64_t)16);
  HIP_TRY(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL(kern, dim3((unsigned)(lanes / 256)), dim3(256), 0, nullptr, d_out, iters);
  HIP_TRY(hipEventRecord(e1, nullptr));
  HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  return 0;
} catch (...) {
  return abi_exception();
}

// ISSUE interval of the instruction the hot loops are made of (v_mad_u64_u32: 76-80 % of the MSM, evaluation and decoding
// streams) with `w` waves per SIMD on the whole chip, and the shader clock that load sustains: 8 INDEPENDENT chains per
// wave, so no dependent-latency effect is measured.  Occupancy by construction: ONE workgroup of 4 w waves per CU (it takes nearly
// all of the CU's LDS, so no second one joins it) -- the waves of a workgroup are resident together, w per SIMD, wherever the
// dispatcher puts the workgroup.  (Two 256-thread workgroups per CU, as this was first written, are co-resident only if the
// dispatcher pairs them: in a process with many queues it sometimes ran them one after the other and the figure halved.)
// bench.py prices SQ_INSTS_VALU with these two numbers (roofline.valu_issue).
__global__ __launch_bounds__(1024) void k_microbench_valu_issue(uint32_t* out, unsigned long long* ticks, uint32_t iters, uint32_t seed) {
  extern __shared__ uint32_t issue_lds[];
  uint64_t x[8];
  for (int c = 0; c < 8; c++) x[c] = ((uint64_t)(seed + threadIdx.x * 7 + c * 0x9e3779b9u) << 20) | 0x3ff0000000000001ull;
  const uint32_t y = seed ^ 0x5555u ^ threadIdx.x, z = seed + 3u;
  if (threadIdx.x == 0) issue_lds[0] = seed;  // touch the allocation
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();      // shader clock
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 32; rep++) {  // 256 instructions per trip: the loop's scalar bookkeeping and taken branch disappear in them
#pragma unroll
      for (int c = 0; c < 8; c++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z) : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  uint64_t r = 0;
  for (int c = 0; c < 8; c++) r ^= x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
  if ((threadIdx.x & 63) == 0) {
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    ticks[2 * wave] = t1 - t0;
    ticks[2 * wave + 1] = r1 - r0;
  }
}

extern "C" int32_t kzg_microbench_valu_issue(const kzg_ctx* ctx, uint32_t waves_per_simd, uint32_t iters, double* cycles_per_inst, double* clock_ghz) try {
  if (!ctx || !cycles_per_inst || !clock_ghz || waves_per_simd < 1 || waves_per_simd > 4 || iters == 0) return fail(KZG_FAIL_ARGUMENT, "bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const uint32_t blocks = ctx->num_cus, threads = 256 * waves_per_simd;
  const size_t lds = (size_t)(160 * 1024) - 1024;  // one workgroup per CU
  HIP_TRY(hipFuncSetAttribute((const void*)k_microbench_valu_issue, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint32_t* d_out = nullptr;
  unsigned long long* d_ticks = nullptr;
  HIP_TRY(hipMalloc(&d_out, (size_t)blocks * threads * 4));
  HIP_TRY(hipMalloc(&d_ticks, (size_t)blocks * (threads / 64) * 2 * 8));
  // warm-up: ~20 ms of the same load, so that the timed launch does not run while the clocks come up from an idle chip (readings
  // taken right after a long idle stretch were off by factors: tests/test_gpu_round4.py::test_measurement_aids)
  hipLaunchKernelGGL(k_microbench_valu_issue, dim3(blocks), dim3(threads), lds, nullptr, d_out, d_ticks, iters > 20000u ? iters : 20000u, 1u);
  hipLaunchKernelGGL(k_microbench_valu_issue, dim3(blocks), dim3(threads), lds, nullptr, d_out, d_ticks, iters, 1u);
  std::vector<unsigned long long> t((size_t)blocks * (threads / 64) * 2);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(t.data(), d_ticks, t.size() * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  (void)hipFree(d_ticks);
  if (e != hipSuccess) return fail(KZG_FAIL_HIP, std::string("valu issue microbenchmark: ") + hipGetErrorString(e));
  std::vector<double> cyc, ghz;
  for (size_t i = 0; i < t.size(); i += 2) {
    cyc.push_back((double)t[i] / ((double)iters * 256.0 * waves_per_simd));
    if (t[i + 1]) ghz.push_back((double)t[i] / ((double)t[i + 1] * 10.0));
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  *cycles_per_inst = cyc[cyc.size() / 2];  // median over the waves
  *clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  return 0;
} catch (...) {
  return abi_exception();
}

// Shader clock UNDER A REAL WORKLOAD: eight single-wave workgroups (dealt round-robin over the XCDs) that do nothing but
// sleep for `ticks` of the 100-MHz real-time counter and report how far the shader-clock counter moved meanwhile.  A probe
// wave needs a handful of registers, so it sits beside whatever fills the chip (two MSM waves hold 464 of a SIMD's 512 VGPRs)
// and issues one instruction per 2 us: the kernels it watches do not notice it.  bench.py launches it on a stream of its
// own next to the timed calls and prices SQ_INSTS_VALU with the clock it reports (roofline.valu_issue.clock_ghz).
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, unsigned long long ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks) {
    __builtin_amdgcn_s_sleep(127);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  out[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
  out[2 * blockIdx.x + 1] = r1 - r0;
}
constexpr int KZG_CLOCK_PROBES = 8;
extern "C" int32_t kzg_clock_probe_launch(const kzg_ctx* ctx, uint32_t duration_us) try {
  if (!ctx || duration_us == 0 || duration_us > 10000000u) return fail(KZG_FAIL_ARGUMENT, "bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> guard(ctx->prof_lock);
  if (!ctx->d_clock_probe) {
    HIP_TRY(hipMalloc(&ctx->d_clock_probe, KZG_CLOCK_PROBES * 2 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->probe_stream, hipStreamNonBlocking));
  }
  HIP_TRY(hipMemsetAsync(ctx->d_clock_probe, 0, KZG_CLOCK_PROBES * 2 * sizeof(unsigned long long), ctx->probe_stream));
  hipLaunchKernelGGL(k_clock_probe, dim3(KZG_CLOCK_PROBES), dim3(64), 0, ctx->probe_stream, ctx->d_clock_probe, (unsigned long long)duration_us * 100ull);
  HIP_TRY(hipGetLastError());
  return 0;
} catch (...) {
  return abi_exception();
}
extern "C" int32_t kzg_clock_probe_read(const kzg_ctx* ctx, double* ghz_mean, double* ghz_min, double* ghz_max) try {
  if (!ctx || !ghz_mean || !ghz_min || !ghz_max) return fail(KZG_FAIL_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> guard(ctx->prof_lock);
  if (!ctx->d_clock_probe) return fail(KZG_FAIL_ARGUMENT, "kzg_clock_probe_read without kzg_clock_probe_launch");
  unsigned long long h[KZG_CLOCK_PROBES * 2];
  HIP_TRY(hipStreamSynchronize(ctx->probe_stream));
  HIP_TRY(hipMemcpyAsync(h, ctx->d_clock_probe, sizeof(h), hipMemcpyDeviceToHost, ctx->probe_stream));
  HIP_TRY(hipStreamSynchronize(ctx->probe_stream));
  double sum = 0, lo = 1e30, hi = 0;
  int cnt = 0;
  for (int k = 0; k < KZG_CLOCK_PROBES; k++) {
    if (!h[2 * k + 1]) continue;
    const double g = (double)h[2 * k] / ((double)h[2 * k + 1] * 10.0);
    sum += g;
    lo = g < lo ? g : lo;
    hi = g > hi ? g : hi;
    cnt++;
  }
  if (!cnt) return fail(KZG_FAIL_HIP, "clock probe returned nothing");
  *ghz_mean = sum / cnt;
  *ghz_min = lo;
  *ghz_max = hi;
  return 0;
} catch (...) {
  return abi_exception();
}

template <class F>
__device__ __forceinline__ uint32_t selftest_one(uint64_t& rng) {
  bn<F::N> a, b, r1, r2;
#pragma unroll
  for (int q = 0; q < F::N; q++) {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    a.v[q] = (uint32_t)rng;
    b.v[q] = (uint32_t)(rng >> 32);
  }
  // bring both below the modulus: clear top bits, then one conditional subtraction
  a.v[F::N - 1] &= (F::mod(F::N - 1) | (F::mod(F::N - 1) >> 1) | (F::mod(F::N - 1) >> 2) | (F::mod(F::N - 1) >> 4) | (F::mod(F::N - 1) >> 8) | (F::mod(F::N - 1) >> 16));
  b.v[F::N - 1] &= (F::mod(F::N - 1) | (F::mod(F::N - 1) >> 1) | (F::mod(F::N - 1) >> 2) | (F::mod(F::N - 1) >> 4) | (F::mod(F::N - 1) >> 8) | (F::mod(F::N - 1) >> 16));
  reduce_once<F>(a, a, 0);
  reduce_once<F>(b, b, 0);
  if ((rng & 15) == 0) a = b;            // squarings
  if ((rng & 1023) == 1) bn_zero(b);     // zero operand
  if ((rng & 1023) == 2) {               // p - 1
    b = modulus<F>();
    b.v[0] -= 1;
  }
  mont_mul<F>(r1, a, b);
  mont_mul_plainc<F>(r2, a, b);
  return bn_eq(r1, r2) ? 0u : 1u;
}

// The carry-free radix-2^28 / 2^29 multipliers of the hot loops (explicit v_mad_u64_u32 chains, fp28.cuh / fr29.cuh)
// against the 32-bit-limb multiplier on the same operands: a*b*2^-392 is brought to a*b*2^-384 with one more
// product by 2^400 (Fr: 2^-261 -> 2^-256 by 2^266), then the canonical limbs must be equal.  Covers f28_mul, f28_sqr,
// f28_mul2 (as a*b + b*a = 2ab), f29_mul, f29_sqr, f29_mul2.
__device__ __forceinline__ void selftest_operand(uint64_t& rng, uint32_t* v, int n, uint32_t top_mask) {
  for (int q = 0; q < n; q++) {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    v[q] = (uint32_t)(rng >> 16);
  }
  v[n - 1] &= top_mask;
}
__device__ __noinline__ uint32_t selftest_radix(uint64_t& rng) {
  uint32_t bad = 0;
  {
    fp_t a, b, want, want2, got;
    selftest_operand(rng, a.v, 12, 0x0fffffffu);  // < 2^380 < p
    selftest_operand(rng, b.v, 12, 0x0fffffffu);
    if ((rng & 7) == 0) b = a;
    mont_mul<FpParams>(want, a, b);
    add_mod<FpParams>(want2, want, want);
    fp28 A, B, k, x;
    f28_from_bn(A, a);
    f28_from_bn(B, b);
    {
      const uint32_t t[F28_N] = KZG_FP28_R400;
#pragma unroll
      for (int q = 0; q < F28_N; q++) k.l[q] = t[q];
    }
    f28_mul(x, A, B);
    f28_mul(x, x, k);
    f28_to_bn(got, x);
    canonicalize<FpParams>(got);
    bad += bn_eq(got, want) ? 0u : 1u;
    f28_mul2(x, A, B, B, A);
    f28_mul(x, x, k);
    f28_to_bn(got, x);
    canonicalize<FpParams>(got);
    bad += bn_eq(got, want2) ? 0u : 1u;
    mont_mul<FpParams>(want, a, a);
    f28_sqr(x, A);
    f28_mul(x, x, k);
    f28_to_bn(got, x);
    canonicalize<FpParams>(got);
    bad += bn_eq(got, want) ? 0u : 1u;
  }
  {
    fr_t a, b, want, want2, got;
    selftest_operand(rng, a.v, 8, 0x3fffffffu);  // < 2^254 < r
    selftest_operand(rng, b.v, 8, 0x3fffffffu);
    if ((rng & 7) == 0) b = a;
    mont_mul<FrParams>(want, a, b);
    add_mod<FrParams>(want2, want, want);
    fr29 A, B, k, x;
    f29_from_bn(A, a);
    f29_from_bn(B, b);
    {
      const uint32_t t[F29_N] = KZG_FR29_R266;
#pragma unroll
      for (int q = 0; q < F29_N; q++) k.l[q] = t[q];
    }
    f29_mul(x, A, B);
    f29_mul(x, x, k);
    f29_to_canonical_bn(got, x);
    bad += bn_eq(got, want) ? 0u : 1u;
    f29_mul2(x, A, B, B, A);
    f29_mul(x, x, k);
    f29_to_canonical_bn(got, x);
    bad += bn_eq(got, want2) ? 0u : 1u;
    mont_mul<FrParams>(want, a, a);
    f29_sqr(x, A);
    f29_mul(x, x, k);
    f29_to_canonical_bn(got, x);
    bad += bn_eq(got, want) ? 0u : 1u;
  }
  return bad;
}

__global__ __launch_bounds__(64) void k_selftest_field_mul(uint64_t iters, unsigned long long* mismatches) {
  uint64_t rng = 0x9E3779B97F4A7C15ull * (blockIdx.x * 64ull + threadIdx.x + 1);
  uint32_t bad = 0;
#pragma unroll 1
  for (uint64_t it = 0; it < iters; it++) {
    bad += selftest_one<FpParams>(rng);
    bad += selftest_one<FrParams>(rng);
    bad += selftest_radix(rng);
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

extern "C" int32_t kzg_selftest_field_mul(const kzg_ctx* ctx, uint64_t lanes, uint64_t iters, uint64_t* mismatches) try {
  if (!ctx || !mismatches || lanes == 0) return fail(KZG_FAIL_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d, 0, sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_selftest_field_mul, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, nullptr, iters, d);
  HIP_TRY(hipGetLastError());
  unsigned long long h = 0;
  HIP_TRY(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  *mismatches = h;
  return 0;
} catch (...) {
  return abi_exception();
}

