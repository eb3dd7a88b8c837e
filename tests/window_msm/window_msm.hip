// TEST-ONLY translation unit: linked with the PRODUCT's object files into tests/window_msm/libkateth_amd_window_msm.so.
// It registers an MsmOverride (kateth_amd/csrc/engine_internal.hpp) at load time, so a context created through that
// library can run, selected by the environment at kzg_ctx_create:
//   KATETH_AMD_MSM=window        round 1's window-table MSM on radix-2^28 limbs (k_msm_fixed28, 2^392-Montgomery table)
//   KATETH_AMD_MSM_RADIX=32      the same walk on 12 x 32-bit limbs (k_msm_fixed, 2^384-Montgomery table)
//   KATETH_AMD_WAVE_TIMES=units  the product's comb kernel instantiated with per-unit timestamps (kzg_test_read_wave_times,
//                                tools/gpu_wave_times.py)
// Independent cross-checks of the comb (tests/test_gpu_parity.py, tools/gpu_soak_msm.py).  Nothing here is in the product
// library, and the product sources carry no conditional compilation for it.
#include "../../kateth_amd/csrc/engine_internal.hpp"
// the kernel headers this unit instantiates for itself (static kernels: its own copies, the product objects keep theirs)
#include "../../kateth_amd/csrc/msm_comb.cuh"
#include "window_msm.cuh"

namespace {

struct WindowState {
  MsmGeom geom{};
  bool radix28 = true;
  uint64_t* d_wave_times = nullptr;  // per-unit timestamps of the last k_msm_comb30 launch that fitted
  uint64_t wave_times_cap = 0;
};
WindowState* state_of(const kzg_ctx* ctx) { return reinterpret_cast<WindowState*>(ctx->override_state); }

MsmGeom make_geom(uint32_t c) {
  MsmGeom g;
  g.c = c;
  g.W = (256 + c - 1) / c;
  g.half = 1u << (c - 1);
  // largest raw top digit of a scalar < 2^255, plus a possible carry
  uint32_t top_bits_lo = c * (g.W - 1);
  uint32_t top_raw_max = (top_bits_lo >= 255) ? 0u : ((1u << (255 - top_bits_lo)) - 1u);
  uint32_t top = top_raw_max + 1u;
  g.top_entries = top < g.half ? top : g.half;
  return g;
}

// ---- window table: every signed-digit multiple of every window base --------------------------------------------------
int32_t window_build(kzg_ctx* ctx) {
  WindowState* ws = state_of(ctx);
  const MsmGeom g = ws->geom;
  hipStream_t st = nullptr;
  const uint64_t entries = table_entries(g);
  ctx->table_bytes = entries * 96;
  HIP_TRY(hipMalloc(&ctx->d_table, ctx->table_bytes));
  uint4* d_win_bases = nullptr;
  g1_xyzz* d_tmp = nullptr;
  uint32_t* d_inf_seen = nullptr;
  auto cleanup = [&]() {
    if (d_win_bases) (void)hipFree(d_win_bases);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_inf_seen) (void)hipFree(d_inf_seen);
  };
  if (hipMalloc(&d_win_bases, (size_t)g.W * 4096 * 96) != hipSuccess || hipMalloc(&d_tmp, (size_t)4096 * g.half * sizeof(g1_xyzz)) != hipSuccess ||
      hipMalloc(&d_inf_seen, sizeof(uint32_t)) != hipSuccess || hipMemset(d_inf_seen, 0, sizeof(uint32_t)) != hipSuccess) {
    cleanup();
    return fail(KZG_FAIL_HIP, "window table: allocation failed");
  }
  hipLaunchKernelGGL(k_table_window_bases, dim3(64), dim3(64), 0, st, ctx->d_bases_brp, d_win_bases, g);
  for (uint32_t j = 0; j < g.W; j++) {
    const uint32_t e = (j + 1 < g.W) ? g.half : g.top_entries;
    const uint64_t count = (uint64_t)4096 * e;
    uint32_t segs = e / 64;  // slices of >= 64 entries, at most 32 per base (two waves per SIMD)
    segs = segs < 1 ? 1 : (segs > 32 ? 32 : segs);
    hipLaunchKernelGGL(k_table_chain, dim3(64 * segs), dim3(64), 0, st, d_win_bases, j, e, segs, d_tmp);
    constexpr int KN = 8;
    const uint64_t threads = (count + KN - 1) / KN;
    hipLaunchKernelGGL(k_table_normalize<KN>, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, d_tmp, count, ctx->d_table,
                       table_index(g, j, 0, 1), ws->radix28, d_inf_seen);
  }
  hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  cleanup();
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(KZG_FAIL_HIP, "window table build failed");
  return 0;
}

int32_t window_launch(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits,
                      uint32_t /*lpb*/, void* /*scratch*/, hipStream_t st) {
  const WindowState* ws = state_of(ctx);
  ProfScope ps(ctx, PROF_MSM_FIXED, st);
  const dim3 grid((unsigned)(n * splits)), block(64);
  if (!ws->radix28) {
    if (be_bytes)
      hipLaunchKernelGGL((k_msm_fixed<true, 2>), grid, block, 0, st, d_scalars, splits, ctx->d_table, ws->geom, partials, d_status);
    else
      hipLaunchKernelGGL((k_msm_fixed<false, 2>), grid, block, 0, st, d_scalars, splits, ctx->d_table, ws->geom, partials, d_status);
  } else {
    if (be_bytes)
      hipLaunchKernelGGL((k_msm_fixed28<true>), grid, block, 0, st, d_scalars, splits, ctx->d_table, ws->geom, partials, d_status);
    else
      hipLaunchKernelGGL((k_msm_fixed28<false>), grid, block, 0, st, d_scalars, splits, ctx->d_table, ws->geom, partials, d_status);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- the product's comb kernel with per-unit timestamps ----------------------------------------------------------------
int32_t timed_comb_launch(const kzg_ctx* ctx, bool be_bytes, const uint8_t* d_scalars, uint64_t n, int32_t* d_status, g1_xyzz* partials, uint32_t splits,
                          uint32_t lpb, void* scratch, hipStream_t st) {
  const WindowState* ws = state_of(ctx);
  uint64_t* masks = reinterpret_cast<uint64_t*>(scratch);
  {
    ProfScope ps(ctx, PROF_TRANSPOSE, st);
    if (be_bytes)
      hipLaunchKernelGGL((k_comb_transpose<true>), dim3((unsigned)(n * 8)), dim3(512), 0, st, d_scalars, n, masks, d_status);
    else
      hipLaunchKernelGGL((k_comb_transpose<false>), dim3((unsigned)(n * 8)), dim3(512), 0, st, d_scalars, n, masks, d_status);
  }
  ProfScope ps(ctx, PROF_MSM_FIXED, st);
  const bool lat = msm_uses_lat(ctx, splits);
  const uint64_t units = msm_units(n, splits, lpb);
  hipLaunchKernelGGL(k_msm_comb30<true>, dim3((unsigned)units), dim3(64), 0, st, masks, n, splits, lpb, lat ? ctx->d_table_lat : ctx->d_table,
                     lat ? ctx->comb_lat : ctx->comb, partials, (const uint4*)(lat ? ctx->d_comb_k_lat : ctx->d_comb_k), units <= ws->wave_times_cap ? ws->d_wave_times : (uint64_t*)nullptr);
  HIP_TRY(hipGetLastError());
  return 0;
}

void override_destroy(kzg_ctx* ctx) {
  WindowState* ws = state_of(ctx);
  if (!ws) return;
  if (ws->d_wave_times) (void)hipFree(ws->d_wave_times);
  delete ws;
  ctx->override_state = nullptr;
}

const MsmOverride OVR_WINDOW28 = {"k_msm_fixed28", true, 0, window_build, window_launch, override_destroy};
const MsmOverride OVR_WINDOW32 = {"k_msm_fixed", true, 0, window_build, window_launch, override_destroy};
const MsmOverride OVR_TIMED_COMB = {"k_msm_comb30", false, 0, nullptr, timed_comb_launch, override_destroy};
MsmOverride g_ovr_slots[2];  // adds_per_blob depends on the window: one mutable copy per kind (contexts of one process use one geometry at a time in the tests)

const MsmOverride* choose_override(kzg_ctx* ctx, uint32_t window_bits) {
  const char* msm = getenv("KATETH_AMD_MSM");
  const char* radix = getenv("KATETH_AMD_MSM_RADIX");
  const bool window_mode = msm != nullptr && std::string(msm) == "window";
  const bool radix32 = radix != nullptr && atoi(radix) == 32;
  if (window_mode || radix32) {
    WindowState* ws = new WindowState();
    ws->geom = make_geom(window_bits > 16 ? 16 : window_bits);
    ws->radix28 = !radix32;
    ctx->override_state = ws;
    ctx->use_comb = false;
    ctx->window_class = ws->geom.c;
    MsmOverride& slot = g_ovr_slots[radix32 ? 1 : 0];
    slot = radix32 ? OVR_WINDOW32 : OVR_WINDOW28;
    slot.adds_per_blob = (uint64_t)ws->geom.W * 4096u;
    return &slot;
  }
  if (const char* e = getenv("KATETH_AMD_WAVE_TIMES")) {
    WindowState* ws = new WindowState();
    ws->wave_times_cap = (uint64_t)atoll(e);
    if (ws->wave_times_cap && (hipMalloc(&ws->d_wave_times, ws->wave_times_cap * 32) != hipSuccess ||
                               hipMemset(ws->d_wave_times, 0, ws->wave_times_cap * 32) != hipSuccess))
      ws->wave_times_cap = 0;
    ctx->override_state = ws;
    return &OVR_TIMED_COMB;
  }
  return nullptr;
}

struct Registrar {
  Registrar() { g_msm_override_hook = choose_override; }
} g_registrar;

}  // namespace

// {wall-clock start, wall-clock end (100 MHz ticks), shader cycles, XCC_ID << 32 | HW_ID} of each unit of the most recent
// k_msm_comb30 launch that fitted the buffer
extern "C" int32_t kzg_test_read_wave_times(const kzg_ctx* ctx, uint64_t* out, uint64_t units) {
  const WindowState* ws = ctx ? state_of(ctx) : nullptr;
  if (!ws || !ws->d_wave_times || units > ws->wave_times_cap) return fail(KZG_FAIL_ARGUMENT, "no wave-time buffer");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ws->d_wave_times, units * 32, hipMemcpyDeviceToHost));
  return 0;
}
