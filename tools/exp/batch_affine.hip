// Experiment (VERDICT r02 "next" 7): batch-affine accumulation measured instead of argued.
//
// The comb MSM's additions are independent sums of table entries, so they can be done as AFFINE additions with Montgomery's
// trick: per lane a batch of m additions shares one inversion.
//   forward :  d_j = x2_j - x1_j ,  pre_j = d_0 ... d_{j-1}  (stored),  1 product per addition
//   invert  :  inv = 1 / (d_0 ... d_{m-1})                   one safegcd per lane and batch (all 64 lanes in lock-step)
//   backward:  1/d_j = inv * pre_j ; inv *= d_j ; lambda = (y2 - y1)/d_j ; x3 = lambda^2 - x1 - x2 ; y3 = lambda (x1 - x3) - y1
// = 5 products + 1 squaring per addition (6 reductions) against madd-2008-s's 6 products + 2 squarings + 1 double product
// (9 reductions) in k_msm_comb28.  The price is memory: m prefixes per lane fit neither the VGPRs nor LDS (320 B per lane at
// two waves per SIMD), so they are staged through HBM ([j][limb-quad][lane]: every store / load is one contiguous 1-KiB row
// per wave), and both operands are gathered twice (x alone on the way up, x and y on the way down).
//
// This file measures the FIRST tree level -- both operands are table entries, half of all additions of a pairwise tree and
// the level with the most gather traffic -- on a real-size table (default 2^31 entries = 192 GiB, random indices), with the
// engine's own field code (fp28.cuh, modinv30.cuh), two waves per SIMD, and beside it the production step (XYZZ accumulator
// += gathered entry, xyzz28_madd_fast) in the same harness on the same index stream.  A small run on valid curve points checks
// the affine formulas against the XYZZ adder on the host.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/batch_affine.hip -o tools/exp/batch_affine
// Run  : tools/exp/batch_affine [log2_entries=31] [waves=2048] [adds_per_lane=3072]     (JSON on stdout)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../kateth_amd/csrc/msm_fixed.cuh"
using namespace kzg;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// ---- index stream: the same pseudo-random entry pair for (lane stream, step) on the way up and on the way down ----------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ void pair_of(uint64_t stream, uint32_t step, uint64_t mask, uint64_t& i1, uint64_t& i2) {
  const uint64_t h = mix64(stream * 0x9E3779B97F4A7C15ull + step);
  i1 = h & mask;
  i2 = (h >> 32 | h << 32) & mask;
  if (i2 == i1) i2 = (i1 + 1) & mask;
}

// table entries for the throughput runs: any residues below 2^380 (the field code only needs value bounds)
__global__ void k_fill(uint4* table, uint64_t quads) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = mix64(2 * i + 1), b = mix64(2 * i + 2);
    uint4 v = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    if (i % 3 == 2) v.w &= 0x0fffffffu;  // top limb of x (quad 2) and of y (quad 5)
    table[i] = v;
  }
}

__device__ __forceinline__ void load_x48(fp_t& x, const uint4* __restrict__ tbl, uint64_t idx) {
  const uint4* p = tbl + idx * 6;
  const uint4 a0 = p[0], a1 = p[1], a2 = p[2];
  x.v[0] = a0.x; x.v[1] = a0.y; x.v[2] = a0.z; x.v[3] = a0.w;
  x.v[4] = a1.x; x.v[5] = a1.y; x.v[6] = a1.z; x.v[7] = a1.w;
  x.v[8] = a2.x; x.v[9] = a2.y; x.v[10] = a2.z; x.v[11] = a2.w;
}
// staged field element: 14 limbs in four 16-byte pieces, piece q of lane l of row r at ((r * 4 + q) * 64 + l)
__device__ __forceinline__ void stage_store(uint4* base, uint64_t row, int lane, const fp28& a) {
  uint4* p = base + row * 256 + lane;
  p[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
  p[64] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
  p[128] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
  p[192] = make_uint4(a.l[12], a.l[13], 0u, 0u);
}
__device__ __forceinline__ void stage_load(fp28& a, const uint4* base, uint64_t row, int lane) {
  const uint4* p = base + row * 256 + lane;
  const uint4 q0 = p[0], q1 = p[64], q2 = p[128], q3 = p[192];
  a.l[0] = q0.x; a.l[1] = q0.y; a.l[2] = q0.z; a.l[3] = q0.w;
  a.l[4] = q1.x; a.l[5] = q1.y; a.l[6] = q1.z; a.l[7] = q1.w;
  a.l[8] = q2.x; a.l[9] = q2.y; a.l[10] = q2.z; a.l[11] = q2.w;
  a.l[12] = q3.x; a.l[13] = q3.y;
}
// d = b - a for canonical a, b: limbs < 3 * 2^28, value < 3p
__device__ __forceinline__ void sub_canon(fp28& d, const fp28& b, const fp28& a) {
  fp28 t;
  f28_neg_2p(t, a);
  f28_add(d, b, t);
}

// One wave per block; every lane runs `batches` batches of M affine additions of two gathered table entries.
// prefix: waves * M rows of staging; results: waves * 2 rows (the sums are overwritten: only their store traffic is modelled,
// a real tree would keep them as the next level's operands, 112 B per sum).
template <int M>
__global__ __launch_bounds__(64, 2) void k_batch_affine(const uint4* __restrict__ table, uint64_t mask, uint4* __restrict__ prefix,
                                                        uint4* __restrict__ results, uint32_t batches, uint32_t* __restrict__ flags) {
  const int lane = threadIdx.x;
  const uint64_t wave = blockIdx.x;
  const uint64_t stream = wave * 64 + lane;
  uint4* pre = prefix + wave * (uint64_t)M * 256;
  uint4* res = results + wave * (uint64_t)(2 * M) * 256;
  uint32_t special = 0;
#pragma unroll 1
  for (uint32_t b = 0; b < batches; b++) {
    const uint32_t step0 = b * (uint32_t)M;
    // ---- forward: prefix products of the denominators
    fp28 run = f28_one();
    fp_t nx1, nx2;
    {
      uint64_t i1, i2;
      pair_of(stream, step0, mask, i1, i2);
      load_x48(nx1, table, i1);
      load_x48(nx2, table, i2);
    }
#pragma unroll 1
    for (int j = 0; j < M; j++) {
      fp28 x1, x2, d;
      f28_from_bn(x1, nx1);
      f28_from_bn(x2, nx2);
      if (j + 1 < M) {
        uint64_t i1, i2;
        pair_of(stream, step0 + j + 1, mask, i1, i2);
        load_x48(nx1, table, i1);
        load_x48(nx2, table, i2);
      }
      sub_canon(d, x2, x1);
      if (f28_maybe_zero(d)) {  // equal x: P + P or P - P would need the slow path; a real kernel substitutes 1 and flags the slot
        special++;
        d = f28_one();
      }
      stage_store(pre, j, lane, run);
      f28_mul(run, run, d);
    }
    // ---- one inversion per lane and batch
    fp28 inv;
    f28_inv(inv, run);
    // ---- backward
    fp_t ax1, ay1, ax2, ay2;
    fp28 npre;
    {
      uint64_t i1, i2;
      pair_of(stream, step0 + M - 1, mask, i1, i2);
      load_affine96(ax1, ay1, table, i1);
      load_affine96(ax2, ay2, table, i2);
      stage_load(npre, pre, M - 1, lane);
    }
#pragma unroll 1
    for (int j = M - 1; j >= 0; j--) {
      fp28 x1, y1, x2, y2, pj = npre;
      f28_from_bn(x1, ax1);
      f28_from_bn(y1, ay1);
      f28_from_bn(x2, ax2);
      f28_from_bn(y2, ay2);
      if (j > 0) {
        uint64_t i1, i2;
        pair_of(stream, step0 + j - 1, mask, i1, i2);
        load_affine96(ax1, ay1, table, i1);
        load_affine96(ax2, ay2, table, i2);
        stage_load(npre, pre, j - 1, lane);
      }
      fp28 d, dy, inv_d, lam, x3, y3, t;
      sub_canon(d, x2, x1);
      if (f28_maybe_zero(d)) d = f28_one();
      f28_mul(inv_d, inv, pj);   // 1 / d_j
      f28_mul(inv, inv, d);      // inverse of the shorter prefix
      sub_canon(dy, y2, y1);
      f28_mul(lam, dy, inv_d);   // lambda
      f28_sqr(x3, lam);
      f28_add(t, x1, x2);        // limbs < 2^29, value < 2p
      f28_sub_8p3(x3, x3, t);    // lambda^2 - x1 - x2: limbs < 5 * 2^28, value < 10p
      f28_carry_pass(x3);
      f28_sub_16p(t, x1, x3);    // x1 - x3: limbs < 3 * 2^28, value < 17p
      f28_mul(y3, lam, t);
      f28_neg_2p(t, y1);
      f28_add(y3, y3, t);        // lambda (x1 - x3) - y1: limbs < 3 * 2^28, value < 4p
      stage_store(res, 2 * j, lane, x3);
      stage_store(res, 2 * j + 1, lane, y3);
    }
  }
  if (special) atomicAdd(flags, special);
}

// The production step in the same harness: XYZZ accumulator += gathered entry (k_msm_comb28's hot loop without the comb walker).
__global__ __launch_bounds__(64, 2) void k_xyzz_chain(const uint4* __restrict__ table, uint64_t mask, uint4* __restrict__ results, uint32_t adds) {
  const int lane = threadIdx.x;
  const uint64_t wave = blockIdx.x;
  const uint64_t stream = wave * 64 + lane;
  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  fp_t nx, ny;
  uint64_t nidx;
  {
    uint64_t i2;
    pair_of(stream, 0, mask, nidx, i2);
    load_affine96(nx, ny, table, nidx);
  }
#pragma unroll 1
  for (uint32_t t = 0; t < adds; t++) {
    fp28 cx, cy;
    f28_load_entry(cx, cy, nx, ny, (t & 1u) != 0u);
    const uint64_t cidx = nidx;
    if (t + 1 < adds) {
      uint64_t i2;
      pair_of(stream, t + 1, mask, nidx, i2);
      load_affine96(nx, ny, table, nidx);
    }
    bool done = false;
    if (!acc.inf) done = xyzz28_madd_fast(acc, cx, cy);
    if (!done) {
      g1_xyzz28 tmp = acc;
      fp_t rx, ry;
      load_affine96(rx, ry, table, cidx);
      fp28 sx, sy;
      f28_load_entry(sx, sy, rx, ry, (t & 1u) != 0u);
      xyzz28_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  uint4* res = results + wave * 2 * 256;
  stage_store(res, 0, lane, acc.x);
  stage_store(res, 1, lane, acc.y);
}

// ---- validity run: 1,024 valid points k*G in the table's format; every sum checked against the XYZZ adder on the host --------
static void host_points(std::vector<uint32_t>& tbl, std::vector<g1_xyzz>& pts, int count) {
  const uint32_t gx[12] = KZG_FP_G1X_MONT, gy[12] = KZG_FP_G1Y_MONT;
  fp_t x, y;
  for (int q = 0; q < 12; q++) {
    x.v[q] = gx[q];
    y.v[q] = gy[q];
  }
  g1_xyzz acc;
  xyzz_set_inf(acc);
  tbl.resize((size_t)count * 24);
  pts.resize(count);
  for (int k = 0; k < count; k++) {
    xyzz_madd(acc, x, y);
    fp_t ax, ay, rx, ry;
    xyzz_to_affine(ax, ay, acc);
    xyzz_from_affine(pts[k], ax, ay);
    fp_to_r392(rx, ax);
    fp_to_r392(ry, ay);
    for (int q = 0; q < 12; q++) {
      tbl[(size_t)k * 24 + q] = rx.v[q];
      tbl[(size_t)k * 24 + 12 + q] = ry.v[q];
    }
  }
}
static uint64_t h_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static int validity_run() {
  constexpr int M = 64, COUNT = 1024;
  std::vector<uint32_t> tbl;
  std::vector<g1_xyzz> pts;
  host_points(tbl, pts, COUNT);
  uint4 *d_tbl, *d_pre, *d_res;
  uint32_t* d_flags;
  CHECK(hipMalloc(&d_tbl, tbl.size() * 4));
  CHECK(hipMemcpy(d_tbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_pre, (size_t)M * 256 * 16));
  CHECK(hipMalloc(&d_res, (size_t)2 * M * 256 * 16));
  CHECK(hipMalloc(&d_flags, 4));
  CHECK(hipMemset(d_flags, 0, 4));
  hipLaunchKernelGGL(k_batch_affine<M>, dim3(1), dim3(64), 0, nullptr, d_tbl, (uint64_t)(COUNT - 1), d_pre, d_res, 1u, d_flags);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> res((size_t)2 * M * 256 * 4);
  CHECK(hipMemcpy(res.data(), d_res, res.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int j = 0; j < M; j++) {
      const uint64_t h = h_mix64((uint64_t)lane * 0x9E3779B97F4A7C15ull + (uint64_t)j);
      uint64_t i1 = h & (COUNT - 1), i2 = (h >> 32 | h << 32) & (COUNT - 1);
      if (i2 == i1) i2 = (i1 + 1) & (COUNT - 1);
      g1_xyzz want = pts[i1];
      xyzz_add(want, pts[i2]);
      fp_t wx, wy;
      xyzz_to_affine(wx, wy, want);
      fp28 gx3, gy3;
      for (int q = 0; q < 14; q++) {
        gx3.l[q] = res[((size_t)(2 * j) * 256 + (q / 4) * 64 + lane) * 4 + (q % 4)];
        gy3.l[q] = res[((size_t)(2 * j + 1) * 256 + (q / 4) * 64 + lane) * 4 + (q % 4)];
      }
      f28_normalize(gx3);
      f28_normalize(gy3);
      fp_t hx, hy;
      f28_to_fp(hx, gx3);
      f28_to_fp(hy, gy3);
      if (!bn_eq(hx, wx) || !bn_eq(hy, wy)) bad++;
    }
  uint32_t flags = 0;
  CHECK(hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_tbl);
  (void)hipFree(d_pre);
  (void)hipFree(d_res);
  (void)hipFree(d_flags);
  fprintf(stderr, "[batch_affine] validity: %d of %d sums differ from the XYZZ adder, %u flagged equal-x slots\n", bad, 64 * M, flags);
  return bad;
}

template <int M>
static double time_batch(const uint4* d_tbl, uint64_t mask, uint32_t waves, uint32_t adds_per_lane, uint4* d_pre, uint4* d_res, uint32_t* d_flags) {
  const uint32_t batches = adds_per_lane / M;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_batch_affine<M>, dim3(waves), dim3(64), 0, nullptr, d_tbl, mask, d_pre, d_res, 1u, d_flags);  // warm-up
  CHECK(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL(k_batch_affine<M>, dim3(waves), dim3(64), 0, nullptr, d_tbl, mask, d_pre, d_res, batches, d_flags);
  CHECK(hipEventRecord(e1, nullptr));
  CHECK(hipGetLastError());
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)waves * 64.0 * batches * M / (ms * 1e-3);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 31;
  const uint32_t waves = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048u;
  const uint32_t adds = argc > 3 ? (uint32_t)atoi(argv[3]) : 3072u;
  const int bad = validity_run();
  const uint64_t entries = 1ull << lg;
  uint4* d_tbl = nullptr;
  CHECK(hipMalloc(&d_tbl, entries * 96));
  hipLaunchKernelGGL(k_fill, dim3(256 * 64), dim3(256), 0, nullptr, d_tbl, entries * 6);  // grid-stride: a launch is limited to 2^32 threads
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  constexpr int MMAX = 256;
  uint4 *d_pre, *d_res;
  uint32_t* d_flags;
  CHECK(hipMalloc(&d_pre, (size_t)waves * MMAX * 256 * 16));
  CHECK(hipMalloc(&d_res, (size_t)waves * 2 * MMAX * 256 * 16));
  CHECK(hipMalloc(&d_flags, 4));
  CHECK(hipMemset(d_flags, 0, 4));
  // the production step
  double xyzz_rate;
  {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_xyzz_chain, dim3(waves), dim3(64), 0, nullptr, d_tbl, entries - 1, d_res, 64u);
    CHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_xyzz_chain, dim3(waves), dim3(64), 0, nullptr, d_tbl, entries - 1, d_res, adds);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    xyzz_rate = (double)waves * 64.0 * adds / (ms * 1e-3);
  }
  const double r16 = time_batch<16>(d_tbl, entries - 1, waves, adds, d_pre, d_res, d_flags);
  const double r32 = time_batch<32>(d_tbl, entries - 1, waves, adds, d_pre, d_res, d_flags);
  const double r64 = time_batch<64>(d_tbl, entries - 1, waves, adds, d_pre, d_res, d_flags);
  const double r128 = time_batch<128>(d_tbl, entries - 1, waves, adds, d_pre, d_res, d_flags);
  const double r256 = time_batch<256>(d_tbl, entries - 1, waves, adds, d_pre, d_res, d_flags);
  uint32_t flags = 0;
  CHECK(hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost));
  printf("{\"table_entries_log2\": %d, \"table_gib\": %.1f, \"waves\": %u, \"adds_per_lane\": %u, \"validity_mismatches\": %d, "
         "\"xyzz_madd_adds_per_s\": %.4g, \"batch_affine_adds_per_s\": {\"m16\": %.4g, \"m32\": %.4g, \"m64\": %.4g, \"m128\": %.4g, \"m256\": %.4g}, "
         "\"ratio_vs_xyzz\": {\"m16\": %.3f, \"m32\": %.3f, \"m64\": %.3f, \"m128\": %.3f, \"m256\": %.3f}, \"equal_x_flags\": %u, "
         "\"bytes_per_add\": {\"gathers\": 288, \"prefix_store_and_load\": 128, \"result_store\": 128}}\n",
         lg, entries * 96 / 1073741824.0, waves, adds, bad, xyzz_rate, r16, r32, r64, r128, r256, r16 / xyzz_rate, r32 / xyzz_rate, r64 / xyzz_rate,
         r128 / xyzz_rate, r256 / xyzz_rate, flags);
  return bad ? 1 : 0;
}
