// Experiment: throughput of the radix-2^28 field/adder (fp28.cuh) against the 12x32 one on gfx950.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/fp28_bench.hip -o tools/exp/fp28_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../kateth_amd/csrc/fp28.cuh"
using namespace kzg;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(64, 2) void k_mul32(uint32_t* out, uint32_t iters) {
  fp_t a, b;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  for (int q = 0; q < 12; q++) { a.v[q] = FpParams::one(q) ^ (t & 0xffu); b.v[q] = FpParams::r2(q) >> 1; }
  a.v[11] &= 0x0fffffffu;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) { fp_t r; mont_mul_lazy<FpParams>(r, a, b); a = b; b = r; }
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= b.v[q];
  out[t] = x;
}
template <int MODE>
__global__ __launch_bounds__(64, 2) void k_mul28(uint32_t* out, uint32_t iters) {
  fp28 a, b;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  for (int q = 0; q < 14; q++) { a.l[q] = f28_one_limb(q) ^ (t & 0xffu); b.l[q] = f28_r384_limb(q); }
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    fp28 r;
    if (MODE == 0) f28_mul(r, a, b);
    if (MODE == 1) { f28_sqr(r, a); f28_add(r, r, b); f28_carry_pass(r); }
    if (MODE == 2) { fp28 c = f28_one(); f28_mul2(r, a, b, c, a); }
    a = b; b = r;
  }
  uint32_t x = 0; for (int q = 0; q < 14; q++) x ^= b.l[q];
  out[t] = x;
}
// adder loops: acc += P_k where the affine operand changes every step (a cheap permutation of two valid points is
// not available without a table, so the operand is a fixed valid point and its negation alternately; the adder is
// complete, the arithmetic executed is the generic path except at the cancellation steps which are excluded by
// using 3 distinct points g, 2g ... supplied by the host)
__global__ __launch_bounds__(64, 2) void k_madd32(uint32_t* out, const fp_t* pts, uint32_t npts, uint32_t iters) {
  g1_xyzz acc; xyzz_set_inf(acc);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t k = (it * 7u + t) % npts;
    fp_t x = pts[2 * k], y = pts[2 * k + 1];
    xyzz_madd_lazy(acc, x, y);
  }
  xyzz_canonicalize(acc);
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= acc.x.v[q] ^ acc.y.v[q] ^ acc.zz.v[q];
  out[t] = x;
}
__global__ __launch_bounds__(64, 2) void k_madd28(uint32_t* out, const fp_t* pts, uint32_t npts, uint32_t iters, uint32_t flip) {
  g1_xyzz28 acc; xyzz28_set_inf(acc);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t k = (it * 7u + t) % npts;
    fp_t x = pts[2 * k], y = pts[2 * k + 1];
    fp28 x2, y2;
    f28_load_entry(x2, y2, x, y, (it ^ flip) == 0x7fffffffu);
    if (acc.inf || !xyzz28_madd_fast(acc, x2, y2)) {
      g1_xyzz28 tmp = acc;  // copy: keeps the hot accumulator out of scratch
      fp_t rx = pts[2 * k], ry = pts[2 * k + 1];
      fp28 sx, sy;  // separate objects: the out-of-line call takes their address
      f28_load_entry(sx, sy, rx, ry, (it ^ flip) == 0x7fffffffu);
      xyzz28_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  g1_xyzz r; xyzz28_to_xyzz(r, acc);
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= r.x.v[q] ^ r.y.v[q] ^ r.zz.v[q];
  out[t] = x;
}

template <class F> float timeit(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main(int argc, char** argv) {
  const uint32_t waves = argc > 1 ? atoi(argv[1]) : 2048 * 4;  // 256 CUs x 4 SIMDs x 2 waves x 4 rounds
  const uint32_t iters = argc > 2 ? atoi(argv[2]) : 2000;
  uint32_t* out; CHECK(hipMalloc(&out, waves * 64 * 4));
  // points: multiples of the generator built on the host with the 12x32 arithmetic (host path of the same headers)
  const uint32_t npts = 61;
  fp_t h32[2 * npts], h28[2 * npts];
  {
    fp_t gx, gy; const uint32_t tx[12] = KZG_FP_G1X_MONT, ty[12] = KZG_FP_G1Y_MONT;
    for (int i = 0; i < 12; i++) { gx.v[i] = tx[i]; gy.v[i] = ty[i]; }
    g1_xyzz acc; xyzz_set_inf(acc);
    for (uint32_t k = 0; k < npts; k++) {
      xyzz_madd(acc, gx, gy);
      fp_t ax, ay; xyzz_to_affine(ax, ay, acc);
      h32[2 * k] = ax; h32[2 * k + 1] = ay;
      fp_to_r392(h28[2 * k], ax); fp_to_r392(h28[2 * k + 1], ay);
    }
  }
  fp_t *d32, *d28; CHECK(hipMalloc(&d32, sizeof(h32))); CHECK(hipMalloc(&d28, sizeof(h28)));
  CHECK(hipMemcpy(d32, h32, sizeof(h32), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d28, h28, sizeof(h28), hipMemcpyHostToDevice));
  const double n = (double)waves * 64 * iters;
  float ms;
  ms = timeit([&] { k_mul32<<<waves, 64>>>(out, iters); });            printf("mul32  (12x32 lazy)      %8.3f ms  %7.2f G/s\n", ms, n / ms / 1e6);
  ms = timeit([&] { k_mul28<0><<<waves, 64>>>(out, iters); });         printf("mul28  (14x28)           %8.3f ms  %7.2f G/s\n", ms, n / ms / 1e6);
  ms = timeit([&] { k_mul28<1><<<waves, 64>>>(out, iters); });         printf("sqr28+add+carry          %8.3f ms  %7.2f G/s\n", ms, n / ms / 1e6);
  ms = timeit([&] { k_mul28<2><<<waves, 64>>>(out, iters); });         printf("mul2_28 (2 products)     %8.3f ms  %7.2f G/s\n", ms, n / ms / 1e6);
  const uint32_t ai = iters / 4;
  const double na = (double)waves * 64 * ai;
  ms = timeit([&] { k_madd32<<<waves, 64>>>(out, d32, npts, ai); });   printf("madd32 (xyzz_madd_lazy)  %8.3f ms  %7.3f G adds/s\n", ms, na / ms / 1e6);
  uint32_t c32[64]; CHECK(hipMemcpy(c32, out, sizeof(c32), hipMemcpyDeviceToHost));
  ms = timeit([&] { k_madd28<<<waves, 64>>>(out, d28, npts, ai, 0u); });   printf("madd28 (xyzz28_madd)     %8.3f ms  %7.3f G adds/s\n", ms, na / ms / 1e6);
  uint32_t c28[64]; CHECK(hipMemcpy(c28, out, sizeof(c28), hipMemcpyDeviceToHost));
  printf("checksums lane0: %08x %08x (must be equal)\n", c32[0], c28[0]);
  return 0;
}
