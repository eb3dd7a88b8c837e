// Experiment (round 5): the signed radix-2^30 field (fp30.cuh: 13 limbs, 169 + 169 v_mad_i64_i32 per product)
// against the radix-2^28 one of the hot loop (fp28.cuh: 14 limbs, 196 + 196 v_mad_u64_u32) -- product, squaring, double product
// and the full mixed addition with table-format operands, whole chip, two waves per SIMD, same points, same launch shape.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/fp30_bench.hip -o /tmp/fp30_bench ; run: /tmp/fp30_bench [json]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../kateth_amd/csrc/fp28.cuh"
#include "../../kateth_amd/csrc/fp30.cuh"
using namespace kzg;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(64, 2) void k_mul28(uint32_t* out, uint32_t iters) {
  fp28 a, b;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  for (int q = 0; q < 14; q++) { a.l[q] = f28_one_limb(q) ^ (t & 0xffu); b.l[q] = f28_r384_limb(q); }
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    fp28 r;
    if (MODE == 0) f28_mul(r, a, b);
    if (MODE == 1) f28_sqr(r, a);
    if (MODE == 2) { fp28 c = f28_one(); f28_mul2(r, a, b, c, a); }
    a = b; b = r;
  }
  uint32_t x = 0; for (int q = 0; q < 14; q++) x ^= b.l[q];
  out[t] = x;
}
template <int MODE>
__global__ __launch_bounds__(64, 2) void k_mul30(uint32_t* out, uint32_t iters) {
  fp30 a, b;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  for (int q = 0; q < 13; q++) { a.l[q] = f30_one_limb(q) ^ (int32_t)(t & 0xffu); b.l[q] = f30_r384_limb(q); }
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    fp30 r;
    if (MODE == 0) f30_mul(r, a, b);
    if (MODE == 1) f30_sqr(r, a);
    if (MODE == 2) { fp30 c = f30_one(); f30_mul2(r, a, b, c, a); }
    if (MODE == 3) { f30_sub(r, a, b); f30_carry(r); }
    a = b; b = r;
  }
  uint32_t x = 0; for (int q = 0; q < 13; q++) x ^= (uint32_t)b.l[q];
  out[t] = x;
}
// adder loops: acc += +-P_k with the operand changing every step (61 multiples of the generator, index and sign from the step
// counter and the lane), entries in each kernel's table format, unpacked per step as the MSM kernel does
__global__ __launch_bounds__(64, 2) void k_madd28(uint32_t* out, const fp_t* pts, uint32_t npts, uint32_t iters) {
  g1_xyzz28 acc; xyzz28_set_inf(acc);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t k = (it * 7u + t) % npts;
    const bool neg = ((it * 0x9e3779b9u + t) >> 13) & 1u;
    fp_t x = pts[2 * k], y = pts[2 * k + 1];
    fp28 x2, y2;
    f28_load_entry(x2, y2, x, y, neg);
    if (acc.inf || !xyzz28_madd_fast(acc, x2, y2)) {
      g1_xyzz28 tmp = acc;
      fp_t rx = pts[2 * k], ry = pts[2 * k + 1];
      fp28 sx, sy;
      f28_load_entry(sx, sy, rx, ry, neg);
      xyzz28_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  g1_xyzz r; xyzz28_to_xyzz(r, acc);
  fp_t yz; fp_mul(yz, r.y, r.zzz);  // (X, Y, ZZ, ZZZ) and (X, -Y, ZZ, -ZZZ) are the same point: the fp30 adder may hold either
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= r.x.v[q] ^ yz.v[q] ^ r.zz.v[q];
  out[t] = x;
}
__global__ __launch_bounds__(64, 2) void k_madd30(uint32_t* out, const fp_t* pts, uint32_t npts, uint32_t iters) {
  g1_xyzz30 acc; xyzz30_set_inf(acc);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t k = (it * 7u + t) % npts;
    const bool neg = ((it * 0x9e3779b9u + t) >> 13) & 1u;
    fp_t x = pts[2 * k], y = pts[2 * k + 1];  // packed 30-bit digits: 12 words each
    fp30 x2, y2;
    f30_load_entry(x2, y2, x.v, y.v, xyzz30_entry_neg(acc, neg));
    if (acc.inf || !xyzz30_madd_fast(acc, x2, y2)) {
      g1_xyzz30 tmp = acc;
      fp_t rx = pts[2 * k], ry = pts[2 * k + 1];
      fp30 sx, sy;
      f30_load_entry(sx, sy, rx.v, ry.v, xyzz30_entry_neg(tmp, neg));
      xyzz30_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  g1_xyzz r; xyzz30_to_xyzz(r, acc);
  fp_t yz; fp_mul(yz, r.y, r.zzz);  // (X, Y, ZZ, ZZZ) and (X, -Y, ZZ, -ZZZ) are the same point: the fp30 adder may hold either
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= r.x.v[q] ^ yz.v[q] ^ r.zz.v[q];
  out[t] = x;
}

// the same with PRE-UNPACKED table entries: 13 + 13 int32 digits = 104 bytes per entry instead of 96 (no alignbit / bfe per limb;
// the negation stays) -- what a wider table entry would buy (second half)
__global__ __launch_bounds__(64, 2) void k_madd30_unpacked(uint32_t* out, const int32_t* pts, uint32_t npts, uint32_t iters) {
  g1_xyzz30 acc; xyzz30_set_inf(acc);
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  auto load = [&](fp30& x2, fp30& y2, uint32_t k, bool neg) {
    const int32_t* e = pts + 26 * k;
    const uint32_t m = neg ? 0xffffffffu : 0u, one = neg ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 13; i++) {
      x2.l[i] = e[i];
      y2.l[i] = (int32_t)(((uint32_t)e[13 + i] ^ m) + one);
    }
  };
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
    const uint32_t k = (it * 7u + t) % npts;
    const bool neg = ((it * 0x9e3779b9u + t) >> 13) & 1u;
    fp30 x2, y2;
    load(x2, y2, k, xyzz30_entry_neg(acc, neg));
    if (acc.inf || !xyzz30_madd_fast(acc, x2, y2)) {
      g1_xyzz30 tmp = acc;
      fp30 sx, sy;
      load(sx, sy, k, xyzz30_entry_neg(tmp, neg));
      xyzz30_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  g1_xyzz r; xyzz30_to_xyzz(r, acc);
  fp_t yz; fp_mul(yz, r.y, r.zzz);  // (X, Y, ZZ, ZZZ) and (X, -Y, ZZ, -ZZZ) are the same point: the fp30 adder may hold either
  uint32_t x = 0; for (int q = 0; q < 12; q++) x ^= r.x.v[q] ^ yz.v[q] ^ r.zz.v[q];
  out[t] = x;
}

template <class F> float timeit(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const uint32_t waves = (uint32_t)prop.multiProcessorCount * 8;  // two waves per SIMD, one round
  const uint32_t iters = 3000;
  uint32_t* out; CHECK(hipMalloc(&out, waves * 64 * 4));
  const uint32_t npts = 61;
  fp_t h28[2 * npts], h30[2 * npts];
  {
    fp_t gx, gy; const uint32_t tx[12] = KZG_FP_G1X_MONT, ty[12] = KZG_FP_G1Y_MONT;
    for (int i = 0; i < 12; i++) { gx.v[i] = tx[i]; gy.v[i] = ty[i]; }
    g1_xyzz acc; xyzz_set_inf(acc);
    for (uint32_t k = 0; k < npts; k++) {
      xyzz_madd(acc, gx, gy);
      fp_t ax, ay; xyzz_to_affine(ax, ay, acc);
      fp_to_r392(h28[2 * k], ax); fp_to_r392(h28[2 * k + 1], ay);
      fp_to_packed30(h30[2 * k].v, ax); fp_to_packed30(h30[2 * k + 1].v, ay);
    }
  }
  int32_t h30u[26 * npts];
  for (uint32_t k = 0; k < npts; k++) {
    fp30 ux, uy;
    f30_unpack(ux, h30[2 * k].v);
    f30_unpack(uy, h30[2 * k + 1].v);
    for (int i = 0; i < 13; i++) { h30u[26 * k + i] = ux.l[i]; h30u[26 * k + 13 + i] = uy.l[i]; }
  }
  int32_t* d30u; CHECK(hipMalloc(&d30u, sizeof(h30u))); CHECK(hipMemcpy(d30u, h30u, sizeof(h30u), hipMemcpyHostToDevice));
  fp_t *d28, *d30; CHECK(hipMalloc(&d28, sizeof(h28))); CHECK(hipMalloc(&d30, sizeof(h30)));
  CHECK(hipMemcpy(d28, h28, sizeof(h28), hipMemcpyHostToDevice)); CHECK(hipMemcpy(d30, h30, sizeof(h30), hipMemcpyHostToDevice));
  const double n = (double)waves * 64 * iters;
  double r[10];
  r[0] = n / timeit([&] { k_mul28<0><<<waves, 64>>>(out, iters); }) / 1e6;
  r[1] = n / timeit([&] { k_mul30<0><<<waves, 64>>>(out, iters); }) / 1e6;
  r[2] = n / timeit([&] { k_mul28<1><<<waves, 64>>>(out, iters); }) / 1e6;
  r[3] = n / timeit([&] { k_mul30<1><<<waves, 64>>>(out, iters); }) / 1e6;
  r[4] = n / timeit([&] { k_mul28<2><<<waves, 64>>>(out, iters); }) / 1e6;
  r[5] = n / timeit([&] { k_mul30<2><<<waves, 64>>>(out, iters); }) / 1e6;
  r[9] = n / timeit([&] { k_mul30<3><<<waves, 64>>>(out, iters); }) / 1e6;
  const uint32_t ai = 1536;  // one lane's additions per blob in half-wave mode
  const double na = (double)waves * 64 * ai;
  r[6] = na / timeit([&] { k_madd28<<<waves, 64>>>(out, d28, npts, ai); }) / 1e6;
  uint32_t c28[64]; CHECK(hipMemcpy(c28, out, sizeof(c28), hipMemcpyDeviceToHost));
  r[7] = na / timeit([&] { k_madd30<<<waves, 64>>>(out, d30, npts, ai); }) / 1e6;
  uint32_t c30[64]; CHECK(hipMemcpy(c30, out, sizeof(c30), hipMemcpyDeviceToHost));
  r[8] = na / timeit([&] { k_madd30_unpacked<<<waves, 64>>>(out, d30u, npts, ai); }) / 1e6;
  uint32_t c30u[64]; CHECK(hipMemcpy(c30u, out, sizeof(c30u), hipMemcpyDeviceToHost));
  int same = 1; for (int i = 0; i < 64; i++) same &= c28[i] == c30[i] && c30u[i] == c30[i];
  printf("product        fp28 %7.2f G/s   fp30 %7.2f G/s  (%+.1f %%)\n", r[0], r[1], 100 * (r[1] / r[0] - 1));
  printf("squaring       fp28 %7.2f G/s   fp30 %7.2f G/s  (%+.1f %%)\n", r[2], r[3], 100 * (r[3] / r[2] - 1));
  printf("double product fp28 %7.2f G/s   fp30 %7.2f G/s  (%+.1f %%)\n", r[4], r[5], 100 * (r[5] / r[4] - 1));
  printf("sub + carry pass (fp30 only) %7.2f G/s\n", r[9]);
  printf("mixed addition fp28 %7.3f G/s   fp30 %7.3f G/s  (%+.1f %%)   results %s\n", r[6], r[7], 100 * (r[7] / r[6] - 1), same ? "IDENTICAL" : "DIFFER");
  printf("mixed addition fp30, pre-unpacked 104-byte entries %7.3f G/s  (%+.1f %% over packed 96-byte entries)\n", r[8], 100 * (r[8] / r[7] - 1));
  FILE* js = fopen(argc > 1 ? argv[1] : "fp30_bench.json", "w");
  fprintf(js, "{\"device\": \"%s\", \"waves\": %u, \"waves_per_simd\": 2, \"g_per_s\": {\"product_fp28\": %.3f, \"product_fp30\": %.3f, \"squaring_fp28\": %.3f, "
          "\"squaring_fp30\": %.3f, \"double_product_fp28\": %.3f, \"double_product_fp30\": %.3f, \"sub_and_carry_fp30\": %.3f, \"mixed_addition_fp28\": %.4f, "
          "\"mixed_addition_fp30\": %.4f, \"mixed_addition_fp30_unpacked_104B_entries\": %.4f}, \"mixed_addition_gain\": %.4f, \"unpacked_entry_gain_over_packed\": %.4f, "
          "\"same_results\": %s}\n",
          prop.gcnArchName, waves, r[0], r[1], r[2], r[3], r[4], r[5], r[9], r[6], r[7], r[8], r[7] / r[6] - 1, r[8] / r[7] - 1, same ? "true" : "false");
  fclose(js);
  return same ? 0 : 1;
}
