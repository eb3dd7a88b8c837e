// Experiment (round 4): what a lone wave per SIMD pays per instruction for the opcodes of the SHA-256 rounds, DEPENDENT and
// INDEPENDENT -- is v_add3_u32's 8 cycles an issue cost or a latency that independent work can hide?
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/lone_wave_issue.hip -o tools/exp/lone_wave_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int P>
__global__ __launch_bounds__(64) void k(uint32_t* out, unsigned long long* ticks, uint32_t iters, uint32_t seed) {
  uint32_t x = seed + threadIdx.x * 7, w = seed * 3 + threadIdx.x, v = seed ^ threadIdx.x;
  uint32_t y = seed ^ 0x5555, z = seed + 3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 64; u++) {
      if (P == 0) asm volatile("v_add3_u32 %0, %0, %1, %2\n\tv_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      if (P == 1) asm volatile("v_add3_u32 %0, %0, %2, %3\n\tv_add3_u32 %1, %1, %2, %3" : "+v"(x), "+v"(w) : "v"(y), "v"(z));
      if (P == 2) asm volatile("v_add3_u32 %0, %0, %2, %3\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96" : "+v"(x), "+v"(w) : "v"(y), "v"(z));
      if (P == 3) asm volatile("v_add_u32_e32 %0, %0, %1\n\tv_add_u32_e32 %0, %0, %1" : "+v"(x) : "v"(y));
      if (P == 4) asm volatile("v_add_u32_dpp %0, %0, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_bitop3_b32 %1, %1, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %3, %4 bitop3:0x96" : "+v"(x), "+v"(w), "+v"(v) : "v"(y), "v"(z));
      if (P == 5) asm volatile("v_alignbit_b32 %0, %0, %0, %1\n\tv_alignbit_b32 %0, %0, %0, %1" : "+v"(x) : "v"(y));
      if (P == 6) asm volatile("v_add3_u32 %0, %0, %3, %4\n\tv_bitop3_b32 %1, %1, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %3, %4 bitop3:0x96" : "+v"(x), "+v"(w), "+v"(v) : "v"(y), "v"(z));
      if (P == 7) asm volatile("v_bitop3_b32 %0, %0, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %1, %1, %3, %4 bitop3:0x96\n\tv_bitop3_b32 %2, %2, %3, %4 bitop3:0x96" : "+v"(x), "+v"(w), "+v"(v) : "v"(y), "v"(z));
      // explicit registers: do three VGPR sources from ONE register bank (index mod 4) cost more than three from different banks?
      if (P == 9) asm volatile("v_bitop3_b32 v40, v40, v44, v48 bitop3:0x96\n\tv_bitop3_b32 v40, v40, v44, v48 bitop3:0x96\n\tv_bitop3_b32 v40, v40, v44, v48 bitop3:0x96" ::: "v40", "v44", "v48");
      if (P == 10) asm volatile("v_bitop3_b32 v40, v40, v45, v50 bitop3:0x96\n\tv_bitop3_b32 v40, v40, v45, v50 bitop3:0x96\n\tv_bitop3_b32 v40, v40, v45, v50 bitop3:0x96" ::: "v40", "v45", "v50");
      if (P == 11) asm volatile("v_add3_u32 v40, v40, v44, v48\n\tv_add3_u32 v40, v40, v44, v48\n\tv_add3_u32 v40, v40, v44, v48" ::: "v40", "v44", "v48");
      if (P == 12) asm volatile("v_add3_u32 v40, v40, v45, v50\n\tv_add3_u32 v40, v40, v45, v50\n\tv_add3_u32 v40, v40, v45, v50" ::: "v40", "v45", "v50");
      if (P == 13) asm volatile("v_alignbit_b32 v40, v44, v44, v48\n\tv_alignbit_b32 v41, v44, v44, v48\n\tv_alignbit_b32 v42, v44, v44, v48" ::: "v40", "v41", "v42", "v44", "v48");
      if (P == 14) asm volatile("v_alignbit_b32 v40, v44, v44, v49\n\tv_alignbit_b32 v41, v44, v44, v49\n\tv_alignbit_b32 v42, v44, v44, v49" ::: "v40", "v41", "v42", "v44", "v49");
      if (P == 8) asm volatile("v_add_u32_e32 %0, %0, %3\n\tv_add_u32_e32 %0, %0, %4\n\tv_bitop3_b32 %1, %1, %3, %4 bitop3:0x96" : "+v"(x), "+v"(w), "+v"(v) : "v"(y), "v"(z));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = x + w + v;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int P>
void run(const char* name, int per_group, uint32_t* out, unsigned long long* d_ticks, FILE* js, bool& first) {
  const int waves = 1024;
  const uint32_t iters = 2000;
  k<P><<<waves, 64>>>(out, d_ticks, 50, 1);
  CHECK(hipDeviceSynchronize());
  k<P><<<waves, 64>>>(out, d_ticks, iters, 1);
  CHECK(hipDeviceSynchronize());
  unsigned long long t[4096];
  CHECK(hipMemcpy(t, d_ticks, waves * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < waves; i++) sum += (double)t[i];
  const double cyc = sum / waves / ((double)iters * 64);
  printf("%-70s %.2f cycles per group of %d (%.2f per instruction)\n", name, cyc, per_group, cyc / per_group);
  fprintf(js, "%s{\"pattern\": \"%s\", \"instructions_per_group\": %d, \"cycles_per_group\": %.3f}", first ? "" : ",\n ", name, per_group, cyc);
  first = false;
}
int main(int argc, char** argv) {
  uint32_t* out; CHECK(hipMalloc(&out, 4096 * 64 * 4));
  unsigned long long* d_ticks; CHECK(hipMalloc(&d_ticks, 4096 * 8));
  FILE* js = fopen(argc > 1 ? argv[1] : "lone_wave_issue.json", "w");
  fprintf(js, "{\"what\": \"s_memtime ticks of a lone wave per SIMD (1,024 waves of 64, launch_bounds(64)) per group of instructions\", \"rows\": [\n ");
  bool first = true;
  run<0>("add3 -> add3 (dependent)", 2, out, d_ticks, js, first);
  run<1>("add3 a ; add3 b (independent chains)", 2, out, d_ticks, js, first);
  run<2>("add3 a ; bitop3 b (independent)", 2, out, d_ticks, js, first);
  run<6>("add3 a ; bitop3 b ; bitop3 c (independent)", 3, out, d_ticks, js, first);
  run<7>("bitop3 a ; bitop3 b ; bitop3 c (independent)", 3, out, d_ticks, js, first);
  run<8>("add a ; add a (dependent) ; bitop3 b", 3, out, d_ticks, js, first);
  run<3>("add -> add (dependent)", 2, out, d_ticks, js, first);
  run<4>("add_dpp row_half_mirror a ; bitop3 b ; bitop3 c", 3, out, d_ticks, js, first);
  run<5>("alignbit -> alignbit (dependent)", 2, out, d_ticks, js, first);
  run<9>("bitop3 x3, sources v40 v44 v48 (one bank)", 3, out, d_ticks, js, first);
  run<10>("bitop3 x3, sources v40 v45 v50 (three banks)", 3, out, d_ticks, js, first);
  run<11>("add3 x3, sources v40 v44 v48 (one bank)", 3, out, d_ticks, js, first);
  run<12>("add3 x3, sources v40 v45 v50 (three banks)", 3, out, d_ticks, js, first);
  run<13>("alignbit x3, sources v44 v44 v48 (one bank)", 3, out, d_ticks, js, first);
  run<14>("alignbit x3, sources v44 v44 v49 (two banks)", 3, out, d_ticks, js, first);
  fprintf(js, "\n]}\n");
  fclose(js);
  return 0;
}
