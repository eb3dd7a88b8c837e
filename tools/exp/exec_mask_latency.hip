// Experiment (round 4): does a wave with few ACTIVE lanes issue dependent VALU instructions faster?  A wave64 instruction passes
// through a 16-lane SIMD in four cycles; if passes whose EXEC bits are all zero were skipped, a single-blob SHA-256 stream (two
// lanes) could run on a nearly empty wave at a fraction of the 5 cycles per dependent instruction it pays today.
// One wave per workgroup, `waves` workgroups (spread one per SIMD), lanes >= active masked off by EXEC for the whole loop.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/exec_mask_latency.hip -o tools/exp/exec_mask_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t* out, unsigned long long* ticks, uint32_t iters, uint32_t seed, uint32_t active) {
  if (threadIdx.x >= active) return;  // EXEC keeps only the first `active` lanes for everything below
  uint32_t x = seed + threadIdx.x * 7;
  uint32_t y = seed ^ 0x5555, z = seed + 3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 128; u++) {
      if (OP == 0) x = __builtin_amdgcn_alignbit(x, x, 7);
      if (OP == 1) x = __builtin_amdgcn_bitop3_b32(x, y, z, 0x96);
      if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      if (OP == 3) x = x + y;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, uint32_t* out, unsigned long long* d_ticks, int waves, uint32_t active, FILE* js, bool& first) {
  const uint32_t iters = 2000;
  k<OP><<<waves, 64>>>(out, d_ticks, 50, 1, active);
  CHECK(hipDeviceSynchronize());
  k<OP><<<waves, 64>>>(out, d_ticks, iters, 1, active);
  CHECK(hipDeviceSynchronize());
  unsigned long long t[4096];
  CHECK(hipMemcpy(t, d_ticks, waves * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < waves; i++) sum += (double)t[i];
  const double cyc = sum / waves / ((double)iters * 128);
  printf("%-22s active lanes %2u, %4d waves: %.2f cycles per dependent instruction\n", name, active, waves, cyc);
  fprintf(js, "%s{\"op\": \"%s\", \"active_lanes\": %u, \"waves\": %d, \"cycles_per_dependent_instruction\": %.3f}", first ? "" : ",\n ", name, active, waves, cyc);
  first = false;
}
int main(int argc, char** argv) {
  uint32_t* out; CHECK(hipMalloc(&out, 4096 * 64 * 4));
  unsigned long long* d_ticks; CHECK(hipMalloc(&d_ticks, 4096 * 8));
  FILE* js = fopen(argc > 1 ? argv[1] : "exec_mask_latency.json", "w");
  fprintf(js, "{\"what\": \"cycles (s_memtime) per DEPENDENT VALU instruction of a lone wave per SIMD, by the number of lanes EXEC leaves active\", \"rows\": [\n ");
  bool first = true;
  for (int waves : {256, 1024})
    for (uint32_t active : {64u, 32u, 16u, 2u, 1u}) {
      run<0>("v_alignbit_b32", out, d_ticks, waves, active, js, first);
      run<1>("v_bitop3_b32", out, d_ticks, waves, active, js, first);
      run<2>("v_add3_u32", out, d_ticks, waves, active, js, first);
      run<3>("v_add_u32", out, d_ticks, waves, active, js, first);
    }
  fprintf(js, "\n]}\n");
  fclose(js);
  return 0;
}
