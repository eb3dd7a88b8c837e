// Experiment: dependent-issue rate of the VALU instructions SHA-256 is made of, one wave per SIMD on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/valu_latency.hip -o tools/exp/valu_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP, int CHAINS>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t iters, uint32_t seed) {
  uint32_t x[CHAINS];
  for (int c = 0; c < CHAINS; c++) x[c] = seed + threadIdx.x * 7 + c * 0x9e3779b9u;
  uint32_t y = seed ^ 0x5555, z = seed + 3;
#pragma unroll 1
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 64; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        if (OP == 0) x[c] = __builtin_amdgcn_alignbit(x[c], x[c], 7);
        if (OP == 1) x[c] = __builtin_amdgcn_bitop3_b32(x[c], y, z, 0x96);
        if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z));
        if (OP == 3) x[c] = x[c] + y;
        if (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "s"(0x428a2f98));
      }
    }
  }
  uint32_t r = 0;
  for (int c = 0; c < CHAINS; c++) r ^= x[c];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int OP, int CHAINS>
void run(const char* name, uint32_t* out, int waves) {
  const uint32_t iters = 20000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP, CHAINS><<<waves, 64>>>(out, 100, 1); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); k<OP, CHAINS><<<waves, 64>>>(out, iters, 1); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)iters * 64 * CHAINS;
  printf("%-28s chains=%d waves=%d: %.3f ms  %.2f ns per instr per wave\n", name, CHAINS, waves, ms, ms * 1e6 / instr);
}
int main() {
  uint32_t* out; CHECK(hipMalloc(&out, 4096 * 64 * 4));
  for (int waves : {1024, 2048}) {
    run<0, 1>("v_alignbit dependent", out, waves); run<0, 2>("v_alignbit", out, waves);
    run<1, 1>("v_bitop3 dependent", out, waves);   run<1, 2>("v_bitop3", out, waves);
    run<2, 1>("v_add3 dependent", out, waves);     run<2, 2>("v_add3", out, waves);
    run<3, 1>("v_add dependent", out, waves);      run<3, 2>("v_add", out, waves);
    run<4, 1>("v_add3 +sgpr literal dep", out, waves);
  }
  return 0;
}
