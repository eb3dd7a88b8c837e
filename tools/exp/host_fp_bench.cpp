// host microbenchmark: the portable unsigned-__int128 Montgomery product of field.cuh against the mulx / adcx / adox one
// (host_fp_mulx.hpp), dependent chains, and agreement on random operands.   clang++ -O3 -std=c++17 tools/exp/host_fp_bench.cpp
#include <chrono>
#include <stdio.h>
#include <string.h>

#include "../../kateth_amd/csrc/field.cuh"
#include "../../kateth_amd/csrc/host_fp_mulx.hpp"
using namespace kzg;

int main() {
  fp_t a = fp_one(), b = fp_one();
  for (int i = 0; i < 12; i++) {
    a.v[i] = 0x12345u * (i + 3) + 77u;
    b.v[i] = 0xabcdefu * (i + 5) + 1u;
  }
  a.v[11] &= 0x0fffffffu;
  b.v[11] &= 0x0fffffffu;
  // agreement (mod p: the mulx product is < 2p, not canonical)
  int bad = 0;
  fp_t x = a, y = b;
  for (int it = 0; it < 200000; it++) {
    fp_t r1, r2;
    mont_mul_host64<FpParams, false>(r1, x, y);
    uint64_t t[6];
    hostmulx::mont_mul_384(t, (const uint64_t*)x.v, (const uint64_t*)y.v);
    memcpy(r2.v, t, 48);
    fp_t p;
    for (int i = 0; i < 12; i++) p.v[i] = FpParams::mod(i);
    if (bn_geq(r2, p)) bn_sub(r2, r2, p);
    if (!bn_eq(r1, r2)) bad++;
    x = y;
    y = r1;
    if ((it & 1023) == 0) y.v[0] ^= it;
  }
  printf("mismatches %d\n", bad);
  const int N = 5000000;
  auto t0 = std::chrono::steady_clock::now();
  fp_t z = a;
  for (int i = 0; i < N; i++) mont_mul_host64<FpParams, false>(z, z, b);
  auto t1 = std::chrono::steady_clock::now();
  uint64_t w[6];
  memcpy(w, a.v, 48);
  for (int i = 0; i < N; i++) hostmulx::mont_mul_384(w, w, (const uint64_t*)b.v);
  auto t2 = std::chrono::steady_clock::now();
  printf("int128 CIOS %.1f ns per product; mulx/adx %.1f ns (chk %08x %08x)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N,
         std::chrono::duration<double, std::nano>(t2 - t1).count() / N, z.v[0], (unsigned)w[0]);
  return bad != 0;
}
