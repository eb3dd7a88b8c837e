// device helper shared by the kernel headers (no kernels here)
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace kzg {
#if defined(__HIPCC__)
// 32 big-endian bytes (16-B aligned) -> 8 plain little-endian limbs
__device__ __forceinline__ void load_scalar_be_(uint32_t* sc, const uint8_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 w0 = q[0], w1 = q[1];
  sc[7] = __builtin_bswap32(w0.x);
  sc[6] = __builtin_bswap32(w0.y);
  sc[5] = __builtin_bswap32(w0.z);
  sc[4] = __builtin_bswap32(w0.w);
  sc[3] = __builtin_bswap32(w1.x);
  sc[2] = __builtin_bswap32(w1.y);
  sc[1] = __builtin_bswap32(w1.z);
  sc[0] = __builtin_bswap32(w1.w);
}

#endif
}  // namespace kzg
