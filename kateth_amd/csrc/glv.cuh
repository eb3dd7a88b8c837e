// GLV split of a scalar for the variable-base lincombs of batch verification (round 5).
//
// On G1 the endomorphism phi(x, y) = (beta x, y) is multiplication by -z^2 (g1.cuh; z the BLS parameter, r = z^4 - z^2 + 1),
// so [z^2](x, y) = (beta x, -y) costs one field product.  A scalar k < r splits as k = k1 + k2 z^2 with k1 = k mod z^2 and
// k2 = floor(k / z^2), both below z^2 < 2^128, and
//     [k] P = [k1] P + [k2] (beta x, -y):
// twice the terms with half the windows -- the same number of bucket additions, HALF the bit sums (k_var_bitsums) and half the
// doublings of the host's Horner loop, i.e. half of what follows the buckets in every lincomb (src/kzg/setup.rs:152-155).
#pragma once
#include "field.cuh"

namespace kzg {

// k (8 x 32-bit limbs, plain, k < r) -> k1 = k mod z^2, k2 = floor(k / z^2), each as 8 limbs (upper four zero)
KZG_HD void glv_split(fr_t& k1, fr_t& k2, const fr_t& k) {
  constexpr uint32_t Z2[4] = KZG_GLV_Z2;
  constexpr uint32_t MU[5] = KZG_GLV_MU;  // floor(2^256 / z^2), 129 bits
  // Barrett: q = floor((k >> 126) * mu >> 130) is floor(k / z^2) or up to 2 less
  uint32_t kh[5];  // k >> 126: 130 bits
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int w = 3 + i;  // bit 126 = word 3, bit 30
    const uint32_t lo = (w < 8) ? k.v[w < 8 ? w : 7] : 0u;
    const uint32_t hi = (w + 1 < 8) ? k.v[w + 1 < 8 ? w + 1 : 7] : 0u;
    kh[i] = (lo >> 30) | (hi << 2);
  }
  uint32_t prod[10];
#pragma unroll
  for (int i = 0; i < 10; i++) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint64_t t = (uint64_t)kh[i] * MU[j] + prod[i + j] + c;
      prod[i + j] = (uint32_t)t;
      c = t >> 32;
    }
    prod[i + 5] = (uint32_t)c;
  }
  uint32_t q[5];  // prod >> 130: word 4, bit 2
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int w = 4 + i;
    const uint32_t lo = prod[w];
    const uint32_t hi = (w + 1 < 10) ? prod[w + 1 < 10 ? w + 1 : 9] : 0u;
    q[i] = (lo >> 2) | (hi << 30);
  }
  // rem = k - q z^2, low 160 bits (the true remainder is < 3 z^2 < 2^130)
  uint32_t qz[5];
#pragma unroll
  for (int i = 0; i < 5; i++) qz[i] = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i + j < 5) {
        const uint64_t t = (uint64_t)q[i] * Z2[j] + qz[i + j] + c;
        qz[i + j] = (uint32_t)t;
        c = t >> 32;
      }
    }
    if (i + 4 < 5) qz[i + 4] += (uint32_t)c;
  }
  uint32_t rem[5];
  {
    uint64_t b = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint64_t t = (uint64_t)k.v[i] - qz[i] - b;
      rem[i] = (uint32_t)t;
      b = (t >> 32) & 1u;
    }
  }
  for (int round = 0; round < 3; round++) {  // at most two corrections; a third round costs nothing when it is not taken
    uint32_t d[5];
    uint64_t b = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint64_t t = (uint64_t)rem[i] - (i < 4 ? Z2[i] : 0u) - b;
      d[i] = (uint32_t)t;
      b = (t >> 32) & 1u;
    }
    if (b == 0) {  // rem >= z^2
#pragma unroll
      for (int i = 0; i < 5; i++) rem[i] = d[i];
      uint64_t c = 1;
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const uint64_t t = (uint64_t)q[i] + c;
        q[i] = (uint32_t)t;
        c = t >> 32;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    k1.v[i] = i < 4 ? rem[i] : 0u;
    k2.v[i] = i < 4 ? q[i] : 0u;
  }
}

}  // namespace kzg
