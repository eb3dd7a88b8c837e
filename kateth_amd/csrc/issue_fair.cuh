// Fair VALU issue between the waves that share a SIMD.
//
// gfx950's instruction arbiter serves the OLDEST ready wave first.  For kernels whose waves run the whole launch in one
// round at two waves per SIMD (k_msm_comb30 at 4,096 blobs, k_g1_decompress at 131,072 points) that starves the younger wave:
// measured per wave (tools/gpu_wave_times.py), the older wave of every pair ran at its solo rate and left after 60 % of
// the launch, and the younger one finished the rest alone at 5.3 instead of 4.1 cycles per instruction (the SIMD's shared rate).
// issue_fair_tick() makes the waves trade priority every 2^shift shader cycles, keyed on the parity of the wave's slot
// (the two waves of a SIMD sit in slots 0 and 1), so they advance at the same average rate and end together.  Call it
// once per loop iteration; the period must be long against one iteration of a low-priority wave (which takes up to 4x as
// long as a high-priority one, so it notices its turn late) and short against the kernel.  It changes no result.
#pragma once
#include "field.cuh"

namespace kzg {

KZG_HD void issue_fair_tick(uint32_t shift) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t parity = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;  // hwreg(HW_REG_HW_ID, 0, 4): wave slot
  if ((((uint32_t)(__builtin_amdgcn_s_memtime() >> shift)) & 1u) == parity)
    __builtin_amdgcn_s_setprio(3);
  else
    __builtin_amdgcn_s_setprio(1);  // not 0: a kernel that ticks stays above one that does not (the point decoder's long chains beside the evaluation kernel)
#else
  (void)shift;
#endif
}

// The fixed-base MSM's variant: levels 2 / 0.  Its waves live for a whole launch (7-29 ms); the short, latency-bound kernels
// that may run beside them on another stream (hash, quotient, bit-plane transposition, lane-sum trees, encoding -- the chunk
// pipelines of the host-buffer entry points) raise themselves to 3 with issue_priority_latency() and so never wait behind an
// MSM wave's instruction stream.
KZG_HD void issue_fair_tick_low(uint32_t shift) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t parity = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
  if ((((uint32_t)(__builtin_amdgcn_s_memtime() >> shift)) & 1u) == parity)
    __builtin_amdgcn_s_setprio(2);
  else
    __builtin_amdgcn_s_setprio(0);
#else
  (void)shift;
#endif
}
KZG_HD void issue_priority_latency() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);
#endif
}

}  // namespace kzg
