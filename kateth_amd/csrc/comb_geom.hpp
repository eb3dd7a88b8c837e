// Geometry of the fixed-base subset-sum comb (msm_comb.cuh): plain host/device data and index arithmetic, no kernels -- shared by
// the host code of every translation unit; the kernels themselves are compiled ONCE (engine.hip owns msm_comb.cuh).
#pragma once
#include <stdint.h>

#include "field.cuh"

namespace kzg {

struct CombGeom {
  uint32_t nb;    // blocks per 64 points: 3 (22 + 21 + 21 points), 4 (16 each), 8 (8 each), 16 (4 each)
  uint32_t G;     // plane groups = tables; H = 256 / G planes each
  uint32_t H;
  uint32_t lpg;   // lanes per group = 64 / G
  uint32_t ep64;  // table entries per 64 points (one group): sum over the blocks of 2^(t-1)
  uint32_t epg;   // entries per group = 64 * ep64
  uint32_t fair;  // s > 0: the two waves of a SIMD trade issue priority every 2^s shader cycles (k_msm_comb30); 0: hardware default (oldest first)
};

KZG_HD uint32_t comb_tbits(uint32_t nb, uint32_t r) { return nb == 3u ? (r == 0u ? 22u : 21u) : 64u / nb; }
KZG_HD uint32_t comb_point_off(uint32_t nb, uint32_t r) { return nb == 3u ? (r == 0u ? 0u : (r == 1u ? 22u : 43u)) : r * (64u / nb); }
KZG_HD uint32_t comb_entry_off(uint32_t nb, uint32_t r) {
  return nb == 3u ? (r == 0u ? 0u : (r == 1u ? (1u << 21) : (1u << 21) + (1u << 20))) : r << (64u / nb - 1u);
}
KZG_HD CombGeom comb_make_geom(uint32_t nb, uint32_t G) {
  CombGeom g;
  g.nb = nb;
  g.G = G;
  g.H = 256u / G;
  g.lpg = 64u / G;
  g.ep64 = nb == 3u ? (1u << 22) : nb << (64u / nb - 1u);
  g.epg = 64u * g.ep64;
  g.fair = 20u;
  return g;
}
KZG_HD uint64_t comb_table_entries(const CombGeom& g) { return (uint64_t)g.G * g.epg; }
// largest `splits` the geometry supports: a lane must own a whole number of blocks
KZG_HD uint32_t comb_max_splits(const CombGeom& g) { return (64u * g.nb) / g.lpg; }

}  // namespace kzg
