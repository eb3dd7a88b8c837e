// Fixed-base 4096-point G1 MSM as a SUBSET-SUM COMB (kernel K2 of SURVEY.md section 2b): replaces
// P1::lincomb_pippenger(setup.g1_lagrange_brp, scalars) (src/bls.rs:416-437, called from src/blob.rs:48-53 and
// src/kzg/poly.rs:68).
//
// Round 1 took the windowed method to its fixed-base limit (now tests/window_msm/window_msm.cuh: every signed c-bit multiple
// of every window base, 192 GiB at c = 16, 65,536 additions per blob).  A table entry there serves ONE point; here an entry serves a BLOCK
// of t points, which is t times more memory-efficient per index bit:
//   * scalars are recoded to signed bits:  2e = sum_k s_k 2^k + (2^256 - 1),  s_k = 2 b_k - 1 = +-1  (k = 0..255), so
//       sum_i e_i L_i = sum_k 2^k sum_i s_{i,k} (L_i / 2)  +  [(2^256 - 1)/2] sum_i L_i ,
//     and sum_i L_i is the G1 generator for a Lagrange basis (it sums to 1): the last term is the constant point K = [c0] sum_i L_i
//     (computed from the actual sum at context creation; one lane per blob starts from it);
//   * the 4096 points are cut into blocks of t consecutive points (per 64 points: 22 + 21 + 21, or 4 x 16, 8 x 8, 16 x 4);
//     for a block the table holds every sign combination  S[m] = sum_p s_p(m) (L_p / 2)  with the top sign fixed to -1
//     (S[~m] = -S[m]: a negation is free), 2^(t-1) affine entries;
//   * bit plane k of a blob is then ONE table lookup + mixed addition per block: 256 planes x 192 blocks = 49,152
//     additions per blob at t = 22/21 instead of 65,536 -- with a 25.8 GB table instead of 192 GiB;
//   * the planes are combined by Horner's rule inside a lane (one doubling of the lane's accumulator per plane).  To keep
//     that overhead small the planes are cut into G groups of H = 256/G, each with its OWN table built on 2^(H g) L_i/2,
//     so a lane only walks H planes (H - 1 doublings against H x blocks-per-lane additions) and the lane sums of a
//     blob add up without any scaling: G = 8 (bench.py's class 22) -> 192 GiB and 31 doublings per 768 additions of a
//     64-lane blob (1,536 in half-wave mode); G = 4 -> 96 GiB, 63 doublings, within 1 % of G = 8.
// The bit planes come from a transposition kernel (k_comb_transpose: the blob as a 256 x 64 array of 64-bit masks, one
// mask = bit k of 64 consecutive scalars; it also performs Blob::from_slice's canonicity check, src/blob.rs:26-37).
//
// Work decomposition: one wave per (blob, split), or per PAIR of blobs from num_CUs x 16 blobs per launch on (half-wave
// mode: 32 lanes per blob); lane = (plane group, block owner).  The hot loop is the mixed addition
// (xyzz30_madd_fast, fp30.cuh: signed radix 2^30 since round 5; the table holds packed centred digits) with the next table entry gathered while the current addition runs, the masks
// of a lane's next four chunks fetched by one 32-byte load, and the two waves of a SIMD trading issue priority
// (issue_fair.cuh) so that they finish together.
#pragma once
#include "comb_geom.hpp"
#include "fp30.cuh"
#include "issue_fair.cuh"
#include "msm_fixed.cuh"

namespace kzg {

#if defined(__HIPCC__)

// 64 x 64 bit-matrix transposition across the lanes of a wave: lane i holds row i (bit j = column j); afterwards lane j
// holds column j.  Six butterfly stages, two cross-lane permutes each.
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane) {
  constexpr uint64_t LO[6] = {0x00000000FFFFFFFFull, 0x0000FFFF0000FFFFull, 0x00FF00FF00FF00FFull,
                              0x0F0F0F0F0F0F0F0Full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
  for (int s = 0; s < 6; s++) {
    const int d = 32 >> s;
    const uint64_t lo = LO[s];
    const uint64_t y = __shfl_xor(x, d, 64);
    x = (lane & d) ? ((x & ~lo) | ((y >> d) & lo)) : ((x & lo) | ((y << d) & ~lo));
  }
  return x;
}

// One 512-thread workgroup per (blob, 8 chunks): wave v transposes chunk q = 8 * (block % 8) + v (64 consecutive scalars);
// masks[(blob * 256 + k) * 64 + q] = bit k of scalars 64q .. 64q+63 (bit p of the mask = point 64q + p).  The [plane][chunk]
// layout is what the MSM reads contiguously (a lane's chunks of one plane are adjacent, the lanes of a plane group cover
// a whole 512-byte row); the eight waves exchange through LDS so that every row segment is written as one 64-byte piece.
// BE_BYTES: raw blob bytes, validated here (Blob::from_slice, src/blob.rs:26-37); a non-canonical element is treated as
// 0 and the blob's status is set.
template <bool BE_BYTES>
static __global__ __launch_bounds__(512) void k_comb_transpose(const uint8_t* __restrict__ scalars, uint64_t n, uint64_t* __restrict__ masks,
                                                               int32_t* __restrict__ status) {
  __shared__ uint64_t tile[256][8];
  issue_priority_latency();
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const uint64_t blob = blockIdx.x >> 3;
  const uint32_t q0 = (uint32_t)(blockIdx.x & 7u) * 8u;
  if (blob >= n) return;
  uint32_t sc[8];
  load_scalar<BE_BYTES>(sc, scalars + blob * (uint64_t)KZG_BYTES_PER_BLOB_ + (uint64_t)((q0 + (uint32_t)wv) * 64u + (uint32_t)lane) * 32u);
  bool bad = false;
  if (BE_BYTES) {
    fr_t v;
#pragma unroll
    for (int w = 0; w < 8; w++) v.v[w] = sc[w];
    if (!fr_is_canonical(v)) {
      bad = true;
#pragma unroll
      for (int w = 0; w < 8; w++) sc[w] = 0;
    }
  }
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint64_t row = ((uint64_t)sc[2 * w + 1] << 32) | sc[2 * w];
    tile[64 * w + lane][wv] = wave_transpose64(row, lane);
  }
  __syncthreads();
  {
    const int k = tid >> 1, part = tid & 1;  // 256 planes x two 32-byte halves of the 64-byte row segment
    uint4* dst = reinterpret_cast<uint4*>(masks + (blob * 256u + (uint64_t)k) * 64u + q0 + 4u * (uint32_t)part);
    const uint4* src = reinterpret_cast<const uint4*>(&tile[k][4 * part]);
    dst[0] = src[0];
    dst[1] = src[1];
  }
  if (BE_BYTES) {
    if (__any(bad) && lane == 0) atomicOr(&status[blob], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
}

// lane-walker of k_msm_comb30: position = (plane h counting down, block counter s, chunk q, block-in-chunk r)
struct CombWalker {
  uint32_t h, s, q, r;
};

// One wave per unit.  lpb = 64 lanes per blob: unit = (blob, split).  lpb = 32 (large batches, splits = 1): a wave carries
// TWO blobs on its two halves -- every lane owns twice the blocks for the same H - 1 doublings, which halves the Horner
// overhead.  Within a blob's lpb lanes: lane l = (group, owner); group grp = l / (lpb / G) walks planes
// [grp H, grp H + H) of its own table; owner = split * (lpb / G) + l % (lpb / G) owns blocks [owner * bpo, owner * bpo + bpo)
// of the 64 nb blocks.
// TIMED: instantiated only by the test-only library (tests/window_msm): every unit records {wall start, wall end, cycles,
// hw id} into wave_times (tools/gpu_wave_times.py); the product instantiates <false> and passes nullptr.
template <bool TIMED>
static __global__ __launch_bounds__(64, 2) void k_msm_comb30(const uint64_t* __restrict__ masks, uint64_t n, uint32_t splits, uint32_t lpb,
                                                             const uint4* __restrict__ table, CombGeom g, g1_xyzz* __restrict__ partials,
                                                             const uint4* __restrict__ comb_k, uint64_t* __restrict__ wave_times) {
  const int lane = threadIdx.x;
  const uint64_t unit = blockIdx.x;
  uint64_t wt_wall0 = 0, wt_cyc0 = 0;
  if (TIMED) {
    wt_wall0 = wall_clock64();
    wt_cyc0 = clock64();
  }
  const uint32_t l = (uint32_t)lane % lpb;
  const uint64_t blob = (lpb == 64u) ? unit / splits : unit * (64u / lpb) + (uint32_t)lane / lpb;
  const uint32_t split = (lpb == 64u) ? (uint32_t)(unit % splits) : 0u;
  if (blob >= n) {  // odd batch in half-wave mode: the idle half contributes the identity
    g1_xyzz none;
    xyzz_set_inf(none);
    partials[unit * 64 + lane] = none;
    return;
  }
  const uint32_t lpg = lpb / g.G;
  const uint32_t grp = l / lpg;
  const uint32_t owner = split * lpg + l % lpg;
  const uint32_t bpo = (64u * g.nb) / (splits * lpg);
  const uint32_t b0 = owner * bpo;
  const uint64_t* mrow = masks + blob * (64u * 256u);  // [plane][chunk]
  const uint32_t kbase = grp * g.H;
  const uint4* tgrp = table + (uint64_t)grp * g.epg * 6u;
  const uint32_t total = g.H * bpo;
  const uint32_t nb = g.nb;

  g1_xyzz30 acc;
  xyzz30_set_inf(acc);
  // The recoding's constant term K = [c0] * (sum of the setup points) is the STARTING VALUE of one lane per blob -- lane 0 of
  // the blob's first unit, plane group 0 -- so no later step has to add it.  The lane doubles its accumulator H - 1 times on
  // its way down the planes: comb_k holds [c0 / 2^(H-1)] * sum (affine, table format; null when it is the identity).
  if (comb_k != nullptr && l == 0u && split == 0u) {
    fp_t kx, ky;
    load_affine96(kx, ky, comb_k, 0);
    f30_unpack(acc.x, kx.v);
    f30_unpack(acc.y, ky.v);
    acc.zz = f30_one();
    acc.zzz = acc.zz;
    acc.inf = 0;
  }

  CombWalker w;  // next mask to load
  w.h = g.H - 1u;
  w.s = 0;
  w.q = b0 / nb;
  w.r = b0 % nb;
  auto advance = [&]() {
    w.s++;
    w.r++;
    if (w.r == nb) {
      w.r = 0;
      w.q++;
    }
    if (w.s == bpo) {
      w.s = 0;
      w.h--;
      w.q = b0 / nb;
      w.r = b0 % nb;
    }
  };
  // pipeline registers: mb* = mask(s) covering step t+1 with its walker position p1; (nx, ny) = table entry of step t.
  // A lane's chunks are consecutive, so when they come in aligned groups of four (bpo a multiple of 4 nb: every production
  // shape) the four masks are ONE 32-byte load per group instead of an 8-byte load per step: with [plane][chunk] rows a
  // 128-byte line holds 16 chunks of one plane = 4 owners x 4 groups, and the table gathers (20 GB per launch through a
  // 4-MB L2) evicted it between a lane's visits -- FETCH_SIZE counted 7.9 GB of mask re-fetches per 4,096 blobs.
  const bool wide = (bpo % (4u * nb)) == 0u;
  const uint32_t q0 = b0 / nb;
  uint64_t mb0 = 0, mb1 = 0, mb2 = 0, mb3 = 0;
  auto fetch_mask = [&]() {  // the mask of the step at w; in wide mode only when w enters a new group of four chunks
    if (wide) {
      if (w.r == 0u && ((w.q - q0) & 3u) == 0u) {
        const uint4* src = reinterpret_cast<const uint4*>(mrow + (kbase + w.h) * 64u + w.q);
        const uint4 lo = src[0], hi = src[1];
        mb0 = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
        mb1 = (uint64_t)lo.z | ((uint64_t)lo.w << 32);
        mb2 = (uint64_t)hi.x | ((uint64_t)hi.y << 32);
        mb3 = (uint64_t)hi.z | ((uint64_t)hi.w << 32);
      }
    } else {
      mb0 = mrow[(kbase + w.h) * 64u + w.q];
    }
  };
  fetch_mask();
  CombWalker p1 = w;
  advance();
  fp_t nx, ny;
  bool nneg = false, ndbl = false;
  uint32_t nidx = 0;
  auto gather = [&]() {  // entry of the step at p1 from its mask
    const uint32_t k = wide ? ((p1.q - q0) & 3u) : 0u;
    const uint64_t m1 = k == 0u ? mb0 : (k == 1u ? mb1 : (k == 2u ? mb2 : mb3));
    const uint32_t tb = comb_tbits(nb, p1.r);
    const uint32_t pat = (uint32_t)(m1 >> comb_point_off(nb, p1.r)) & ((1u << tb) - 1u);
    nneg = (pat >> (tb - 1u)) != 0u;  // top sign +1: the table holds the mirrored pattern, negated
    const uint32_t m = nneg ? (~pat & ((1u << (tb - 1u)) - 1u)) : pat;
    nidx = p1.q * g.ep64 + comb_entry_off(nb, p1.r) + m;
    ndbl = (p1.s == 0u) && (p1.h != g.H - 1u);
    load_affine96(nx, ny, tgrp, nidx);
  };
  gather();  // always before the next fetch_mask(): a new group's load overwrites masks whose patterns are already cut out
  if (total > 1u) {
    fetch_mask();
    p1 = w;
    advance();
  }

  // The two waves of a SIMD trade issue priority every 2^g.fair shader cycles (issue_fair.cuh).  Measured at 4,096 blobs = one
  // round of 2,048 waves: left to the hardware the older wave of every pair finished after 18.0 ms and the younger after
  // 29.7 ms; with a period of 2^16 cycles 24.7 / 29.2 ms (a low-priority wave's steps are 4x as long, it notices its turn
  // late); with 2^20 cycles 27.8 / 28.4 ms and the launch takes 29.2 instead of 30.7 ms (profiles/r02/wave_fairness_sweep.json).
#pragma unroll 1
  for (uint32_t t = 0; t < total; t++) {
    if (g.fair) issue_fair_tick_low(g.fair);
    fp30 cx, cy;
    f30_load_entry(cx, cy, nx.v, ny.v, xyzz30_entry_neg(acc, nneg));
    const bool cneg = nneg, cdbl = ndbl;
    const uint32_t cidx = nidx;
    if (t + 1u < total) {
      gather();
      if (t + 2u < total) {
        fetch_mask();
        p1 = w;
        advance();
      }
    }
    if (cdbl && !acc.inf) {  // Horner step between two planes: out of line, on a copy (63 times per lane at G = 4)
      g1_xyzz30 tmp = acc;
      xyzz30_dbl(tmp);
      acc = tmp;
    }
    bool done = false;
    if (!acc.inf) done = xyzz30_madd_fast(acc, cx, cy);
    if (!done) {
      g1_xyzz30 tmp = acc;
      fp_t rx, ry;
      load_affine96(rx, ry, tgrp, cidx);
      fp30 sx, sy;  // separate objects: the call takes their address
      f30_load_entry(sx, sy, rx.v, ry.v, xyzz30_entry_neg(tmp, cneg));
      xyzz30_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
  g1_xyzz out;
  xyzz30_to_xyzz(out, acc);  // back to canonical 2^384-Montgomery limbs for k_msm_reduce
  partials[unit * 64 + lane] = out;
  if (TIMED) {
    if (wave_times && lane == 0) {
      uint32_t hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      wave_times[unit * 4 + 0] = wt_wall0;
      wave_times[unit * 4 + 1] = wall_clock64();
      wave_times[unit * 4 + 2] = clock64() - wt_cyc0;
      wave_times[unit * 4 + 3] = ((uint64_t)xcc << 32) | hwid;
    }
  }
}

// ---- table build -------------------------------------------------------------------------------------------------
// thread i: B[g][i] = 2^(H g) * [1/2] L_i and D[g][i] = 2 B[g][i], affine, canonical 2^384-Montgomery (12 x 32 limbs)
static __global__ __launch_bounds__(64) void k_comb_bases(const uint4* __restrict__ bases_brp, CombGeom g, uint4* __restrict__ B, uint4* __restrict__ D) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4096) return;
  fp_t x, y;
  load_affine96(x, y, bases_brp, i);
  const uint32_t half[8] = KZG_FR_HALF_PLAIN;  // 1/2 mod r = (r + 1) / 2
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int bit = 254; bit >= 0; bit--) {
    xyzz_dbl(acc);
    if ((half[bit >> 5] >> (bit & 31)) & 1u) {
      g1_xyzz mine = acc;
      xyzz_madd(mine, x, y);
      acc = mine;
    }
  }
  for (uint32_t grp = 0; grp < g.G; grp++) {
    fp_t ax, ay;
    xyzz_to_affine(ax, ay, acc);
    store_affine96(B, (uint64_t)grp * 4096u + i, ax, ay);
    xyzz_from_affine(acc, ax, ay);
    g1_xyzz d2 = acc;
    xyzz_dbl(d2);
    fp_t dx, dy;
    xyzz_to_affine(dx, dy, d2);
    store_affine96(D, (uint64_t)grp * 4096u + i, dx, dy);
    if (grp + 1 < g.G)
      for (uint32_t k = 0; k < g.H; k++) xyzz_dbl(acc);
  }
}

// Subset sums of group `grp` for chunks [q_first, q_first + nq): one thread per segment of 2^sl consecutive entries of a
// block.  The segment's first entry is formed from the block's t points; the rest follows a Gray code over the low sl
// bits (one mixed addition of +-2B_p per entry).  tmp[(q - q_first) * ep64 + entry] in XYZZ; k_table_normalize writes the table.
static __global__ __launch_bounds__(64) void k_comb_chain(const uint4* __restrict__ B, const uint4* __restrict__ D, uint32_t grp, uint32_t q_first,
                                                          uint32_t nq, CombGeom g, uint32_t sl, g1_xyzz* __restrict__ tmp) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t segs_per_chunk = g.ep64 >> sl;
  if (tid >= (uint64_t)nq * segs_per_chunk) return;
  const uint32_t qi = (uint32_t)(tid / segs_per_chunk);
  const uint32_t e_start = (uint32_t)(tid % segs_per_chunk) << sl;
  uint32_t r = 0;
  while (r + 1 < g.nb && comb_entry_off(g.nb, r + 1) <= e_start) r++;
  const uint32_t tb = comb_tbits(g.nb, r);
  const uint32_t m_start = e_start - comb_entry_off(g.nb, r);
  const uint32_t i0 = (q_first + qi) * 64u + comb_point_off(g.nb, r);
  const uint4* Bg = B + (uint64_t)grp * 4096u * 6u;
  const uint4* Dg = D + (uint64_t)grp * 4096u * 6u;
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (uint32_t p = 0; p < tb; p++) {
    fp_t x, y;
    load_affine96(x, y, Bg, i0 + p);
    if (!((m_start >> p) & 1u)) fp_neg(y, y);  // sign -1 (always so for p = t-1)
    g1_xyzz mine = acc;
    xyzz_madd(mine, x, y);
    acc = mine;
  }
  g1_xyzz* o = tmp + (uint64_t)qi * g.ep64 + comb_entry_off(g.nb, r);
  o[m_start] = acc;
  uint32_t gray = 0;
#pragma unroll 1
  for (uint32_t j = 1; j < (1u << sl); j++) {
    const uint32_t p = (uint32_t)__builtin_ctz(j);
    gray ^= 1u << p;
    fp_t x, y;
    load_affine96(x, y, Dg, i0 + p);
    if (!((gray >> p) & 1u)) fp_neg(y, y);  // the sign went from +1 to -1: subtract 2B_p
    g1_xyzz mine = acc;
    xyzz_madd(mine, x, y);
    acc = mine;
    o[m_start | gray] = acc;
  }
}

#endif  // __HIPCC__
}  // namespace kzg
