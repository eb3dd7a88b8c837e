// Kernels of verify_blob_kzg_proof_batch (Setup::verify_blob_proof_batch,
// src/kzg/setup.rs:223-275 and Setup::verify_proof_batch, :115-161).
#pragma once
#include "blob_device.cuh"
#include "fr29.cuh"
#include "glv.cuh"
#include "msm_fixed.cuh"

namespace kzg {
#if defined(__HIPCC__)

// ---------------------------------------------------------------------------
// K1 + K5 for verification: Blob::from_slice validation + Polynomial::evaluate
// (src/kzg/poly.rs:10-33), G lanes per blob.
//
// The reference spends one field inversion per element (poly.rs:26).  Here the
// barycentric sum  S = sum_i e_i w_i / (z - w_i)  is accumulated as ONE fraction
// per lane,  (N, D) <- (N*d + a*D, D*d)  per PAIR of elements (below), nothing stored;
// the lane fractions are merged by a shuffle tree.  The blob elements stay plain:
// mont_mul(plain, X*R) = plain*X, mont_mul(plain, X*R^2) = plain*X*R.
// Because prod_i (z - w_i) = z^4096 - 1, the merged denominator cancels the
// barycentric factor and y = N / 4096: the whole evaluation needs no inversion.
// ---------------------------------------------------------------------------
// The arithmetic runs in the carry-free radix-2^29 representation of Fr (fr29.cuh, Montgomery radix R = 2^261).
// Bit-reversed order puts SIXTEEN related roots next to each other: elements 16h + 2k, 2k + 1 sit at +-w rho_k with
// rho = 1, i, c, ic, s, is, cs, ics (i, c, s: the primitive 4th, 8th and 16th roots of unity).  A pair at roots (x, -x) contributes
//   e0 x/(z - x) - e1 x/(z + x) = x [ (e0 - e1) z + (e0 + e1) x ] / (z^2 - x^2) = x u / d ,
// two pairs at (x, ix) make a quad, with d' = z^2 - (ix)^2 = z^2 + x^2 and d d' = z^4 - x^4:
//   x u/d + i x u'/d' = x [ u d' + u' (i d) ] / (z^4 - x^4) = x A~ / dd ,            i d = i z^2 - i x^2 ,
// two quads at (x, cx) make an oct, with dd' = z^4 - (cx)^4 = z^4 + x^4 and dd dd' = z^8 - x^8:
//   x A~/dd + c x A~'/dd' = x [ A~ dd' + A~' (c dd) ] / (z^8 - x^8) = x B~ / ddd ,     c dd = c z^4 - c x^4 ,
// and the octs at x = w and x = sw make the hex, with ddd' = z^8 - (sw)^8 = z^8 + w^8 and ddd ddd' = z^16 - w^16:
//   w B~/ddd + s w B~'/ddd' = w [ B~ ddd' + B~' (s ddd) ] / (z^16 - w^16) = w H~ / dddd ,   s ddd = s z^8 - s w^8 .
// Every denominator and every "i d" / "c dd" / "s ddd" is a limb-wise sum or difference of a per-blob power of z (LDS) and a
// table slot -- no product.  The quads of one oct share two slots ((cx)^2 = i x^2: d = z^2 - i x^2, d' = z^2 + i x^2,
// i d = i z^2 + x^2), the two octs have the same shape with (x^2, i x^2, x^4, c x^4) = (w^2, i w^2, w^4, c w^4) and
// (c w^2, i c w^2, i w^4, c i w^4): one loop body, run twice.  The root w is applied once per hex.  Per hex the kernel does 34
// products with 18 reductions (2.125 + 1.125 per element; per oct it was 2.25 + 1.25, per quad 2.5 + 1.5, per pair 3 + 2):
//   u   = ((e0 - e1) * zR + (e0 + e1) * xR) / R           plain, one reduction for two products        (eight pairs)
//   A~  = (u * d' + u' * (i d)) / R                       plain                                        (four quads)
//   B~  = (A~ * dd' + A~' * (c dd)) / R                   plain                                        (two octs)
//   H~  = (B~ * ddd' + B~' * (s ddd)) / R                 plain
//   H   = (H~ * wR^2) / R = (H~ w) R                       Montgomery: the hex's numerator
//   N'  = (N * dddd + H * D) / R,  D' = (D * dddd) / R     dddd = z^16 R - w^16 R
// eval_tab[h]: twenty slots (fr29.cuh), w = roots_brp[16 h]; slots 1, 2 and 4 of hex 0 (w = 1) are i R, c R and s R themselves.
template <int G>
__device__ __forceinline__ fr29 shfl_down_fr29(const fr29& a, int delta) {
  fr29 r;
#pragma unroll
  for (int q = 0; q < F29_N; q++) r.l[q] = __shfl_down(a.l[q], delta, G);
  return r;
}
__device__ __forceinline__ void eval_load_element(fr_t& e, const uint4& hi, const uint4& lo, bool& bad) {  // 32 big-endian bytes -> 8 little-endian limbs
  e.v[7] = __builtin_bswap32(hi.x); e.v[6] = __builtin_bswap32(hi.y); e.v[5] = __builtin_bswap32(hi.z); e.v[4] = __builtin_bswap32(hi.w);
  e.v[3] = __builtin_bswap32(lo.x); e.v[2] = __builtin_bswap32(lo.y); e.v[1] = __builtin_bswap32(lo.z); e.v[0] = __builtin_bswap32(lo.w);
  if (!fr_is_canonical(e)) {
    bad = true;
    bn_zero(e);
  }
}
// one 9-limb slot of a hex's table entry (three 16-byte loads)
__device__ __forceinline__ void eval_tab_slot(fr29& o, const uint32_t* __restrict__ entry, int slot) {
  const uint4* t = reinterpret_cast<const uint4*>(entry + slot * EVAL_TAB_SLOT);
  const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
  o.l[0] = t0.x; o.l[1] = t0.y; o.l[2] = t0.z; o.l[3] = t0.w;
  o.l[4] = t1.x; o.l[5] = t1.y; o.l[6] = t1.z; o.l[7] = t1.w;
  o.l[8] = t2.x;
}
// u = (e0 - e1) z + (e0 + e1) w for one pair (plain, N-form)
__device__ __forceinline__ void eval_pair_numerator(fr29& u, const fr_t& e0, const fr_t& e1, const fr29& z, const fr29& w) {
  fr29 x0, x1, sm, df;
  f29_from_bn(x0, e0);
  f29_from_bn(x1, e1);
  f29_add(sm, x0, x1);        // limbs < 2^30, value < 2r
  f29_sub_2r(df, x0, x1);     // limbs < 3*2^29, value < 3r
  f29_mul2(u, df, z, sm, w);  // 9*(3 + 2)*2^58 + 9*2^58 = 54*2^58 < 2^64
}
// G lanes work on one blob (64 / G blobs per wave).  G = 64 has the shortest latency (4 hexes per lane); G = 16 does
// 16 hexes per lane and a 4-level merge instead of 4 hexes and a 6-level one -- the merge is 11 % of a wave's work
// at G = 64, 2 % at G = 16 -- and is used when the batch fills the chip.
template <int G>
static __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_eval_frac(const uint8_t* __restrict__ blobs, const fr_t* __restrict__ z_plain,
                                                         const fr_t* __restrict__ roots_brp, const uint32_t* __restrict__ eval_tab,
                                                         fr_t* __restrict__ y_plain, int32_t* __restrict__ status, uint64_t n) {
  constexpr int PER_LANE = EVAL_TAB_HEXES / G;  // hexes per lane
  const int lane = threadIdx.x % G;   // position inside the blob's group
  const int group = threadIdx.x / G;
  uint64_t b = (uint64_t)blockIdx.x * (64 / G) + group;
  const bool live = b < n;
  if (!live) b = n - 1;  // idle groups shadow the last blob (they take part in the shuffles, never store)
  const uint8_t* blob = blobs + b * 131072ull;
  // z stays in registers (two products per pair); seven multiples of powers of z are read once or twice per hex and live in
  // LDS (one copy per blob group of the wave) -- with the pair-sized prefetch below that keeps the kernel at three waves per SIMD
  enum { Z2 = 0, IZ2 = 1, Z4 = 2, CZ4 = 3, Z8 = 4, SZ8 = 5, Z16 = 6, NZP = 7 };
  __shared__ alignas(16) uint32_t zpow[64 / G][NZP][12];
  fr29 z;
  {
    fr29 zp, z2, z4, z8, z16, kR, t;
    f29_from_bn(zp, z_plain[b]);
    f29_to_mont(z, zp);  // N-form
    f29_sqr(z2, z);
    f29_sqr(z4, z2);
    f29_sqr(z8, z4);
    f29_sqr(z16, z8);
    auto put = [&](int which, const fr29& v) {
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < F29_N; q++) zpow[group][which][q] = v.l[q];
      }
    };
    put(Z2, z2);
    put(Z4, z4);
    put(Z8, z8);
    put(Z16, z16);
    eval_tab_slot(kR, eval_tab, 1);  // hex 0 (w = 1): slot 1 = i R, slot 2 = c R, slot 4 = s R
    f29_mul(t, z2, kR);
    put(IZ2, t);
    eval_tab_slot(kR, eval_tab, 2);
    f29_mul(t, z4, kR);
    put(CZ4, t);
    eval_tab_slot(kR, eval_tab, 4);
    f29_mul(t, z8, kR);
    put(SZ8, t);
  }
  __syncthreads();
  auto load_zpow = [&](fr29& o, int which) {
    const uint4* t = reinterpret_cast<const uint4*>(&zpow[group][which][0]);
    const uint4 t0 = t[0], t1 = t[1], t2 = t[2];
    o.l[0] = t0.x; o.l[1] = t0.y; o.l[2] = t0.z; o.l[3] = t0.w;
    o.l[4] = t1.x; o.l[5] = t1.y; o.l[6] = t1.z; o.l[7] = t1.w;
    o.l[8] = t2.x;
  };
  // lane fraction N/D: both N-form between steps
  fr29 N, D = f29_const_one();
  KZG_UNROLL_FULL
  for (int q = 0; q < F29_N; q++) N.l[q] = 0;
  fr_t e_dom;
  bn_zero(e_dom);
  bool bad = false;
  int dom = -1, dom_h = -1;
  // The blob elements come from HBM (each byte is read exactly once): the next PAIR (64 contiguous bytes) is in flight while
  // this one is processed; the table slots are L2-resident and loaded where they are used (short live ranges).
  // Loads return in issue order, so the wait for a table slot also waits for every load issued before it: the table slots
  // of a step are therefore issued FIRST and the blob prefetch LAST (pinned with sched_barrier) -- the prefetch then stays
  // in flight for a whole pair step instead of being drained by the next table access a few dozen instructions later.  For
  // the same reason no load sits behind a branch (the compiler's wait counts turn pessimistic at a join).
  uint4 nb0, nb1, nb2, nb3;
  {
    const uint4* src = reinterpret_cast<const uint4*>(blob + (uint64_t)lane * 512u);  // first pair of hex `lane`
    nb0 = src[0]; nb1 = src[1]; nb2 = src[2]; nb3 = src[3];
  }
  // one pair step: u = (e0 - e1) z + (e0 + e1) x with x = table slot `slot`; the pair at byte offset `next` is fetched meanwhile
  auto pair_step = [&](fr29& u, const uint32_t* tab, int slot, uint64_t next) {
    fr29 x;
    eval_tab_slot(x, tab, slot);
    __builtin_amdgcn_sched_barrier(0);
    fr_t e0, e1;
    eval_load_element(e0, nb0, nb1, bad);
    eval_load_element(e1, nb2, nb3, bad);
    const uint4* src = reinterpret_cast<const uint4*>(blob + next);
    nb0 = src[0]; nb1 = src[1]; nb2 = src[2]; nb3 = src[3];
    __builtin_amdgcn_sched_barrier(0);
    eval_pair_numerator(u, e0, e1, z, x);
  };
#pragma unroll 1
  for (int k = 0; k < PER_LANE; k++) {
    const int hd = k * G + lane;  // hex index: elements 16 hd .. 16 hd + 15
    const uint32_t* tab = eval_tab + (uint64_t)hd * EVAL_TAB_DWORDS;
    // is z one of this hex's roots (z^16 == w^16)?  Tested up here, where the pending prefetch is needed at once anyway.
    bool in_domain = false;
    {
      fr29 w16, z16, d16;
      eval_tab_slot(w16, tab, 18);
      load_zpow(z16, Z16);
      f29_sub_2r(d16, z16, w16);
      if (f29_maybe_zero(d16)) in_domain = f29_is_zero_exact(d16);
    }
    fr29 Bt1, Bt;
#pragma unroll 1
    for (int oc = 0; oc < 2; oc++) {  // the octs at w and at s w: the same shape on different slots
      const uint64_t base = (uint64_t)hd * 512u + (uint32_t)oc * 256u;
      const int ps = 4 * oc;        // pair roots x, ix, cx, icx
      const int sq = 8 + 2 * oc;    // x^2, i x^2
      const int q4 = 12 + 2 * oc;   // x^4, c x^4
      fr29 At1;
      {  // quad 1: pairs at x and ix
        fr29 u0, u1, dp, id;
        pair_step(u0, tab, ps, base + 64u);
        {
          fr29 xsq, ixsq, z2, iz2;
          eval_tab_slot(xsq, tab, sq);
          eval_tab_slot(ixsq, tab, sq + 1);
          load_zpow(z2, Z2);
          load_zpow(iz2, IZ2);
          f29_add(dp, z2, xsq);       // d' = z^2 + x^2: limbs < 2^30, value < 3r
          f29_sub_2r(id, iz2, ixsq);  // i d = i z^2 - i x^2: limbs < 3*2^29, value < 4r
        }
        pair_step(u1, tab, ps + 1, base + 128u);
        f29_mul2(At1, u0, dp, u1, id);  // 9*(2 + 3)*2^58 + 9*2^58 < 2^64;  value 2*3 + 2*4 = 14 < 2^6;  plain
      }
      fr29 At2;
      {  // quad 2: pairs at cx and icx; (cx)^2 = i x^2
        fr29 u2, u3, dp, id;
        pair_step(u2, tab, ps + 2, base + 192u);
        {
          fr29 xsq, ixsq, z2, iz2;
          eval_tab_slot(xsq, tab, sq);
          eval_tab_slot(ixsq, tab, sq + 1);
          load_zpow(z2, Z2);
          load_zpow(iz2, IZ2);
          f29_add(dp, z2, ixsq);  // d' = z^2 + i x^2: limbs < 2^30, value < 3r
          f29_add(id, iz2, xsq);  // i d = i z^2 + x^2: limbs < 2^30, value < 3r
        }
        // unconditional prefetch (a conditional load makes the wait counts pessimistic): the very last step re-reads the lane's
        // first hex, which is in bounds and in L2
        const uint64_t next = (oc == 0) ? base + 256u : (uint64_t)((k + 1 < PER_LANE) ? hd + G : lane) * 512u;
        pair_step(u3, tab, ps + 3, next);
        f29_mul2(At2, u2, dp, u3, id);
      }
      {
        fr29 ddp, cdd;
        {
          fr29 x4, cx4, z4, cz4;
          eval_tab_slot(x4, tab, q4);
          eval_tab_slot(cx4, tab, q4 + 1);
          load_zpow(z4, Z4);
          load_zpow(cz4, CZ4);
          f29_add(ddp, z4, x4);        // dd' = z^4 + x^4: limbs < 2^30, value < 3r
          f29_sub_2r(cdd, cz4, cx4);   // c dd = c z^4 - c x^4: limbs < 3*2^29, value < 4r
        }
        f29_mul2(Bt, At1, ddp, At2, cdd);  // same bounds as A~
      }
      if (oc == 0) Bt1 = Bt;
    }
    if (in_domain) {  // resolved after the loop (no loads in here)
      dom_h = hd;
      continue;
    }
    fr29 H;
    {
      fr29 e8p, sd8;
      {
        fr29 w8, sw8, z8, sz8;
        eval_tab_slot(w8, tab, 16);
        eval_tab_slot(sw8, tab, 17);
        load_zpow(z8, Z8);
        load_zpow(sz8, SZ8);
        f29_add(e8p, z8, w8);        // ddd' = z^8 + w^8
        f29_sub_2r(sd8, sz8, sw8);   // s ddd = s z^8 - s w^8
      }
      f29_mul2(H, Bt1, e8p, Bt, sd8);
    }
    {
      fr29 wr2;
      eval_tab_slot(wr2, tab, 19);
      f29_mul(H, H, wr2);  // (H~ w) R
    }
    fr29 d16;
    {
      fr29 w16, z16;
      eval_tab_slot(w16, tab, 18);
      load_zpow(z16, Z16);
      f29_sub_2r(d16, z16, w16);  // z^16 - w^16: limbs < 3*2^29, value < 4r
    }
    f29_mul2(N, N, d16, H, D);  // 9*(3 + 1)*2^58 + 9*2^58;  value 2*4 + 2*2 = 12
    f29_mul(D, D, d16);
  }
  if (dom_h >= 0) {  // rare (poly.rs:14-18): which of the hex's sixteen roots is z?  The evaluation is that element (re-read).
    const uint32_t* tab = eval_tab + (uint64_t)dom_h * EVAL_TAB_DWORDS;
    int which = 15;
#pragma unroll 1
    for (int pr = 7; pr >= 0; pr--) {
      fr29 x, t;
      eval_tab_slot(x, tab, pr);
      f29_sub_2r(t, z, x);
      if (f29_is_zero_exact(t)) which = 2 * pr;
      f29_add(t, z, x);
      if (f29_is_zero_exact(t)) which = 2 * pr + 1;
    }
    dom = 16 * dom_h + which;
    const uint4* src = reinterpret_cast<const uint4*>(blob + (uint64_t)dom * 32u);
    bool dummy = false;
    eval_load_element(e_dom, src[0], src[1], dummy);
  }
  // merge lane fractions: (N1/D1) + (N2/D2) = (N1 D2 + N2 D1) / (D1 D2)
#pragma unroll 1
  for (int delta = G / 2; delta >= 1; delta >>= 1) {
    const fr29 N2 = shfl_down_fr29<G>(N, delta), D2 = shfl_down_fr29<G>(D, delta);
    f29_mul2(N, N, D2, N2, D);
    f29_mul(D, D, D2);
  }
  int dom_any = dom;
#pragma unroll
  for (int delta = G / 2; delta >= 1; delta >>= 1) {
    const int o = __shfl_xor(dom_any, delta, G);
    dom_any = o > dom_any ? o : dom_any;
  }
  // The merged denominator is prod_q (z^4 - w_q^4) = z^4096 - 1 exactly, so
  //   y = (N / D) * (z^4096 - 1) / 4096 = N / 4096          -- no inversion at all.
  fr_t y;
  if (dom_any >= 0) {
    const int owner = (dom_any >> 4) % G;  // hex index hd = k*G + lane
#pragma unroll
    for (int q = 0; q < 8; q++) y.v[q] = __shfl(e_dom.v[q], owner, G);  // already plain
  } else {
    fr29 f, t;
    KZG_UNROLL_FULL
    for (int q = 0; q < F29_N; q++) f.l[q] = f29_inv4096_limb(q);
    f29_mul(t, N, f);  // (N R)(1/4096) / R: plain
    f29_to_canonical_bn(y, t);
  }
  const unsigned long long bad_lanes = __ballot(bad);
  const unsigned long long group_mask = (G == 64) ? ~0ull : (((1ull << (G % 64)) - 1ull) << (group * G));
  if (live && lane == 0) {
    y_plain[b] = y;
    if (bad_lanes & group_mask) atomicOr(&status[b], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
}

// ---------------------------------------------------------------------------
// Batch challenge transcript.  kateth derives r from the batch SIZE only
// (src/kzg/setup.rs:127-136); the Deneb spec binds every
// input.  The engine binds every input through a SHA-256 tree so the hashing is
// parallel AND short:  leaf_i = H(C_i || z_i || y_i || pi_i)  (160 B),
// mid_j = H(leaf_{16j} .. leaf_{16j+15}), node_g = H(mid_{16g} .. mid_{16g+15})
// (ragged at the end), and the host hashes
// "RCKZGBATCH___V1_" || u128(4096) || u128(n) || node_0 || ... into the seed.
// (Fan-out 16 twice instead of 256 once: a thread hashes 9 blocks, not 129 -- the
// 256-ary level was a 0.6 ms serial chain at the end of phase 1.)
// ---------------------------------------------------------------------------
// Both transcript kernels stay within 64 VGPRs (second launch bound: eight waves per SIMD): a SIMD that holds two point-decoder
// waves (2 x 224 registers) has 64 left, so these run in the decoder's shadow instead of after it.  The leaf's 160 bytes are
// fetched block by block for that.
static __global__ __launch_bounds__(256, 8) void k_transcript_leaves(const uint8_t* __restrict__ commitments48, const uint8_t* __restrict__ proofs48,
                                                             const fr_t* __restrict__ z_plain, const fr_t* __restrict__ y_plain, uint64_t n,
                                                             uint32_t* __restrict__ leaves /* n x 8 words, big-endian word values */) {
  issue_priority_latency();  // short kernels in the shadow of the point decoder (whose waves trade priorities 3 / 1)
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* c = reinterpret_cast<const uint32_t*>(commitments48 + i * 48);
  const uint32_t* p = reinterpret_cast<const uint32_t*>(proofs48 + i * 48);
  const uint32_t* zw = reinterpret_cast<const uint32_t*>(z_plain + i);
  const uint32_t* yw = reinterpret_cast<const uint32_t*>(y_plain + i);
  sha256_state s;
  sha256_init(s);
  uint32_t w[16];
  // words 0..39 of the message: C (12, byte-swapped), z (8, limbs 7..0), y (8), pi (12, byte-swapped)
#pragma unroll
  for (int q = 0; q < 12; q++) w[q] = __builtin_bswap32(c[q]);
#pragma unroll
  for (int q = 0; q < 4; q++) w[12 + q] = zw[7 - q];
  sha256_block(s, w);
#pragma unroll
  for (int q = 0; q < 4; q++) w[q] = zw[3 - q];
#pragma unroll
  for (int q = 0; q < 8; q++) w[4 + q] = yw[7 - q];
#pragma unroll
  for (int q = 0; q < 4; q++) w[12 + q] = __builtin_bswap32(p[q]);
  sha256_block(s, w);
#pragma unroll
  for (int q = 0; q < 8; q++) w[q] = __builtin_bswap32(p[4 + q]);
  w[8] = 0x80000000u;
#pragma unroll
  for (int q = 9; q < 15; q++) w[q] = 0;
  w[15] = 160 * 8;
  sha256_block(s, w);
#pragma unroll
  for (int q = 0; q < 8; q++) leaves[i * 8 + q] = s.h[q];
}

// out[g] = H(in[g * fan] .. in[min(n_in, (g + 1) * fan) - 1]) over 32-byte digests (8 big-endian word values each)
static __global__ __launch_bounds__(64, 8) void k_transcript_nodes(const uint32_t* __restrict__ in, uint64_t n_in, uint32_t fan, uint32_t* __restrict__ nodes) {
  issue_priority_latency();
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t groups = (n_in + fan - 1) / fan;
  if (g >= groups) return;
  const uint32_t* leaves = in;
  const uint64_t first = g * fan;
  const uint64_t cnt = (n_in - first < fan) ? (n_in - first) : fan;
  sha256_state s;
  sha256_init(s);
  uint32_t w[16];
  // cnt digests of 32 B: two per block
  uint64_t k = 0;
  for (; k + 2 <= cnt; k += 2) {
#pragma unroll
    for (int q = 0; q < 16; q++) w[q] = leaves[(first + k) * 8 + q];
    sha256_block(s, w);
  }
  const uint64_t bits = cnt * 256;
  if (k < cnt) {  // odd count: one digest + padding in the same block
#pragma unroll
    for (int q = 0; q < 8; q++) w[q] = leaves[(first + k) * 8 + q];
    w[8] = 0x80000000u;
#pragma unroll
    for (int q = 9; q < 15; q++) w[q] = 0;
    w[15] = (uint32_t)bits;
    sha256_block(s, w);
  } else {
    w[0] = 0x80000000u;
#pragma unroll
    for (int q = 1; q < 15; q++) w[q] = 0;
    w[15] = (uint32_t)bits;
    sha256_block(s, w);
  }
#pragma unroll
  for (int q = 0; q < 8; q++) nodes[g * 8 + q] = s.h[q];
}

// ---------------------------------------------------------------------------
// Random-linear-combination scalars (src/kzg/setup.rs:138-150).  With r the
// transcript challenge, item i (GLOBAL index g = first_index + i) gets r^g, the
// spec's powers 0..n-1 (kateth's own r.pow(0) == r quirk, SURVEY Q2, is not
// reproduced).  rpow2[k] = r^(2^k) (Montgomery).
//   out scalars (plain):  sa[i] = r_i            (for proof_i in A = sum r_i pi_i)
//                         sb[i] = r_i * z_i      (for proof_i in B)
//                         sc[i] = r_i            (for commitment_i in B)   -- same array as sa
//   and the per-thread partial of  s = sum r_i y_i  reduced per block into ysum_blocks.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_batch_scalars(const fr_t* __restrict__ rpow2, const fr_t* __restrict__ z_plain,
                                                       const fr_t* __restrict__ y_plain, uint64_t n, uint64_t first_index,
                                                       fr_t* __restrict__ sa, fr_t* __restrict__ sb, fr_t* __restrict__ ysum_blocks) {
  __shared__ fr_t red[256];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  fr_t term;
  bn_zero(term);
  if (i < n) {
    uint64_t e = first_index + i;
    fr_t r = fr_one();
    for (int k = 0; k < 64 && (e >> k); k++)
      if ((e >> k) & 1) fr_mul(r, r, rpow2[k]);
    fr_t zm, ym, t;
    to_mont<FrParams>(zm, z_plain[i]);
    to_mont<FrParams>(ym, y_plain[i]);
    fr_mul(t, r, zm);
    fr_t rp, tp;
    from_mont<FrParams>(rp, r);
    from_mont<FrParams>(tp, t);
    sa[i] = rp;
    sb[i] = tp;
    fr_mul(term, r, ym);
  }
  red[threadIdx.x] = term;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      fr_t a = red[threadIdx.x], c = red[threadIdx.x + w], s;
      fr_add(s, a, c);
      red[threadIdx.x] = s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) ysum_blocks[blockIdx.x] = red[0];  // Montgomery
}

// s = sum of the block partials; writes -s (plain) as the scalar of the generator term
static __global__ __launch_bounds__(256) void k_batch_ysum_finish(const fr_t* __restrict__ ysum_blocks, uint32_t nblocks, fr_t* __restrict__ out_neg_plain) {
  __shared__ fr_t red[256];
  fr_t acc;
  bn_zero(acc);
  for (uint32_t k = threadIdx.x; k < nblocks; k += 256) fr_add(acc, acc, ysum_blocks[k]);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      fr_t a = red[threadIdx.x], c = red[threadIdx.x + w], s;
      fr_add(s, a, c);
      red[threadIdx.x] = s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    fr_t neg, p;
    fr_neg(neg, red[0]);
    from_mont<FrParams>(p, neg);
    *out_neg_plain = p;
  }
}

// ---------------------------------------------------------------------------
// K8: variable-base MSM (replaces the three P1::lincomb calls,
// src/kzg/setup.rs:152-155 -> src/bls.rs:406-413), Pippenger with signed
// base-2^c digits and counting-sorted bucket lists:
//   k_var_count    : histogram of (window, |digit|) over all terms
//   (host-side exclusive scan is replaced by k_var_scan, one block)
//   k_var_scatter  : term ids into per-bucket lists
//   k_var_buckets  : one thread per bucket, complete mixed adds in XYZZ
//   k_var_fold     : wave-tree sum of the K partials of each bucket
//   k_var_windows  : per window, sum_d d*B_d as the sum of all suffix sums (scan + tree in one wave)
// The W window sums go back to the host, which does the Horner combine
// (W*c doublings -- a serial chain that one CPU core finishes in < 1 ms).
// Terms: point index t in [0, nterms), affine points + infinity flags.
// ---------------------------------------------------------------------------
struct VarGeom {
  uint32_t c, W, half;  // buckets per window = half = 2^(c-1)
  // Large batches (the "flat" path, top_n != 0): c = 13, ONE thread per bucket for the full windows, the short top window
  // (8 scalar bits: at most top_n magnitudes, each with 2^(c-8) times the load) split over ktop threads per bucket, and the
  // per-window sum_d d*B_d replaced by bit sums (k_var_bitsums) that the host folds into its Horner loop.
  uint32_t top_n, ktop;
  uint32_t seg;  // flat path: entries per lane of the balanced bucket kernel (k_var_buckets_seg); 0 = one thread per bucket
};

// signed digits of one scalar, all windows, through a callback
template <class F>
__device__ __forceinline__ void var_digits(const fr_t& s_plain, const VarGeom& g, F&& f) {
  uint32_t sc[8];
#pragma unroll
  for (int q = 0; q < 8; q++) sc[q] = s_plain.v[q];
  const uint32_t mask = (1u << g.c) - 1u;
  uint32_t carry = 0;
  for (uint32_t j = 0; j < g.W; j++) {
    uint32_t u = (sc[0] & mask) + carry;
#pragma unroll
    for (int q = 0; q < 7; q++) sc[q] = (sc[q] >> g.c) | (sc[q + 1] << (32u - g.c));
    sc[7] >>= g.c;
    const bool neg = u > g.half;
    const uint32_t d = neg ? ((1u << g.c) - u) : u;
    carry = neg ? 1u : 0u;
    if (d) f(j, d, neg);
  }
}

static __global__ __launch_bounds__(256) void k_var_count(const fr_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint64_t nterms,
                                                   VarGeom g, uint32_t* __restrict__ counts) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nterms || (inf && inf[t])) return;  // inf == null: points at infinity are skipped in the bucket chains instead (zero entries)
  var_digits(scalars[t], g, [&](uint32_t j, uint32_t d, bool) { atomicAdd(&counts[(uint64_t)j * g.half + (d - 1)], 1u); });
}

// exclusive scan of `len` counters into offsets (single 1024-thread block, chunked)
static __global__ __launch_bounds__(1024) void k_var_scan(const uint32_t* __restrict__ counts, uint32_t len, uint32_t* __restrict__ offsets,
                                                   uint32_t* __restrict__ cursors) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < len; base += 1024) {
    const uint32_t idx = base + t;
    const uint32_t v = idx < len ? counts[idx] : 0u;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t add = (t >= off) ? sh[t - off] : 0u;
      __syncthreads();
      sh[t] += add;
      __syncthreads();
    }
    const uint32_t excl = sh[t] - v + carry;
    if (idx < len) {
      offsets[idx] = excl;
      cursors[idx] = excl;
    }
    __syncthreads();
    if (t == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (t == 0) offsets[len] = carry;
}

// the same scan for long counter arrays (81,920 counters at c = 13): each of the 1024 threads owns a contiguous segment of
// PER counters (PER a multiple of 4: uint4 loads and stores, all issued before the first use), one block-wide scan of the
// segment totals in between.  len <= 1024 * PER; counters past len read as zero.
template <int PER>
static __global__ __launch_bounds__(1024) void k_var_scan_wide(const uint32_t* __restrict__ counts, uint32_t len, uint32_t* __restrict__ offsets,
                                                               uint32_t* __restrict__ cursors) {
  static_assert(PER % 4 == 0, "vector width");
  __shared__ uint32_t sh[1024];
  const int t = threadIdx.x;
  const uint32_t lo = (uint32_t)t * PER;
  uint32_t v[PER];
#pragma unroll
  for (int q = 0; q < PER / 4; q++) {
    uint4 x = make_uint4(0, 0, 0, 0);
    if (lo + 4u * q + 3u < len) {
      x = *reinterpret_cast<const uint4*>(counts + lo + 4u * q);
    } else {
      if (lo + 4u * q < len) x.x = counts[lo + 4u * q];
      if (lo + 4u * q + 1u < len) x.y = counts[lo + 4u * q + 1u];
      if (lo + 4u * q + 2u < len) x.z = counts[lo + 4u * q + 2u];
    }
    v[4 * q] = x.x;
    v[4 * q + 1] = x.y;
    v[4 * q + 2] = x.z;
    v[4 * q + 3] = x.w;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) sum += v[q];
  sh[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t add = (t >= off) ? sh[t - off] : 0u;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  uint32_t run = sh[t] - sum;  // exclusive prefix of this segment
#pragma unroll
  for (int q = 0; q < PER / 4; q++) {
    uint4 o;
    o.x = run;
    o.y = o.x + v[4 * q];
    o.z = o.y + v[4 * q + 1];
    o.w = o.z + v[4 * q + 2];
    run = o.w + v[4 * q + 3];
    if (lo + 4u * q + 3u < len) {
      *reinterpret_cast<uint4*>(offsets + lo + 4u * q) = o;
      *reinterpret_cast<uint4*>(cursors + lo + 4u * q) = o;
    } else {
      const uint32_t ov[4] = {o.x, o.y, o.z, o.w};
      for (uint32_t e = 0; e < 4u; e++)
        if (lo + 4u * q + e < len) {
          offsets[lo + 4u * q + e] = ov[e];
          cursors[lo + 4u * q + e] = ov[e];
        }
    }
  }
  if (t == 1023) offsets[len] = sh[1023];
}

// The same scan for the sorting that runs in the point decoder's shadow: ONE workgroup of four waves within 64 VGPRs (a SIMD
// holding two decoder waves has 64 registers left), every thread walks its contiguous run of counters twice -- totals, then
// offsets -- instead of holding it in registers.  len <= 256 * per, per a multiple of 4.
static __global__ __launch_bounds__(256, 8) void k_var_scan_lean(const uint32_t* __restrict__ counts, uint32_t len, uint32_t per, uint32_t* __restrict__ offsets,
                                                         uint32_t* __restrict__ cursors) {
  __shared__ uint32_t sh[256];
  const uint32_t t = threadIdx.x;
  const uint32_t lo = t * per, hi = (lo + per < len) ? lo + per : len;
  uint32_t sum = 0;
  for (uint32_t q = lo; q + 3u < hi; q += 4u) {
    const uint4 x = *reinterpret_cast<const uint4*>(counts + q);
    sum += x.x + x.y + x.z + x.w;
  }
  for (uint32_t q = lo + ((hi > lo ? hi - lo : 0u) & ~3u); q < hi; q++) sum += counts[q];
  sh[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 256u; off <<= 1) {
    const uint32_t add = (t >= off) ? sh[t - off] : 0u;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  uint32_t run = sh[t] - sum;  // exclusive prefix of this run
  uint32_t q = lo;
  for (; q + 3u < hi; q += 4u) {
    const uint4 x = *reinterpret_cast<const uint4*>(counts + q);
    uint4 o;
    o.x = run;
    o.y = o.x + x.x;
    o.z = o.y + x.y;
    o.w = o.z + x.z;
    run = o.w + x.w;
    *reinterpret_cast<uint4*>(offsets + q) = o;
    *reinterpret_cast<uint4*>(cursors + q) = o;
  }
  for (; q < hi; q++) {
    const uint32_t v = counts[q];
    offsets[q] = run;
    cursors[q] = run;
    run += v;
  }
  if (t == 255u) {
    offsets[len] = sh[255];
    cursors[len] = sh[255];
  }
}

// An entry names the POINT of its term: term t uses point t, or -- GLV, the second half of a lincomb's terms (t >= split) -- point
// second_base + (t - split), the [z^2]-image of point t - split (k_glv_points).  Without GLV split = nterms.
static __global__ __launch_bounds__(256) void k_var_scatter(const fr_t* __restrict__ scalars, const uint8_t* __restrict__ inf, uint64_t nterms,
                                                     VarGeom g, uint32_t* __restrict__ cursors, uint32_t* __restrict__ entries, uint64_t split,
                                                     uint64_t second_base) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nterms || (inf && inf[t])) return;
  const uint32_t point = (uint32_t)(t < split ? t : second_base + (t - split));
  var_digits(scalars[t], g, [&](uint32_t j, uint32_t d, bool neg) {
    const uint32_t pos = atomicAdd(&cursors[(uint64_t)j * g.half + (d - 1)], 1u);
    entries[pos] = (point << 1) | (neg ? 1u : 0u);
  });
}

// ---- GLV (glv.cuh): scalars split at z^2, points mapped by [z^2](x, y) = (beta x, -y) ------------------------------------------
// scal = the batch scalars as k_batch_scalars / k_batch_ysum_finish leave them: [r_i z_i (n) | r_i (n) | -sum r_i y_i (1)], plain.
// out_b (2 (2n + 1)): lincomb B's terms  [k1 of all 2n + 1 | k2 of all 2n + 1];  out_a (2n): lincomb A's  [k1 of r_i | k2 of r_i].
static __global__ __launch_bounds__(256) void k_glv_split(const fr_t* __restrict__ scal, uint64_t n, fr_t* __restrict__ out_b, fr_t* __restrict__ out_a) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nt = 2 * n + 1;
  if (t >= nt) return;
  fr_t k1, k2;
  glv_split(k1, k2, scal[t]);
  out_b[t] = k1;
  out_b[nt + t] = k2;
  if (t >= n && t < 2 * n) {
    out_a[t - n] = k1;
    out_a[t] = k2;  // n + (t - n)
  }
}
// points [0, npts) of `aff` (x * 2^392, y * 2^392, canonical; all zero = infinity or a rejected input) -> their [z^2]-images
// (beta x, -y) at [phi_off, phi_off + npts)
static __global__ __launch_bounds__(64) void k_glv_points(uint4* __restrict__ aff, uint64_t npts, uint64_t phi_off) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npts) return;
  fp_t x, y;
  load_affine96(x, y, aff, t);
  if (!(bn_is_zero(x) && bn_is_zero(y))) {
    fp28 a, b;
    f28_from_bn(a, x);
    KZG_UNROLL_FULL
    for (int i = 0; i < F28_N; i++) b.l[i] = f28_beta_limb(i);
    f28_mul(a, a, b);  // beta x * 2^392, N-form
    f28_to_bn(x, a);
    canonicalize<FpParams>(x);
    fp_neg(y, y);      // p - y: y is canonical and non-zero on this curve's group (no point of order two)
  }
  store_affine96(aff, phi_off + t, x, y);
}

// thread = (bucket, k of K): entries lo+k, lo+k+K, ... of the bucket's sorted list.  Points are affine in the
// 2^392-Montgomery domain (k_g1_decompress), the accumulator is XYZZ in radix-2^28 limbs with the MSM hot loop's inline
// adder (xyzz28_madd_fast; the out-of-line complete adder takes the identity and the P == +-Q cases, which DO occur
// here: a batch may repeat a point); the next entry is in flight while the current one is added.
// entries lo + k0, lo + k0 + K, ... of one bucket's sorted list, summed
__device__ __forceinline__ void var_bucket_chain(g1_xyzz28& acc, const uint4* __restrict__ points, const uint32_t* __restrict__ entries, uint32_t lo,
                                                 uint32_t hi, uint32_t k0, uint32_t K) {
  xyzz28_set_inf(acc);
  fp_t nx, ny;
  bn_zero(nx);
  bn_zero(ny);
  uint32_t ne = 0;
  if (lo + k0 < hi) {
    ne = entries[lo + k0];
    load_affine96(nx, ny, points, ne >> 1);
  }
#pragma unroll 1
  for (uint32_t k = lo + k0; k < hi; k += K) {
    const uint32_t e = ne;
    // an all-zero entry is the point at infinity (x = y = 0 is not on the curve): it contributes nothing.  The sorting kernels
    // do not look at the decoder's infinity flags any more -- they run BESIDE the decoder -- so such terms reach the chains.
    const bool skip = bn_is_zero(nx) && bn_is_zero(ny);
    fp28 cx, cy;
    f28_load_entry(cx, cy, nx, ny, (e & 1u) != 0);
    if (k + K < hi) {
      ne = entries[k + K];
      load_affine96(nx, ny, points, ne >> 1);
    }
    if (skip) continue;
    bool done = false;
    if (!acc.inf) done = xyzz28_madd_fast(acc, cx, cy);
    if (!done) {
      g1_xyzz28 tmp = acc;  // copy: the call takes addresses
      fp_t rx, ry;
      load_affine96(rx, ry, points, e >> 1);
      fp28 sx, sy;
      f28_load_entry(sx, sy, rx, ry, (e & 1u) != 0);
      xyzz28_madd_complete(tmp, sx, sy);
      acc = tmp;
    }
  }
}
static __global__ __launch_bounds__(64, 2) void k_var_buckets(const uint4* __restrict__ points, const uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ entries, uint32_t nbuckets, uint32_t K,
                                                       g1_xyzz28* __restrict__ partial_sums) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (uint64_t)nbuckets * K) return;
  const uint32_t bkt = (uint32_t)(id / K), k0 = (uint32_t)(id % K);
  g1_xyzz28 acc;
  var_bucket_chain(acc, points, entries, offsets[bkt], offsets[bkt + 1], k0, K);
  partial_sums[id] = acc;  // the fold and window kernels stay in the radix-2^28 field
}
// flat path: threads [0, regular) own one bucket of a full window each (sum straight into bucket_sums); the rest split the
// top window's buckets [regular, regular + top_n) over ktop threads each (partials for k_var_fold)
static __global__ __launch_bounds__(64, 2) void k_var_buckets_flat(const uint4* __restrict__ points, const uint32_t* __restrict__ offsets,
                                                                   const uint32_t* __restrict__ entries, uint32_t regular, uint32_t top_n, uint32_t ktop,
                                                                   g1_xyzz28* __restrict__ bucket_sums, g1_xyzz28* __restrict__ top_partials) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz28 acc;
  if (id < regular) {
    var_bucket_chain(acc, points, entries, offsets[id], offsets[id + 1], 0u, 1u);
    bucket_sums[id] = acc;
  } else {
    const uint64_t t = id - regular;
    if (t >= (uint64_t)top_n * ktop) return;
    const uint32_t bkt = regular + (uint32_t)(t / ktop);
    var_bucket_chain(acc, points, entries, offsets[bkt], offsets[bkt + 1], (uint32_t)(t % ktop), ktop);
    top_partials[t] = acc;
  }
}

// ---- flat path, BALANCED ("segmented"): every lane sums the same number of entries --------------------------------------------
// With one thread per bucket a wave runs until its LONGEST bucket is done: bucket sizes are Poisson (mean 16 / 32 entries for the
// two lincombs of 65,536 triples), the longest of 64 is 26 / 46, so 30-40 % of the lanes of every instruction are idle -- and the
// 5,120 waves of the two kernels are 2.5 rounds of the chip's 2,048 wave slots at this register budget.  Here the full windows'
// sorted entry list [0, offsets[regular]) is cut into equal shares of E consecutive entries, one per lane (E chosen by the host so
// that both lincombs' lanes are ONE round); a lane walks its share and, where a bucket's list ends inside it, stores that bucket's
// sum and starts the next.  A bucket that lies inside one share is written to bucket_sums directly; one that crosses share
// boundaries leaves partial sums -- seg_part[2 l + 1]: the bucket that BEGINS in share l and runs past its end; seg_part[2 l]: the
// one that was already running when share l began (its middle or its end) -- which k_var_seg_fixup adds up, one thread per bucket
// (it also writes the identity for empty buckets, which no share visits).  The top window's threads are those of
// k_var_buckets_flat.
static __global__ __launch_bounds__(64, 2) void k_var_buckets_seg(const uint4* __restrict__ points, const uint32_t* __restrict__ offsets,
                                                                  const uint32_t* __restrict__ entries, uint32_t regular, uint32_t E, uint32_t nseg,
                                                                  uint32_t top_n, uint32_t ktop, g1_xyzz28* __restrict__ bucket_sums,
                                                                  g1_xyzz28* __restrict__ top_partials, g1_xyzz28* __restrict__ seg_part) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  g1_xyzz28 acc;
  if (id >= nseg) {
    const uint64_t t = id - nseg;
    if (t >= (uint64_t)top_n * ktop) return;
    const uint32_t bkt = regular + (uint32_t)(t / ktop);
    var_bucket_chain(acc, points, entries, offsets[bkt], offsets[bkt + 1], (uint32_t)(t % ktop), ktop);
    top_partials[t] = acc;
    return;
  }
  const uint32_t T = offsets[regular];
  const uint64_t lo64 = id * E;
  if (lo64 >= T) return;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (T - lo > E) ? lo + E : T;
  // the bucket of entry lo: offsets[b] <= lo < offsets[b + 1]   (offsets[0] = 0 <= lo < T = offsets[regular])
  uint32_t b = 0, z = regular;
  while (z - b > 1u) {
    const uint32_t m = (b + z) >> 1;
    if (offsets[m] <= lo)
      b = m;
    else
      z = m;
  }
  bool started = offsets[b] == lo;  // the bucket's list begins with this share's first entry
  uint32_t bstop = offsets[b + 1];  // end of the current bucket's list
  // ... and of the next one's, read one bucket AHEAD: a load issued at a boundary and consumed there would wait, in order, for the
  // point gather issued just before it -- in nearly every step of a wave, since some lane of 64 is at a boundary in nearly every step
  uint32_t bnext = offsets[(b + 2u < regular) ? b + 2u : regular];
  xyzz28_set_inf(acc);
  fp_t nx, ny;
  uint32_t ne = entries[lo];
  load_affine96(nx, ny, points, ne >> 1);
#pragma unroll 1
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t e = ne;
    const bool skip = bn_is_zero(nx) && bn_is_zero(ny);  // an all-zero entry is the point at infinity (var_bucket_chain)
    fp28 cx, cy;
    f28_load_entry(cx, cy, nx, ny, (e & 1u) != 0);
    if (k + 1u < hi) {
      ne = entries[k + 1u];
      load_affine96(nx, ny, points, ne >> 1);
    }
    if (!skip) {
      if (acc.inf) {  // a bucket's first point, inline: some lane of a wave is at one in nearly every step
        acc.x = cx;
        acc.y = cy;
        f28_normalize(acc.y);  // a negated y (2p - y, limbs < 2^29) -> limbs < 2^28: N-form
        acc.zz = f28_one();
        acc.zzz = acc.zz;
        acc.inf = 0;
      } else if (!xyzz28_madd_fast(acc, cx, cy)) {
        g1_xyzz28 tmp = acc;  // copy: the call takes addresses
        fp_t rx, ry;
        load_affine96(rx, ry, points, e >> 1);
        fp28 sx, sy;
        f28_load_entry(sx, sy, rx, ry, (e & 1u) != 0);
        xyzz28_madd_complete(tmp, sx, sy);
        acc = tmp;
      }
    }
    if (k + 1u == bstop || k + 1u == hi) {  // the bucket's list, or the share, ends here
      const bool ends = bstop <= hi;        // ... the bucket's list (it may end exactly with the share)
      g1_xyzz28* dst = (started && ends) ? bucket_sums + b : seg_part + (2 * id + (started ? 1u : 0u));
      *dst = acc;
      if (k + 1u < hi) {  // next non-empty bucket (it begins at k + 1)
        b++;
        bstop = bnext;
        while (bstop == k + 1u) {  // an empty bucket (e^-16 of them at 65,536 triples)
          b++;
          bstop = offsets[b + 1];
        }
        bnext = offsets[(b + 2u < regular) ? b + 2u : regular];
        started = true;
        xyzz28_set_inf(acc);
      }
    }
  }
}
// one thread per full-window bucket: the identity for an empty one, the sum of the partial sums of one that crosses share boundaries
static __global__ __launch_bounds__(64) void k_var_seg_fixup(const uint32_t* __restrict__ offsets, uint32_t regular, uint32_t E,
                                                             const g1_xyzz28* __restrict__ seg_part, g1_xyzz28* __restrict__ bucket_sums) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= regular) return;
  const uint32_t lo = offsets[b], hi = offsets[b + 1];
  g1_xyzz28 acc;
  if (lo == hi) {
    xyzz28_set_inf(acc);
    bucket_sums[b] = acc;
    return;
  }
  const uint32_t first = lo / E, last = (hi - 1u) / E;
  if (first == last) return;  // inside one share: that lane wrote the sum
  acc = seg_part[2 * (uint64_t)first + 1u];
#pragma unroll 1
  for (uint32_t l = first + 1u; l <= last; l++) {
    const g1_xyzz28 other = seg_part[2 * (uint64_t)l];
    xyzz28_add_complete_inl<true>(acc, other);
  }
  bucket_sums[b] = acc;
}

// One wave folds the K partial sums of 64/K buckets: K is a power of two <= 64, lanes
// [g*K, (g+1)*K) hold bucket g's partials and are summed by a segmented tree through LDS.
// (Radix-2^28 field throughout: a full XYZZ addition is ~7.5 k instead of ~11 k VALU instructions, and these two kernels
// are chains of dependent additions.)
static __global__ __launch_bounds__(64) void k_var_fold(const g1_xyzz28* __restrict__ partial_sums, uint32_t nbuckets, uint32_t K,
                                                        g1_xyzz28* __restrict__ bucket_sums) {
  __shared__ g1_xyzz28 lds[32];
  const int lane = threadIdx.x;
  const uint64_t id = (uint64_t)blockIdx.x * 64 + lane;
  const uint32_t bkt = (uint32_t)(id / K);
  g1_xyzz28 acc;
  if (bkt < nbuckets)
    acc = partial_sums[id];
  else
    xyzz28_set_inf(acc);
#pragma unroll 1
  for (uint32_t step = 1; step < K; step <<= 1) {
    const uint32_t m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = acc;
    __syncthreads();
    if ((lane & m) == 0) {
      const g1_xyzz28 other = lds[(lane + step) >> 1];
      xyzz28_add_complete_inl<true>(acc, other);  // inlined: the out-of-line adder passes both operands through scratch on every level
    }
    __syncthreads();
  }
  if ((lane & (K - 1)) == 0 && bkt < nbuckets) bucket_sums[bkt] = acc;
}

// One wave per window:  sum_d d * B_d  =  sum over j of the suffix sums  sum_{d >= j} B_d .
// Lane l owns `per` consecutive buckets; local suffix sums, a 6-step suffix scan of the lane
// totals across the wave (Hillis-Steele through LDS), then a tree sum of all suffix sums:
// ~2*per + 13 sequential additions, no scalar multiplications.  The window sum leaves in the 12 x 32-limb format
// (host Horner).
static __global__ __launch_bounds__(64) void k_var_windows(const g1_xyzz28* __restrict__ bucket_sums, VarGeom g, g1_xyzz* __restrict__ window_sums) {
  __shared__ g1_xyzz28 buf[2][64];
  __shared__ g1_xyzz28 lds[32];
  const int lane = threadIdx.x;
  const uint32_t j = blockIdx.x;
  const uint32_t per = (g.half + 63) / 64;
  const g1_xyzz28* B = bucket_sums + (uint64_t)j * g.half;
  // local pass (descending): run = suffix sum within the lane, tot = sum of the lane's suffix sums
  g1_xyzz28 run, tot;
  xyzz28_set_inf(run);
  xyzz28_set_inf(tot);
  uint32_t owned = 0;
  for (int k = (int)per - 1; k >= 0; k--) {
    const uint32_t idx = lane * per + k;
    if (idx < g.half) {
      g1_xyzz28 b = B[idx];
      xyzz28_add_complete_inl<true>(run, b);
      xyzz28_add_complete_inl<true>(tot, run);
      owned++;
    }
  }
  // exclusive suffix scan of `run` over lanes: X_l = sum_{l' > l} run_{l'}
  int cur = 0;
  buf[0][lane] = run;
  __syncthreads();
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    g1_xyzz28 v = buf[cur][lane];
    if (lane + off < 64) {
      g1_xyzz28 o = buf[cur][lane + off];
      xyzz28_add_complete_inl<true>(v, o);
    }
    buf[cur ^ 1][lane] = v;
    __syncthreads();
    cur ^= 1;
  }
  // inclusive suffix sum is buf[cur][lane]; exclusive = that of lane+1
  g1_xyzz28 X;
  if (lane + 1 < 64)
    X = buf[cur][lane + 1];
  else
    xyzz28_set_inf(X);
  // each of the lane's `owned` suffix sums gains X: tot += owned * X  (owned <= per, tiny)
  for (uint32_t k = 0; k < owned; k++) xyzz28_add_complete_inl<true>(tot, X);
#pragma unroll 1
  for (int step = 1; step < 64; step <<= 1) {
    const int m = 2 * step - 1;
    if ((lane & m) == step) lds[lane >> 1] = tot;
    __syncthreads();
    if ((lane & m) == 0) {
      g1_xyzz28 other = lds[(lane + step) >> 1];
      g1_xyzz28 mine = tot;
      xyzz28_add_complete_inl<true>(mine, other);
      tot = mine;
    }
    __syncthreads();
  }
  if (lane == 0) {
    g1_xyzz out;
    xyzz28_to_xyzz(out, tot);
    window_sums[j] = out;
  }
}

// flat path: T[j][b] = sum of the buckets B[j][d] whose magnitude d has bit b set, so that
//   sum_j 2^(c j) sum_d d B[j][d]  =  sum_{j,b} 2^(c j + b) T[j][b]
// -- the per-window running sums (a chain of 2 * half dependent additions, or ~50 even when cut into segments) become W*c
// independent tree sums of depth log2(256) + 8, and the weights 2^(c j + b) are the host's Horner loop, which doubles once
// per bit anyway.  One 256-thread workgroup per (window, bit): 260 workgroups of four 248-VGPR waves are one round on 256 CUs
// (512 threads took two).  The top window's buckets arrive as ktop partial sums each
// (top_partials[(d - 1) * ktop + k]) and are summed here directly: top_n * ktop = half items, the same depth as a full window.
static __global__ __launch_bounds__(256) void k_var_bitsums(const g1_xyzz28* __restrict__ bucket_sums, const g1_xyzz28* __restrict__ top_partials, VarGeom g,
                                                            g1_xyzz* __restrict__ out) {
  __shared__ g1_xyzz28 lds[128];
  const uint32_t j = blockIdx.x / g.c, b = blockIdx.x % g.c;
  const bool top = (j + 1 == g.W);
  const uint32_t limit = top ? g.top_n : g.half;
  const uint32_t per = top ? g.ktop : 1u;  // stored points per bucket
  const g1_xyzz28* B = top ? top_partials : bucket_sums + (uint64_t)j * g.half;
  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  const uint32_t low_mask = (1u << b) - 1u;
#pragma unroll 1
  for (uint32_t i = threadIdx.x; i < (limit / 2u + 1u) * per; i += 256u) {
    const uint32_t di = i / per, k = i % per;
    const uint32_t d = ((di >> b) << (b + 1u)) | (1u << b) | (di & low_mask);  // the di-th number with bit b set
    if (d > limit) break;  // d grows with i
    const g1_xyzz28 t = B[(uint64_t)(d - 1u) * per + k];
    xyzz28_add_complete_inl<true>(acc, t);  // inlined: the out-of-line adder passes both operands through scratch on every step
  }
#pragma unroll 1
  for (uint32_t step = 128; step >= 1; step >>= 1) {
    if (threadIdx.x >= step && threadIdx.x < 2 * step) lds[threadIdx.x - step] = acc;
    __syncthreads();
    if (threadIdx.x < step) {
      const g1_xyzz28 other = lds[threadIdx.x];
      xyzz28_add_complete_inl<true>(acc, other);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    g1_xyzz o;
    xyzz28_to_xyzz(o, acc);
    out[blockIdx.x] = o;
  }
}

#endif
}  // namespace kzg
