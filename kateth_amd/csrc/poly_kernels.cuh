// The proof path's polynomial kernels (compiled once: engine_proof.hip owns this header).
#pragma once
#include "field.cuh"
#include "issue_fair.cuh"
#include "scalar_load.cuh"

namespace kzg {
#if defined(__HIPCC__)

// ---------------------------------------------------------------------------
// K1 + K5 + K6: Blob::from_slice validation (src/blob.rs:26-37),
// Polynomial::evaluate (src/kzg/poly.rs:10-33) and the quotient of
// Polynomial::prove (src/kzg/poly.rs:44-66), one 512-thread workgroup per blob,
// 8 elements per thread held in registers.
//
// The reference performs one field inversion per element (4096 + 4096 per
// proof); here all 4096 denominators (z - w_i) are inverted with ONE inversion
// per blob: per-thread prefix products, a 512-leaf product tree in LDS, a single
// Fermat inversion of the root, and the inverse pushed back down the tree.
//   y   = (z^4096 - 1)/4096 * sum_i e_i w_i / (z - w_i)          (z outside the domain)
//   q_i = (e_i - y) / (w_i - z) = (y - e_i) * inv(z - w_i)
// In-domain z == w_m (poly.rs:14-18, :50-64): y = e_m and
//   q_m = w_m^-1 * sum_{j != m} (e_j - y) w_j / (w_m - w_j) = -w_m^-1 * sum_{j != m} q_j w_j .
// status[b] |= KZG_ERR_BLOB_INVALID_FIELD_ELEMENT when an element is >= r.
// Outputs are plain (non-Montgomery) little-endian limbs.
// ---------------------------------------------------------------------------
// The root of k_poly's product tree is prod_i (z - w_i) = z^4096 - 1 (with the matching factor
// replaced by 1 for an in-domain z = w_m: prod_{i != m} (w_m - w_i) = 4096 / w_m).  Its inverse is
// therefore computed here, one blob per lane, instead of serially inside every workgroup -- and with it the factor
// (z^4096 - 1) / 4096 of the barycentric sum, which needs the same twelve squarings: inv_root[2b] and inv_root[2b + 1].
static __global__ __launch_bounds__(64) void k_poly_root_inverse(const fr_t* __restrict__ z_plain, uint64_t n, fr_t* __restrict__ inv_root) {
  issue_priority_latency();
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  fr_t z, zn, r;
  to_mont<FrParams>(z, z_plain[b]);
  zn = z;
  for (int q = 0; q < 12; q++) fr_sqr(zn, zn);
  fr_sub(zn, zn, fr_one());
  fr_t f;
  {
    const uint32_t c4096[8] = KZG_FR_INV4096_MONT;
#pragma unroll
    for (int q = 0; q < 8; q++) f.v[q] = c4096[q];
  }
  if (bn_is_zero(zn)) {  // z is a 4096th root of unity: inverse of 4096 / z
    fr_mul(r, z, f);
  } else {
    fr_inv_inl(r, zn);
  }
  inv_root[2 * b] = r;
  fr_mul(f, f, zn);  // (z^4096 - 1) / 4096, Montgomery (zero for an in-domain z: y is the matching element then)
  inv_root[2 * b + 1] = f;
}

// (512, 4): at most 128 VGPRs, so that two 8-wave workgroups share a CU (129 VGPRs would halve the occupancy)
template <bool QUOTIENT>
static __global__ __launch_bounds__(512, 4) void k_poly(const uint8_t* __restrict__ blobs, const fr_t* __restrict__ z_plain,
                                              const fr_t* __restrict__ roots_brp, const fr_t* __restrict__ inv_root,
                                              fr_t* __restrict__ y_plain, fr_t* __restrict__ q_plain, int32_t* __restrict__ status) {
  __shared__ fr_t tree[1024];
  __shared__ int sh_domain;
  __shared__ int sh_bad;
  __shared__ fr_t sh_y;
  issue_priority_latency();  // short beside an MSM launch of another stream (host-buffer proof pipeline)
  const int t = threadIdx.x;
  const uint64_t b = blockIdx.x;
  const uint8_t* blob = blobs + b * 131072ull;
  if (t == 0) {
    sh_domain = -1;
    sh_bad = 0;
  }
  __syncthreads();
  fr_t z;
  to_mont<FrParams>(z, z_plain[b]);
  // The thread's eight blob elements are NOT kept in registers: with the eight prefix products they would be 128 VGPRs before any
  // temporary, the kernel's whole budget at four waves per SIMD (592 bytes of scratch per lane while they were); they are read
  // again where they are used -- twice more, 16-KiB coalesced rows that mostly still sit in the L2.  An element stays PLAIN:
  // mont_mul(plain, X*R) = plain*X, so neither a to_mont nor a from_mont per element is needed.
  auto element = [&](int k, bool& noncanonical) -> fr_t {
    uint32_t sc[8];
    load_scalar_be_(sc, blob + (uint64_t)(k * 512 + t) * 32u);
    fr_t v;
#pragma unroll
    for (int q = 0; q < 8; q++) v.v[q] = sc[q];
    noncanonical = !fr_is_canonical(v);
    if (noncanonical) bn_zero(v);
    return v;
  };
  fr_t pre[8];
  fr_t run = fr_one();
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int i = k * 512 + t;
    bool nc;
    (void)element(k, nc);
    bad |= nc;
    fr_t d;
    fr_sub(d, z, roots_brp[i]);
    if (bn_is_zero(d)) {
      sh_domain = i;  // at most one index can match
      d = fr_one();
    }
    fr_mul(run, run, d);
    pre[k] = run;
  }
  if (bad) sh_bad = 1;
  tree[512 + t] = run;
  __syncthreads();
  // product tree: node j = node 2j * node 2j+1
  for (int width = 256; width >= 1; width >>= 1) {
    if (t < width) {
      fr_t a = tree[2 * (width + t)], c = tree[2 * (width + t) + 1], r;
      fr_mul(r, a, c);
      tree[width + t] = r;
    }
    __syncthreads();
  }
  if (t == 0) tree[1] = inv_root[2 * b];  // = 1 / tree[1], from k_poly_root_inverse
  __syncthreads();
  // push inverses down: children of j get inv(j) * sibling product
  for (int width = 1; width <= 256; width <<= 1) {
    if (t < width) {
      const int j = width + t;
      fr_t ip = tree[j], a = tree[2 * j], c = tree[2 * j + 1], ra, rc;
      fr_mul(ra, ip, c);
      fr_mul(rc, ip, a);
      tree[2 * j] = ra;
      tree[2 * j + 1] = rc;
    }
    __syncthreads();
  }
  const int domain = sh_domain;
  fr_t inv_run = tree[512 + t];  // inverse of this thread's total product
  __syncthreads();
  fr_t ysum;
  bn_zero(ysum);
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    const int i = k * 512 + t;
    const fr_t w = roots_brp[i];
    fr_t d, inv_d, term;
    fr_sub(d, z, w);
    if (i == domain) d = fr_one();
    if (k == 0)
      inv_d = inv_run;
    else
      fr_mul(inv_d, inv_run, pre[k - 1]);
    fr_mul(inv_run, inv_run, d);
    pre[k] = inv_d;  // slot k now holds 1/(z - w_i)
    fr_mul(term, w, inv_d);      // (w R)(inv_d R)/R = w inv_d R
    bool nc;
    const fr_t ek = element(k, nc);
    fr_mul(term, ek, term);      // plain e * (w inv_d R) / R = plain e w / (z - w)
    if (i != domain) fr_add(ysum, ysum, term);
  }
  // block sum of ysum
  tree[t] = ysum;
  __syncthreads();
  for (int width = 256; width >= 1; width >>= 1) {
    if (t < width) {
      fr_t a = tree[t], c = tree[t + width], r;
      fr_add(r, a, c);
      tree[t] = r;
    }
    __syncthreads();
  }
  if (t == 0) {
    fr_t total = tree[0];
    fr_mul(total, total, inv_root[2 * b + 1]);  // plain sum * Montgomery (z^4096 - 1) / 4096 = plain y
    sh_y = total;
  }
  __syncthreads();
  if (domain >= 0 && (domain & 511) == t) {
    bool nc;
    sh_y = element(domain >> 9, nc);  // y = e_m (poly.rs:14-18)
  }
  __syncthreads();
  const fr_t y = sh_y;  // plain
  if (t == 0) {
    y_plain[b] = y;
    if (sh_bad) atomicOr(&status[b], KZG_ERR_BLOB_INVALID_FIELD_ELEMENT);
  }
  if (QUOTIENT) {
    fr_t* qout = q_plain + b * 4096ull;
    fr_t ssum;
    bn_zero(ssum);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = k * 512 + t;
      fr_t q;
      bool nc;
      const fr_t ek = element(k, nc);
      fr_sub(q, y, ek);          // plain
      fr_mul(q, q, pre[k]);      // plain * (1/(z - w_i)) R / R: plain quotient element
      if (i == domain) bn_zero(q);
      if (domain >= 0) {  // block-uniform
        fr_t qw;
        fr_mul(qw, q, roots_brp[i]);  // plain q_i w_i
        fr_add(ssum, ssum, qw);
      }
      qout[i] = q;
    }
    if (domain >= 0) {  // rare in-domain branch (poly.rs:50-64)
      __syncthreads();
      tree[t] = ssum;
      __syncthreads();
      for (int width = 256; width >= 1; width >>= 1) {
        if (t < width) {
          fr_t a = tree[t], c = tree[t + width], r;
          fr_add(r, a, c);
          tree[t] = r;
        }
        __syncthreads();
      }
      if (t == 0) {
        fr_t wm = roots_brp[domain], wi, qm;
        fr_inv_fermat(wi, wm);  // rare branch, one thread: the leaner out-of-line power keeps the kernel at 128 VGPRs
        fr_mul(qm, tree[0], wi);  // plain sum * Montgomery 1/w_m = plain
        fr_neg(qm, qm);
        qout[domain] = qm;
      }
    }
  }
}


#endif
}  // namespace kzg
