// Host-side G1 pieces of a SINGLE-item verification (engine_verify.hip, verify_one_on_host): the decoder's output format back
// to the host's, and [k1] P - [k2] G as one double-scalar multiplication.  Header-only so that tests/hostmath can check them
// against the oracle without a GPU.  Replaces, for n = 1, the two P1::lincomb calls of Setup::verify_proof_batch
// (src/kzg/setup.rs:139-156).
#pragma once
#include "g1.cuh"
#include "pairing.hpp"

namespace kzg {
namespace host {

// stored v 2^392 mod p (the variable-base MSM's operand format)  ->  canonical 2^384-Montgomery:  times 2^-8 = mont_mul by 2^376
inline fp_t r392_to_r384_factor() {
  fp_t v = fp_one();  // 2^384 mod p
  const fp_t pm = modulus<FpParams>();
  for (int k = 0; k < 8; k++) {  // halve modulo p
    uint32_t top = 0;
    if (v.v[0] & 1u) {
      fp_t t;
      top = bn_add(t, v, pm);
      v = t;
    }
    for (int i = 0; i < 11; i++) v.v[i] = (v.v[i] >> 1) | (v.v[i + 1] << 31);
    v.v[11] = (v.v[11] >> 1) | (top << 31);
  }
  return v;
}
inline void host_point_from_r392(g1_host_affine& out, const uint32_t* xy24, bool inf) {
  static const fp_t k = r392_to_r384_factor();
  out.inf = inf;
  if (inf) {
    bn_zero(out.x);
    bn_zero(out.y);
    return;
  }
  fp_t x, y;
  for (int i = 0; i < 12; i++) {
    x.v[i] = xy24[i];
    y.v[i] = xy24[12 + i];
  }
  fp_mul(out.x, x, k);
  fp_mul(out.y, y, k);
}
// multiples 1..15 of a finite affine point
inline void host_window_table(g1_xyzz* tab15, const g1_host_affine& p) {
  xyzz_from_affine(tab15[0], p.x, p.y);
  for (int d = 1; d < 15; d++) {
    tab15[d] = tab15[d - 1];
    xyzz_madd(tab15[d], p.x, p.y);
  }
}
struct NegGeneratorTable {
  g1_xyzz t[15];
  NegGeneratorTable() {
    const uint32_t gx[12] = KZG_FP_G1X_R392, gy[12] = KZG_FP_G1Y_R392;
    uint32_t xy[24];
    for (int q = 0; q < 12; q++) {
      xy[q] = gx[q];
      xy[12 + q] = gy[q];
    }
    g1_host_affine g;
    host_point_from_r392(g, xy, false);
    fp_neg(g.y, g.y);
    host_window_table(t, g);
  }
};
// [k1] P + [k2] (-G): plain scalars, P finite or infinity
inline void host_double_scalar_mul(g1_xyzz& acc, const fr_t& k1, const g1_host_affine& p, const fr_t& k2) {
  static const NegGeneratorTable neg_g;
  g1_xyzz tab[15];
  if (!p.inf) host_window_table(tab, p);
  xyzz_set_inf(acc);
  for (int w = 63; w >= 0; w--) {
    for (int k = 0; k < 4; k++) xyzz_dbl(acc);
    const uint32_t d1 = (k1.v[w >> 3] >> (4 * (w & 7))) & 15u, d2 = (k2.v[w >> 3] >> (4 * (w & 7))) & 15u;
    if (d1 && !p.inf) xyzz_add(acc, tab[d1 - 1]);
    if (d2) xyzz_add(acc, neg_g.t[d2 - 1]);
  }
}

}  // namespace host
}  // namespace kzg
