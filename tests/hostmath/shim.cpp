// CPU build of the product's single-source field/curve/hash headers, for unit
// tests only (tests/test_hostmath.py).  NOT part of the product library: the
// shipped .so has no CPU compute path for the batch entry points.
#include <stdint.h>
#include <string.h>

#include "../../kateth_amd/csrc/g1.cuh"
#include "../../kateth_amd/csrc/sha256.cuh"

using namespace kzg;

template <int N>
static void load_le(bn<N>& r, const uint8_t* p) {
  for (int i = 0; i < N; i++) r.v[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
}
template <int N>
static void store_le(uint8_t* p, const bn<N>& a) {
  for (int i = 0; i < N; i++) {
    p[4 * i] = a.v[i];
    p[4 * i + 1] = a.v[i] >> 8;
    p[4 * i + 2] = a.v[i] >> 16;
    p[4 * i + 3] = a.v[i] >> 24;
  }
}

template <class F>
static void binop(int op, uint8_t* out, const uint8_t* a, const uint8_t* b) {
  bn<F::N> x, y, r;
  load_le(x, a);
  load_le(y, b);
  to_mont<F>(x, x);
  to_mont<F>(y, y);
  switch (op) {
    case 0: mont_mul<F>(r, x, y); break;
    case 1: add_mod<F>(r, x, y); break;
    case 2: sub_mod<F>(r, x, y); break;
    case 3: neg_mod<F>(r, x); break;
    case 4: mont_sqr<F>(r, x); break;
    default: bn_zero(r);
  }
  from_mont<F>(r, r);
  store_le(out, r);
}

extern "C" {
void hm_fp_op(int op, uint8_t* out48, const uint8_t* a48, const uint8_t* b48) {
  if (op == 5) {
    fp_t x, r;
    load_le(x, a48);
    to_mont<FpParams>(x, x);
    fp_inv(r, x);
    from_mont<FpParams>(r, r);
    store_le(out48, r);
    return;
  }
  if (op == 6) {
    fp_t x, r;
    load_le(x, a48);
    to_mont<FpParams>(x, x);
    fp_sqrt_candidate(r, x);
    from_mont<FpParams>(r, r);
    store_le(out48, r);
    return;
  }
  binop<FpParams>(op, out48, a48, b48);
}
void hm_fr_op(int op, uint8_t* out32, const uint8_t* a32, const uint8_t* b32) {
  if (op == 5) {
    fr_t x, r;
    load_le(x, a32);
    to_mont<FrParams>(x, x);
    fr_inv(r, x);
    from_mont<FrParams>(r, r);
    store_le(out32, r);
    return;
  }
  binop<FrParams>(op, out32, a32, b32);
}
// sum of n compressed points via decompress + xyzz_madd ; returns first decode status
int32_t hm_g1_sum(uint8_t* out48, const uint8_t* pts48, int n, int check_subgroup) {
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    int32_t st = check_subgroup ? g1_decompress(x, y, inf, pts48 + 48 * i) : g1_uncompress(x, y, inf, pts48 + 48 * i);
    if (st) return st;
    if (!inf) xyzz_madd(acc, x, y);
  }
  g1_compress_xyzz(out48, acc);
  return 0;
}
// out = a + b through the full XYZZ+XYZZ adder (both operands scaled to non-trivial ZZ)
int32_t hm_g1_add_full(uint8_t* out48, const uint8_t* a48, const uint8_t* b48) {
  fp_t x, y;
  bool inf;
  g1_xyzz A, B;
  int32_t st = g1_uncompress(x, y, inf, a48);
  if (st) return st;
  if (inf) xyzz_set_inf(A); else { xyzz_mdbl(A, x, y); fp_neg(y, y); xyzz_madd(A, x, y); }  // A = 2a - a (non-trivial ZZ)
  st = g1_uncompress(x, y, inf, b48);
  if (st) return st;
  if (inf) xyzz_set_inf(B); else { xyzz_mdbl(B, x, y); fp_neg(y, y); xyzz_madd(B, x, y); }
  xyzz_add(A, B);
  g1_compress_xyzz(out48, A);
  return 0;
}
int32_t hm_g1_mul(uint8_t* out48, const uint8_t* a48, const uint8_t* k32be) {
  fp_t x, y;
  bool inf;
  int32_t st = g1_uncompress(x, y, inf, a48);
  if (st) return st;
  fr_t k;
  fr_from_be_bytes_plain(k, k32be);
  g1_xyzz acc;
  if (inf) xyzz_set_inf(acc); else g1_mul_affine(acc, x, y, k);
  g1_compress_xyzz(out48, acc);
  return 0;
}
int32_t hm_g1_decompress_status(const uint8_t* a48) {
  fp_t x, y;
  bool inf;
  return g1_decompress(x, y, inf, a48);
}
void hm_sha256(uint8_t* out32, const uint8_t* msg, uint64_t len) { sha256_bytes(out32, msg, len); }
void hm_hash_to_fr(uint8_t* out32be, const uint8_t* msg, uint64_t len) {
  uint8_t d[32];
  sha256_bytes(d, msg, len);
  fr_t v;
  fr_from_be_bytes_plain(v, d);
  fr_reduce_256(v);
  fr_to_be_bytes_plain(out32be, v);
}
}
