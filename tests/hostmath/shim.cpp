// CPU build of the product's single-source field/curve/hash headers, for unit
// tests only (tests/test_hostmath.py).  NOT part of the product library: the
// shipped .so has no CPU compute path for the batch entry points.
#include <stdint.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>
#define KZG_FP28_CHECK 1
#include "../../kateth_amd/csrc/fp28.cuh"
#include "../../kateth_amd/csrc/fp30.cuh"
#include "../../kateth_amd/csrc/fr29.cuh"
#include "../../kateth_amd/csrc/g1_decode28.cuh"
#include "../../kateth_amd/csrc/modinv30.cuh"
#include "../../kateth_amd/csrc/sha256.cuh"
#include "../../kateth_amd/csrc/multi_split.hpp"
#include "../../kateth_amd/csrc/glv.cuh"

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
// group contexts: the split of a batch over the members and the merge of their first-error records (multi_split.hpp)
// out = (member, first, count) triples; returns the number of shares
int hm_multi_shares(uint64_t n, uint32_t members, uint32_t rotate, uint64_t* out, int cap) {
  const std::vector<kzg::multi::Share> sh = kzg::multi::shares_of(n, members, rotate);
  for (size_t j = 0; j < sh.size() && (int)j < cap; j++) {
    out[3 * j] = sh[j].member;
    out[3 * j + 1] = sh[j].first;
    out[3 * j + 2] = sh[j].count;
  }
  return (int)sh.size();
}
int32_t hm_multi_first_error(uint64_t n, uint32_t members, uint32_t rotate, const int32_t* err6) {
  return kzg::multi::merged_first_error(kzg::multi::shares_of(n, members, rotate), err6);
}
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

// ---- host pairing (kateth_amd/csrc/pairing.hpp) test hooks ---------------------------------
#include "../../kateth_amd/csrc/pairing.hpp"
using namespace kzg::host;

static bool load_g1(g1_host_affine& p, const uint8_t* in48) {
  fp_t x, y;
  bool inf;
  if (g1_uncompress(x, y, inf, in48) != 0) return false;
  p.x = x;
  p.y = y;
  p.inf = inf;
  return true;
}

extern "C" {
// e(-a1, q1) * e(b1, q2) == 1 with arbitrary G2 points (lines computed on the fly)
int32_t hm_pairing_check(const uint8_t* a48, const uint8_t* q1_96, const uint8_t* b48, const uint8_t* q2_96) {
  g1_host_affine a, b;
  g2_affine q1, q2;
  if (!load_g1(a, a48) || !load_g1(b, b48)) return -1;
  if (g2_decompress(q1, q1_96) != 0 || g2_decompress(q2, q2_96) != 0) return -2;
  static frob_consts fc = make_frob_consts();
  miller_lines l1 = precompute_lines(q1), l2 = precompute_lines(q2);
  g1_host_affine na = a;
  if (!na.inf) fp_neg(na.y, na.y);
  g1_host_affine ps[2] = {na, b};
  const miller_lines* ls[2] = {&l1, &l2};
  fp12 f = multi_miller(ps, ls, 2);
  return final_exp_is_one(f, fc) ? 1 : 0;
}
int32_t hm_g2_decompress_status(const uint8_t* in96) {
  g2_affine q;
  return g2_decompress(q, in96);
}
// compares the addition-chain final exponentiation with the definition f^((p^12-1)/r)
// on the Miller-loop value of (a, q); returns 1 if both agree on "== 1", plus 2 if that value is one
int32_t hm_final_exp_crosscheck(const uint8_t* a48, const uint8_t* q96, const uint8_t* exp_le, int exp_limbs) {
  g1_host_affine a;
  g2_affine q;
  if (!load_g1(a, a48) || g2_decompress(q, q96) != 0) return -1;
  static frob_consts fc = make_frob_consts();
  miller_lines l = precompute_lines(q);
  const miller_lines* ls[1] = {&l};
  fp12 f = multi_miller(&a, ls, 1);
  std::vector<uint32_t> e(exp_limbs);
  for (int i = 0; i < exp_limbs; i++) e[i] = (uint32_t)exp_le[4 * i] | ((uint32_t)exp_le[4 * i + 1] << 8) | ((uint32_t)exp_le[4 * i + 2] << 16) | ((uint32_t)exp_le[4 * i + 3] << 24);
  bool slow = f12_is_one(f12_pow_big(f, e));
  bool fast = final_exp_is_one(f, fc);
  return (slow == fast ? 1 : 0) | (fast ? 2 : 0);
}
// frobenius sanity: frob^12 == identity and frob(f) == f^p is checked through frob^6 == conj
int32_t hm_frobenius_check(const uint8_t* a48, const uint8_t* q96) {
  g1_host_affine a;
  g2_affine q;
  if (!load_g1(a, a48) || g2_decompress(q, q96) != 0) return -1;
  static frob_consts fc = make_frob_consts();
  miller_lines l = precompute_lines(q);
  const miller_lines* ls[1] = {&l};
  fp12 f = multi_miller(&a, ls, 1);
  fp12 g = f;
  for (int i = 0; i < 6; i++) g = f12_frobenius(g, fc);
  if (!f12_eq(g, f12_conj(f))) return 0;
  fp12 inv = f12_inv(f);
  if (!f12_is_one(f12_mul(inv, f))) return 0;
  if (!f12_eq(f12_sqr(f), f12_mul(f, f))) return 0;
  return 1;
}
}

extern "C" int32_t hm_g1_subgroup_both(const uint8_t* a48) {  // bit0 = fast test, bit1 = naive [r]P test
  fp_t x, y;
  bool inf;
  if (g1_uncompress(x, y, inf, a48) != 0) return -1;
  return (g1_in_subgroup(x, y, inf) ? 1 : 0) | (g1_in_subgroup_naive(x, y, inf) ? 2 : 0);
}

// lazy-reduction mixed add (MSM hot loop) against the canonical one over a chain of points
extern "C" int32_t hm_g1_sum_lazy(uint8_t* out48, const uint8_t* pts48, int n) {
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    int32_t st = g1_uncompress(x, y, inf, pts48 + 48 * i);
    if (st) return st;
    if (!inf) xyzz_madd_lazy(acc, x, y);
  }
  xyzz_canonicalize(acc);
  g1_compress_xyzz(out48, acc);
  return 0;
}

// ---- radix-2^28 field / adder of the fixed-base MSM (kateth_amd/csrc/fp28.cuh) ----------------------
static int g_fp28_violations = 0;
extern "C" void kzg_fp28_check_failed(const char* what) {
  g_fp28_violations++;
  fprintf(stderr, "fp28 bound violated: %s\n", what);
}
extern "C" int32_t hm_f28_violations() { return g_fp28_violations; }

static void f28_in(fp28& r, const uint8_t* a48) {  // plain value -> x * 2^392 mod p, 14 x 28
  fp_t x;
  load_le(x, a48);
  to_mont<FpParams>(x, x);
  fp_to_r392(x, x);
  f28_from_bn(r, x);
}
static void f28_out(uint8_t* out48, const fp28& a) {
  fp_t x;
  f28_to_fp(x, a);
  from_mont<FpParams>(x, x);
  store_le(out48, x);
}
// op 0: a*b, 1: a^2, 2: a*b + c*d, 3: round trip, 4: a + 4p - b, 5: carry pass of (a + 16p - b),
// 6: is_zero_exact(a + 4p - b) -> out[0]
extern "C" void hm_f28_op(int op, uint8_t* out48, const uint8_t* a48, const uint8_t* b48, const uint8_t* c48, const uint8_t* d48) {
  fp28 a, b, c, d, r;
  f28_in(a, a48);
  f28_in(b, b48);
  f28_in(c, c48);
  f28_in(d, d48);
  switch (op) {
    case 0: f28_mul(r, a, b); break;
    case 1: f28_sqr(r, a); break;
    case 2: f28_mul2(r, a, b, c, d); break;
    case 3: r = a; break;
    case 4: f28_sub_4p(r, a, b); break;
    case 5:
      f28_sub_16p(r, a, b);
      f28_carry_pass(r);
      break;
    case 6: {
      f28_sub_4p(r, a, b);
      bool z = f28_maybe_zero(r) && f28_is_zero_exact(r);
      memset(out48, 0, 48);
      out48[0] = z ? 1 : 0;
      return;
    }
    default: r = a;
  }
  f28_out(out48, r);
}
// sum of +-points through xyzz28_madd (the MSM hot-loop adder); signs[i] != 0 negates point i
static int g_fp28_slow_calls = 0;
extern "C" int32_t hm_f28_slow_calls() { return g_fp28_slow_calls; }
extern "C" int32_t hm_g1_sum28(uint8_t* out48, const uint8_t* pts48, const uint8_t* signs, int n, int force_complete) {
  g1_xyzz28 acc;
  xyzz28_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    int32_t st = g1_uncompress(x, y, inf, pts48 + 48 * i);
    if (st) return st;
    if (inf) continue;
    fp_to_r392(x, x);
    fp_to_r392(y, y);
    fp28 x2, y2;
    f28_load_entry(x2, y2, x, y, signs[i] != 0);
    // the kernel's flow: hot path first, complete adder when it declines (or when the accumulator is the identity)
    if (force_complete || acc.inf || !xyzz28_madd_fast(acc, x2, y2)) {
      g_fp28_slow_calls++;
      xyzz28_madd_complete(acc, x2, y2);
    }
  }
  g1_xyzz r;
  xyzz28_to_xyzz(r, acc);
  g1_compress_xyzz(out48, r);
  return 0;
}

// ---- radix-2^29 Fr of the verification's evaluation kernel (kateth_amd/csrc/fr29.cuh) ---------------
static void f29_in(fr29& r, const uint8_t* a32, bool mont) {  // plain value -> 9 x 29 limbs (optionally x 2^261)
  fr_t x;
  load_le(x, a32);
  f29_from_bn(r, x);
  if (mont) f29_to_mont(r, r);
}
// op 0: a*b (Montgomery), 1: a^2, 2: a*b + c*d, 3: round trip, 4: plain a + 2r - b then reduced through a product,
// 5: is_zero(a + 2r - b) -> out[0], 6: the pair update of k_eval_frac (see tests)
extern "C" void hm_f29_op(int op, uint8_t* out32, const uint8_t* a32, const uint8_t* b32, const uint8_t* c32, const uint8_t* d32) {
  fr29 a, b, c, d, r;
  fr_t o;
  if (op == 3) {
    f29_in(a, a32, false);
    f29_to_bn(o, a);
    store_le(out32, o);
    return;
  }
  if (op == 4 || op == 5) {
    f29_in(a, a32, true);
    f29_in(b, b32, false);
    fr29 bm;
    f29_to_mont(bm, b);
    fr_t bc;
    f29_to_canonical_bn(bc, bm);  // canonical Montgomery b
    f29_from_bn(bm, bc);
    f29_sub_2r(r, a, bm);
    if (op == 5) {
      memset(out32, 0, 32);
      out32[0] = (f29_maybe_zero(r) && f29_is_zero_exact(r)) ? 1 : 0;
      return;
    }
    fr29 one;
    for (int i = 0; i < F29_N; i++) one.l[i] = i == 0 ? 1u : 0u;
    f29_mul(r, r, one);  // out of Montgomery form
    f29_to_canonical_bn(o, r);
    store_le(out32, o);
    return;
  }
  f29_in(a, a32, true);
  f29_in(b, b32, true);
  f29_in(c, c32, true);
  f29_in(d, d32, true);
  switch (op) {
    case 0: f29_mul(r, a, b); break;
    case 1: f29_sqr(r, a); break;
    case 2: f29_mul2(r, a, b, c, d); break;
    default: r = a;
  }
  fr29 one;
  for (int i = 0; i < F29_N; i++) one.l[i] = i == 0 ? 1u : 0u;
  f29_mul(r, r, one);
  f29_to_canonical_bn(o, r);
  store_le(out32, o);
}

// ---- point decoding in the radix-2^28 representation (kateth_amd/csrc/g1_decode28.cuh) ------------------
// returns g1_decompress28's status in the low byte and g1_decompress's (the 12 x 32-bit-limb reference path) in the
// next byte; out48 = re-encoding of what the radix-2^28 path decoded; bit 16 set if the two paths' (x, y) differ
extern "C" int32_t hm_g1_decompress28(uint8_t* out48, const uint8_t* in48) {
  fp_t x, y, x0, y0;
  bool inf = false, inf0 = false;
  const int32_t st = g1_decompress28(x, y, inf, in48);
  const int32_t st0 = g1_decompress(x0, y0, inf0, in48);
  int32_t diff = 0;
  if (st == 0 && st0 == 0) {
    if (inf != inf0 || !bn_eq(x, x0) || !bn_eq(y, y0)) diff = 1 << 16;
    g1_compress_affine(out48, x, y, inf);
  } else {
    memset(out48, 0, 48);
  }
  return (st & 0xff) | ((st0 & 0xff) << 8) | diff;
}

// compress(sum of points) through g1_compress_xyzz28 (radix-2^28 inversion) and through g1_compress_xyzz; returns 1 if the
// two encodings are identical, out48 = the radix-2^28 one
extern "C" int32_t hm_g1_sum_compress28(uint8_t* out48, const uint8_t* pts48, int n) {
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    if (g1_uncompress(x, y, inf, pts48 + 48 * i) != 0) return -1;
    if (!inf) xyzz_madd(acc, x, y);
  }
  uint8_t ref[48];
  g1_compress_xyzz(ref, acc);
  g1_compress_xyzz28(out48, nullptr, acc);
  return memcmp(ref, out48, 48) == 0 ? 1 : 0;
}

// the blst_p1_affine image (x || y, 2^384-Montgomery, little-endian limbs; infinity = zeros) of sum of points, as the
// affine-output entry points of the C ABI produce it
extern "C" int32_t hm_g1_sum_affine96(uint8_t* out96, uint8_t* out48, const uint8_t* pts48, int n) {
  g1_xyzz acc;
  xyzz_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    if (g1_uncompress(x, y, inf, pts48 + 48 * i) != 0) return -1;
    if (!inf) xyzz_madd(acc, x, y);
  }
  uint32_t aff[24];
  g1_compress_xyzz28(out48, aff, acc);
  memcpy(out96, aff, 96);
  return 0;
}

// ---- safegcd inversion (kateth_amd/csrc/modinv30.cuh): plain residue in, plain inverse out ----------------------
extern "C" void hm_modinv30(int which, uint8_t* out, const uint8_t* a) {
  if (which == 0) {
    fp_t x, r;
    load_le(x, a);
    modinv30<FpInv30>(r, x);
    store_le(out, r);
  } else {
    fr_t x, r;
    load_le(x, a);
    modinv30<FrInv30>(r, x);
    store_le(out, r);
  }
}

// ---- single-item verification's host lincomb (kateth_amd/csrc/host_lincomb.hpp) ---------------------------------------------
#include "../../kateth_amd/csrc/host_lincomb.hpp"
// out48 = compress([k1] P - [k2] G + C) with P, C given compressed and passed through the decoder's r392 output format
// (g1_decompress28(..., r392_out = true)) and host_point_from_r392, as verify_one_on_host does; k1, k2: 32 bytes big-endian
extern "C" int32_t hm_single_item_lincomb(uint8_t* out48, const uint8_t* k1_be, const uint8_t* p48, const uint8_t* k2_be, const uint8_t* c48) {
  kzg::host::g1_host_affine pts[2];
  const uint8_t* enc[2] = {p48, c48};
  for (int j = 0; j < 2; j++) {
    fp_t x, y;
    bool inf = false;
    if (g1_decompress28(x, y, inf, enc[j], true) != 0) return -1;
    uint32_t xy[24];
    for (int q = 0; q < 12; q++) {
      xy[q] = x.v[q];
      xy[12 + q] = y.v[q];
    }
    kzg::host::host_point_from_r392(pts[j], xy, inf);
  }
  fr_t k1, k2;
  for (int q = 0; q < 8; q++) {
    k1.v[7 - q] = ((uint32_t)k1_be[4 * q] << 24) | ((uint32_t)k1_be[4 * q + 1] << 16) | ((uint32_t)k1_be[4 * q + 2] << 8) | k1_be[4 * q + 3];
    k2.v[7 - q] = ((uint32_t)k2_be[4 * q] << 24) | ((uint32_t)k2_be[4 * q + 1] << 16) | ((uint32_t)k2_be[4 * q + 2] << 8) | k2_be[4 * q + 3];
  }
  g1_xyzz s;
  kzg::host::host_double_scalar_mul(s, k1, pts[0], k2);
  if (!pts[1].inf) xyzz_madd(s, pts[1].x, pts[1].y);
  g1_compress_xyzz(out48, s);
  return 0;
}

// ---- the host's mulx / adcx / adox Fp product (kateth_amd/csrc/host_fp_mulx.hpp) against the portable loop ------------------
// returns -1 when this CPU (or architecture) has no mulx path, else the number of disagreements over `count` chained products
extern "C" int32_t hm_host_mulx_crosscheck(uint64_t seed, int32_t count) {
#if defined(KZG_HOST_FP_MULX)
  if (!hostmulx::cpu_ok()) return -1;
  fp_t x, y;
  uint64_t st = seed | 1;
  auto next = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return (uint32_t)(st >> 16);
  };
  const fp_t p = modulus<FpParams>();
  auto fresh = [&](fp_t& v) {
    for (int i = 0; i < 12; i++) v.v[i] = next();
    v.v[11] &= 0x0fffffffu;  // < 2^380 < p
  };
  fresh(x);
  fresh(y);
  int32_t bad = 0;
  for (int32_t it = 0; it < count; it++) {
    fp_t fast, slow, lazy_fast, lazy_slow;
    host_fp_force_portable() = false;
    mont_mul<FpParams>(fast, x, y);
    mont_mul_lazy<FpParams>(lazy_fast, x, y);
    host_fp_force_portable() = true;
    mont_mul<FpParams>(slow, x, y);
    mont_mul_lazy<FpParams>(lazy_slow, x, y);
    host_fp_force_portable() = false;
    if (!bn_eq(fast, slow) || bn_geq(fast, p)) bad++;
    // lazy results may differ by p between the two paths; both must be < 2p and congruent
    fp_t a = lazy_fast, b = lazy_slow;
    if (bn_geq(a, p)) bn_sub(a, a, p);
    if (bn_geq(b, p)) bn_sub(b, b, p);
    if (!bn_eq(a, b) || !bn_eq(a, slow)) bad++;
    x = y;
    y = fast;
    if ((it & 255) == 0) fresh(y);
    if ((it & 1023) == 1) {  // edge operands: 0, 1, p - 1
      bn_zero(x);
    } else if ((it & 1023) == 3) {
      x = p;
      x.v[0] -= 1;
    }
  }
  return bad;
#else
  (void)seed;
  (void)count;
  return -1;
#endif
}

// ---- signed radix-2^30 Fp of the fixed-base MSM hot loop (kateth_amd/csrc/fp30.cuh) ------------------------------------------
// limbs in, limbs out (13 x int32): Python builds C-form, L-form and extreme operands itself and checks value and ranges.
// op 0: a*b, 1: a^2, 2: a*b + c*d, 3: carry pass, 4: wide carry pass, 5: pack -> unpack, 6: f30_to_fp (12 words out),
// 7: f30_from_bn (a13 = 12 words in), 8: is_zero(a) -> out[0], 9: fp_to_packed30 -> unpack (a13 = 12 words: canonical x*2^384)
extern "C" void hm_f30_op(int op, int32_t* out13, const int32_t* a13, const int32_t* b13, const int32_t* c13, const int32_t* d13) {
  fp30 a, b, c, d, r;
  for (int i = 0; i < 13; i++) {
    a.l[i] = a13[i];
    b.l[i] = b13[i];
    c.l[i] = c13[i];
    d.l[i] = d13[i];
    r.l[i] = 0;
  }
  switch (op) {
    case 0: f30_mul(r, a, b); break;
    case 1: f30_sqr(r, a); break;
    case 2: f30_mul2(r, a, b, c, d); break;
    case 3: r = a; f30_carry(r); break;
    case 4: r = a; f30_carry<true>(r); break;
    case 5: {
      uint32_t w[12];
      f30_pack(w, a);
      f30_unpack(r, w);
      break;
    }
    case 6: {
      fp_t v;
      f30_to_fp(v, a);
      for (int i = 0; i < 12; i++) out13[i] = (int32_t)v.v[i];
      out13[12] = 0;
      return;
    }
    case 7: {
      fp_t v;
      for (int i = 0; i < 12; i++) v.v[i] = (uint32_t)a13[i];
      f30_from_bn(r, v);
      break;
    }
    case 8: r.l[0] = f30_is_zero(a) ? 1 : 0; break;
    case 10: f30_mul_inj<-1>(r, a, b, c); break;          // a b / 2^390 - c
    case 11: f30_sqr_inj2<-1, -3>(r, a, c, d); break;     // a^2 / 2^390 - c - 3 d
    case 12: f30_mul_u(r, a, b); break;                   // a b / 2^390 with floor digits (U-form)
    case 9: {
      fp_t v;
      for (int i = 0; i < 12; i++) v.v[i] = (uint32_t)a13[i];
      uint32_t w[12];
      fp_to_packed30(w, v);
      f30_unpack(r, w);
      break;
    }
    default: r = a;
  }
  for (int i = 0; i < 13; i++) out13[i] = r.l[i];
}
// sum of +-points through xyzz30_madd (table-format entries: fp_to_packed30 / f30_load_entry), as k_msm_comb30 adds them;
// doublings > 0: the sum is doubled that many times through xyzz30_dbl at the end (the Horner step between bit planes)
static int g_fp30_slow_calls = 0;
extern "C" int32_t hm_f30_slow_calls() { return g_fp30_slow_calls; }
extern "C" int32_t hm_g1_sum30(uint8_t* out48, const uint8_t* pts48, const uint8_t* signs, int n, int force_complete, int doublings) {
  g1_xyzz30 acc;
  xyzz30_set_inf(acc);
  for (int i = 0; i < n; i++) {
    fp_t x, y;
    bool inf;
    int32_t st = g1_uncompress(x, y, inf, pts48 + 48 * i);
    if (st) return st;
    if (inf) continue;
    uint32_t wx[12], wy[12];
    fp_to_packed30(wx, x);
    fp_to_packed30(wy, y);
    fp30 x2, y2;
    f30_load_entry(x2, y2, wx, wy, xyzz30_entry_neg(acc, signs[i] != 0));
    if (force_complete || acc.inf || !xyzz30_madd_fast(acc, x2, y2)) {
      g_fp30_slow_calls++;
      xyzz30_madd_complete(acc, x2, y2);
    }
  }
  for (int k = 0; k < doublings; k++) xyzz30_dbl(acc);
  g1_xyzz r;
  xyzz30_to_xyzz(r, acc);
  g1_compress_xyzz(out48, r);
  return 0;
}

// ---- GLV split of a scalar (kateth_amd/csrc/glv.cuh): k (32 B little-endian) -> k1 || k2 (32 B each, little-endian) -------------
extern "C" void hm_glv_split(uint8_t* out64, const uint8_t* k32) {
  fr_t k, k1, k2;
  load_le(k, k32);
  glv_split(k1, k2, k);
  for (int i = 0; i < 8; i++)
    for (int b = 0; b < 4; b++) {
      out64[4 * i + b] = (uint8_t)(k1.v[i] >> (8 * b));
      out64[32 + 4 * i + b] = (uint8_t)(k2.v[i] >> (8 * b));
    }
}
