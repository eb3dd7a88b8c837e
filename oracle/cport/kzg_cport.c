/*
 * oracle/cport -- TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline).
 * Nothing under kateth_amd/ links or calls this; only tests/, smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Plain-C restatement of kateth's CPU path with the algorithm choices of its
 * dependency blst 0.3.x (not vendored under /root/reference; published
 * behaviour restated):
 *   - Fp / Fr: 64-bit-limb Montgomery arithmetic      (blst_fp / blst_fr, src/bls.rs:8-19)
 *   - G1: Jacobian add/double, XYZZ buckets            (blst_p1_add / pippenger.c)
 *   - MSM: Pippenger, window c = 10 for 4096 points, Booth-style signed digits,
 *     512 buckets per window, tiles spread over threads (p1_affines::mult, src/bls.rs:434)
 *   - every MSM call first re-normalises the 4096 bases (p1_affines::from, src/bls.rs:426)
 *   - per-element field inversion in evaluate / prove  (src/kzg/poly.rs:26,49 -> src/bls.rs:300-311)
 * Functions cite the kateth lines they follow.  PARITY STATUS: pinned against
 * oracle/pyref (tests/test_cport.py), which in turn is pinned by public KATs;
 * "parity unpinned" with respect to reference-owned vectors (none available).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ Fp --- */
#define NP 6
typedef struct { u64 l[NP]; } fp;
static const fp FP_P = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const u64 FP_INV = 0x89f3fffcfffcfffdULL;
static fp FP_ONE, FP_R2, FP_B; /* filled by init */

static inline int fp_is_zero(const fp* a) { u64 o = 0; for (int i = 0; i < NP; i++) o |= a->l[i]; return o == 0; }
static inline int fp_eq(const fp* a, const fp* b) { u64 o = 0; for (int i = 0; i < NP; i++) o |= a->l[i] ^ b->l[i]; return o == 0; }
static inline u64 fp_raw_sub(fp* r, const fp* a, const fp* b) {
  u64 br = 0;
  for (int i = 0; i < NP; i++) { u128 t = (u128)a->l[i] - b->l[i] - br; r->l[i] = (u64)t; br = (u64)(t >> 64) & 1; }
  return br;
}
static inline u64 fp_raw_add(fp* r, const fp* a, const fp* b) {
  u64 c = 0;
  for (int i = 0; i < NP; i++) { u128 t = (u128)a->l[i] + b->l[i] + c; r->l[i] = (u64)t; c = (u64)(t >> 64); }
  return c;
}
static inline void fp_add(fp* r, const fp* a, const fp* b) { fp t, s; u64 c = fp_raw_add(&t, a, b); u64 br = fp_raw_sub(&s, &t, &FP_P); *r = (c || !br) ? s : t; }
static inline void fp_sub(fp* r, const fp* a, const fp* b) { fp t, s; u64 br = fp_raw_sub(&t, a, b); fp_raw_add(&s, &t, &FP_P); *r = br ? s : t; }
static inline void fp_neg(fp* r, const fp* a) { if (fp_is_zero(a)) { *r = *a; return; } fp_raw_sub(r, &FP_P, a); }
static inline void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }

static inline void fp_mul(fp* r, const fp* a, const fp* b) {
  u64 t[NP + 2] = {0};
  for (int i = 0; i < NP; i++) {
    u64 c = 0;
    for (int j = 0; j < NP; j++) { u128 s = (u128)a->l[j] * b->l[i] + t[j] + c; t[j] = (u64)s; c = (u64)(s >> 64); }
    u128 s = (u128)t[NP] + c; t[NP] = (u64)s; t[NP + 1] = (u64)(s >> 64);
    u64 m = t[0] * FP_INV;
    s = (u128)m * FP_P.l[0] + t[0]; c = (u64)(s >> 64);
    for (int j = 1; j < NP; j++) { s = (u128)m * FP_P.l[j] + t[j] + c; t[j - 1] = (u64)s; c = (u64)(s >> 64); }
    s = (u128)t[NP] + c; t[NP - 1] = (u64)s; t[NP] = t[NP + 1] + (u64)(s >> 64);
  }
  fp x, y; memcpy(x.l, t, sizeof x.l);
  u64 br = fp_raw_sub(&y, &x, &FP_P);
  *r = (t[NP] || !br) ? y : x;
}
static inline void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static void fp_pow(fp* r, const fp* a, const u64* e, int nlimbs) {
  fp acc = FP_ONE;
  for (int i = nlimbs * 64 - 1; i >= 0; i--) { fp_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a); }
  *r = acc;
}
static void fp_inv(fp* r, const fp* a) { fp e = FP_P; e.l[0] -= 2; fp_pow(r, a, e.l, NP); }
static void fp_from_plain(fp* r, const fp* a) { fp_mul(r, a, &FP_R2); }
static void fp_to_plain(fp* r, const fp* a) { fp one = {{1, 0, 0, 0, 0, 0}}; fp_mul(r, a, &one); }
static void fp_from_be48(fp* r, const uint8_t* b) { for (int i = 0; i < NP; i++) { u64 v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | b[8 * (NP - 1 - i) + k]; r->l[i] = v; } }
static void fp_to_be48(uint8_t* b, const fp* a) { for (int i = 0; i < NP; i++) for (int k = 0; k < 8; k++) b[8 * (NP - 1 - i) + k] = (uint8_t)(a->l[i] >> (56 - 8 * k)); }

/* ------------------------------------------------------------------ Fr --- */
#define NR 4
typedef struct { u64 l[NR]; } fr;
static const fr FR_R = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const u64 FR_INV = 0xfffffffeffffffffULL;
static fr FR_ONE, FR_R2;

static inline int fr_is_zero(const fr* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_eq(const fr* a, const fr* b) { return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0; }
static inline u64 fr_raw_sub(fr* r, const fr* a, const fr* b) {
  u64 br = 0;
  for (int i = 0; i < NR; i++) { u128 t = (u128)a->l[i] - b->l[i] - br; r->l[i] = (u64)t; br = (u64)(t >> 64) & 1; }
  return br;
}
static inline u64 fr_raw_add(fr* r, const fr* a, const fr* b) {
  u64 c = 0;
  for (int i = 0; i < NR; i++) { u128 t = (u128)a->l[i] + b->l[i] + c; r->l[i] = (u64)t; c = (u64)(t >> 64); }
  return c;
}
static inline void fr_add(fr* r, const fr* a, const fr* b) { fr t, s; u64 c = fr_raw_add(&t, a, b); u64 br = fr_raw_sub(&s, &t, &FR_R); *r = (c || !br) ? s : t; }
static inline void fr_sub(fr* r, const fr* a, const fr* b) { fr t, s; u64 br = fr_raw_sub(&t, a, b); fr_raw_add(&s, &t, &FR_R); *r = br ? s : t; }
static inline void fr_mul(fr* r, const fr* a, const fr* b) {
  u64 t[NR + 2] = {0};
  for (int i = 0; i < NR; i++) {
    u64 c = 0;
    for (int j = 0; j < NR; j++) { u128 s = (u128)a->l[j] * b->l[i] + t[j] + c; t[j] = (u64)s; c = (u64)(s >> 64); }
    u128 s = (u128)t[NR] + c; t[NR] = (u64)s; t[NR + 1] = (u64)(s >> 64);
    u64 m = t[0] * FR_INV;
    s = (u128)m * FR_R.l[0] + t[0]; c = (u64)(s >> 64);
    for (int j = 1; j < NR; j++) { s = (u128)m * FR_R.l[j] + t[j] + c; t[j - 1] = (u64)s; c = (u64)(s >> 64); }
    s = (u128)t[NR] + c; t[NR - 1] = (u64)s; t[NR] = t[NR + 1] + (u64)(s >> 64);
  }
  fr x, y; memcpy(x.l, t, sizeof x.l);
  u64 br = fr_raw_sub(&y, &x, &FR_R);
  *r = (t[NR] || !br) ? y : x;
}
static void fr_from_plain(fr* r, const fr* a) { fr_mul(r, a, &FR_R2); }
static void fr_to_plain(fr* r, const fr* a) { fr one = {{1, 0, 0, 0}}; fr_mul(r, a, &one); }
static void fr_from_be32(fr* r, const uint8_t* b) { for (int i = 0; i < NR; i++) { u64 v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | b[8 * (NR - 1 - i) + k]; r->l[i] = v; } }
static void fr_to_be32(uint8_t* b, const fr* a) { for (int i = 0; i < NR; i++) for (int k = 0; k < 8; k++) b[8 * (NR - 1 - i) + k] = (uint8_t)(a->l[i] >> (56 - 8 * k)); }
/* blst_fr_eucl_inverse (src/bls.rs:305): restated as the Fermat inverse a^(r-2); see README for the cost note */
static void fr_inv(fr* r, const fr* a) {
  fr e = FR_R; e.l[0] -= 2;
  fr acc = FR_ONE;
  for (int i = 254; i >= 0; i--) { fr_mul(&acc, &acc, &acc); if ((e.l[i >> 6] >> (i & 63)) & 1) fr_mul(&acc, &acc, a); }
  *r = acc;
}
/* blst_fr_eucl_inverse (src/bls.rs:305) restated as a binary extended Euclid (HAC 14.61 style) on
   plain integers: the CPU baseline's per-element inversions (src/kzg/poly.rs:26,49) should cost what a
   Euclidean inverse costs, not what a 255-bit Fermat ladder costs.  Input/output Montgomery. */
static inline int fr_is_even(const fr* a) { return (a->l[0] & 1) == 0; }
static inline void fr_shr1(fr* a, u64 top) { for (int i = 0; i < NR - 1; i++) a->l[i] = (a->l[i] >> 1) | (a->l[i + 1] << 63); a->l[NR - 1] = (a->l[NR - 1] >> 1) | (top << 63); }
static inline int fr_geq_raw(const fr* a, const fr* b) { fr t; return !fr_raw_sub(&t, a, b); }
static void fr_eucl_inverse(fr* r, const fr* a_mont) {
  fr a; fr_to_plain(&a, a_mont);
  if (fr_is_zero(&a)) { memset(r, 0, sizeof *r); return; }
  fr u = a, v = FR_R, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  const fr one = {{1, 0, 0, 0}};
  while (!fr_eq(&u, &one) && !fr_eq(&v, &one)) {
    while (fr_is_even(&u)) {
      fr_shr1(&u, 0);
      if (fr_is_even(&x1)) fr_shr1(&x1, 0); else { u64 c = fr_raw_add(&x1, &x1, &FR_R); fr_shr1(&x1, c); }
    }
    while (fr_is_even(&v)) {
      fr_shr1(&v, 0);
      if (fr_is_even(&x2)) fr_shr1(&x2, 0); else { u64 c = fr_raw_add(&x2, &x2, &FR_R); fr_shr1(&x2, c); }
    }
    if (fr_geq_raw(&u, &v)) { fr_raw_sub(&u, &u, &v); fr_sub(&x1, &x1, &x2); }
    else { fr_raw_sub(&v, &v, &u); fr_sub(&x2, &x2, &x1); }
  }
  fr res = fr_eq(&u, &one) ? x1 : x2;  /* plain a^-1 */
  fr_from_plain(r, &res);
}

/* Fr::from_be_slice (src/bls.rs:130-139): 0 ok, 1 not in field */
static int fr_from_be_checked(fr* mont, const uint8_t* b) {
  fr p, t; fr_from_be32(&p, b);
  if (!fr_raw_sub(&t, &p, &FR_R)) return 1; /* p >= r */
  fr_from_plain(mont, &p);
  return 0;
}

/* ------------------------------------------------------------------ G1 --- */
typedef struct { fp x, y; int inf; } g1a;      /* affine, Montgomery */
typedef struct { fp x, y, z; } g1j;            /* Jacobian, z==0 infinity (blst_p1) */
typedef struct { fp x, y, zz, zzz; } g1x;      /* XYZZ bucket */

static void g1j_set_inf(g1j* p) { memset(p, 0, sizeof *p); }
static int g1j_is_inf(const g1j* p) { return fp_is_zero(&p->z); }
static void g1j_double(g1j* r, const g1j* p) { /* dbl-2009-l */
  if (g1j_is_inf(p)) { *r = *p; return; }
  fp a, b, c, d, e, f, t;
  fp_sqr(&a, &p->x); fp_sqr(&b, &p->y); fp_sqr(&c, &b);
  fp_add(&t, &p->x, &b); fp_sqr(&t, &t); fp_sub(&t, &t, &a); fp_sub(&t, &t, &c); fp_dbl(&d, &t);
  fp_dbl(&e, &a); fp_add(&e, &e, &a); fp_sqr(&f, &e);
  fp x3, y3, z3;
  fp_sub(&x3, &f, &d); fp_sub(&x3, &x3, &d);
  fp_mul(&z3, &p->y, &p->z); fp_dbl(&z3, &z3);
  fp_sub(&t, &d, &x3); fp_mul(&y3, &e, &t); fp_dbl(&c, &c); fp_dbl(&c, &c); fp_dbl(&c, &c); fp_sub(&y3, &y3, &c);
  r->x = x3; r->y = y3; r->z = z3;
}
/* complete Jacobian addition (the oracle uses the complete law; quirk Q3 inputs are outside the parity set) */
static void g1j_add(g1j* r, const g1j* p, const g1j* q) {
  if (g1j_is_inf(p)) { *r = *q; return; }
  if (g1j_is_inf(q)) { *r = *p; return; }
  fp z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  fp_sqr(&z1z1, &p->z); fp_sqr(&z2z2, &q->z);
  fp_mul(&u1, &p->x, &z2z2); fp_mul(&u2, &q->x, &z1z1);
  fp_mul(&s1, &p->y, &q->z); fp_mul(&s1, &s1, &z2z2);
  fp_mul(&s2, &q->y, &p->z); fp_mul(&s2, &s2, &z1z1);
  if (fp_eq(&u1, &u2)) { if (fp_eq(&s1, &s2)) { g1j_double(r, p); } else { g1j_set_inf(r); } return; }
  fp_sub(&h, &u2, &u1); fp_dbl(&i, &h); fp_sqr(&i, &i); fp_mul(&j, &h, &i);
  fp_sub(&rr, &s2, &s1); fp_dbl(&rr, &rr); fp_mul(&v, &u1, &i);
  fp x3, y3, z3;
  fp_sqr(&x3, &rr); fp_sub(&x3, &x3, &j); fp_sub(&x3, &x3, &v); fp_sub(&x3, &x3, &v);
  fp_sub(&t, &v, &x3); fp_mul(&y3, &rr, &t); fp_mul(&t, &s1, &j); fp_dbl(&t, &t); fp_sub(&y3, &y3, &t);
  fp_add(&z3, &p->z, &q->z); fp_sqr(&z3, &z3); fp_sub(&z3, &z3, &z1z1); fp_sub(&z3, &z3, &z2z2); fp_mul(&z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}
static void g1j_from_affine(g1j* r, const g1a* a) { if (a->inf) { g1j_set_inf(r); return; } r->x = a->x; r->y = a->y; r->z = FP_ONE; }
static void g1j_to_affine(g1a* r, const g1j* p) {
  if (g1j_is_inf(p)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
  fp zi, zi2; fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi);
  fp_mul(&r->x, &p->x, &zi2); fp_mul(&zi2, &zi2, &zi); fp_mul(&r->y, &p->y, &zi2); r->inf = 0;
}
/* blst_p1_mult (src/bls.rs:474-489) */
static void g1j_mul(g1j* r, const g1j* p, const fr* k_plain) {
  g1j acc; g1j_set_inf(&acc);
  for (int i = 254; i >= 0; i--) { g1j_double(&acc, &acc); if ((k_plain->l[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, p); }
  *r = acc;
}
/* XYZZ bucket += affine (complete) */
static void g1x_set_inf(g1x* p) { memset(p, 0, sizeof *p); }
static int g1x_is_inf(const g1x* p) { return fp_is_zero(&p->zz); }
static void g1x_dbl_affine(g1x* p, const fp* x, const fp* y) {
  fp u, v, w, s, m, t;
  fp_dbl(&u, y); fp_sqr(&v, &u); fp_mul(&w, &u, &v); fp_mul(&s, x, &v);
  fp_sqr(&m, x); fp_dbl(&t, &m); fp_add(&m, &m, &t);
  fp_sqr(&p->x, &m); fp_sub(&p->x, &p->x, &s); fp_sub(&p->x, &p->x, &s);
  fp_sub(&t, &s, &p->x); fp_mul(&t, &m, &t); fp_mul(&u, &w, y); fp_sub(&p->y, &t, &u);
  p->zz = v; p->zzz = w;
}
static void g1x_madd(g1x* p, const fp* x2, const fp* y2) {
  if (g1x_is_inf(p)) { p->x = *x2; p->y = *y2; p->zz = FP_ONE; p->zzz = FP_ONE; return; }
  fp u2, s2, pp, ppp, q, r, t;
  fp_mul(&u2, x2, &p->zz); fp_mul(&s2, y2, &p->zzz); fp_sub(&u2, &u2, &p->x); fp_sub(&r, &s2, &p->y);
  if (fp_is_zero(&u2)) { if (fp_is_zero(&r)) g1x_dbl_affine(p, x2, y2); else g1x_set_inf(p); return; }
  fp_sqr(&pp, &u2); fp_mul(&ppp, &u2, &pp); fp_mul(&q, &p->x, &pp);
  fp_sqr(&t, &r); fp_sub(&t, &t, &ppp); fp_sub(&t, &t, &q); fp_sub(&t, &t, &q);
  fp_sub(&q, &q, &t); fp_mul(&q, &r, &q); fp_mul(&s2, &p->y, &ppp); fp_sub(&p->y, &q, &s2); p->x = t;
  fp_mul(&p->zz, &p->zz, &pp); fp_mul(&p->zzz, &p->zzz, &ppp);
}
static void g1x_to_jac(g1j* r, const g1x* p) {
  /* x = X/ZZ, y = Y/ZZZ with ZZ = z^2, ZZZ = z^3  ==>  (X*ZZ, Y*ZZZ, ZZ) is Jacobian for Z = ZZ */
  if (g1x_is_inf(p)) { g1j_set_inf(r); return; }
  fp_mul(&r->x, &p->x, &p->zz);
  fp_mul(&r->y, &p->y, &p->zzz);
  r->z = p->zz;
}
static void g1x_add_xyzz(g1x* p, const g1x* q) { /* add-2008-s, complete */
  if (g1x_is_inf(q)) return;
  if (g1x_is_inf(p)) { *p = *q; return; }
  fp u1, u2, s1, s2, pp, ppp, qq, r, t;
  fp_mul(&u1, &p->x, &q->zz); fp_mul(&u2, &q->x, &p->zz); fp_mul(&s1, &p->y, &q->zzz); fp_mul(&s2, &q->y, &p->zzz);
  fp_sub(&u2, &u2, &u1); fp_sub(&r, &s2, &s1);
  if (fp_is_zero(&u2)) {
    if (fp_is_zero(&r)) { /* double p */
      fp u, v, w, s, m; fp_dbl(&u, &p->y); fp_sqr(&v, &u); fp_mul(&w, &u, &v); fp_mul(&s, &p->x, &v);
      fp_sqr(&m, &p->x); fp_dbl(&t, &m); fp_add(&m, &m, &t);
      fp x3, y3; fp_sqr(&x3, &m); fp_sub(&x3, &x3, &s); fp_sub(&x3, &x3, &s);
      fp_sub(&t, &s, &x3); fp_mul(&t, &m, &t); fp_mul(&u, &w, &p->y); fp_sub(&y3, &t, &u);
      fp_mul(&p->zz, &v, &p->zz); fp_mul(&p->zzz, &w, &p->zzz); p->x = x3; p->y = y3;
    } else g1x_set_inf(p);
    return;
  }
  fp_sqr(&pp, &u2); fp_mul(&ppp, &u2, &pp); fp_mul(&qq, &u1, &pp);
  fp_sqr(&t, &r); fp_sub(&t, &t, &ppp); fp_sub(&t, &t, &qq); fp_sub(&t, &t, &qq);
  fp_sub(&qq, &qq, &t); fp_mul(&qq, &r, &qq); fp_mul(&s1, &s1, &ppp); fp_sub(&p->y, &qq, &s1); p->x = t;
  fp_mul(&p->zz, &p->zz, &q->zz); fp_mul(&p->zz, &p->zz, &pp); fp_mul(&p->zzz, &p->zzz, &q->zzz); fp_mul(&p->zzz, &p->zzz, &ppp);
}

/* ZCash compressed encoding (src/bls.rs:491-531) */
static int fp_lex_larger(const fp* y_mont) {
  static const fp HALF = {{0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL, 0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL}};
  fp p, t; fp_to_plain(&p, y_mont);
  return fp_raw_sub(&t, &HALF, &p) != 0;
}
static void g1_compress(uint8_t* out, const g1a* a) {
  if (a->inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
  fp xp; fp_to_plain(&xp, &a->x); fp_to_be48(out, &xp);
  out[0] |= 0x80; if (fp_lex_larger(&a->y)) out[0] |= 0x20;
}
/* returns 0 ok, 3 InvalidEncoding, 4 NotOnCurve (blst_p1_uncompress, src/bls.rs:514-521) */
static int g1_uncompress(g1a* r, const uint8_t* in) {
  uint8_t b0 = in[0];
  if (!(b0 & 0x80)) return 3;
  if (b0 & 0x40) { uint8_t o = b0 & 0x3f; for (int i = 1; i < 48; i++) o |= in[i]; if (o) return 3; memset(r, 0, sizeof *r); r->inf = 1; return 0; }
  uint8_t tmp[48]; memcpy(tmp, in, 48); tmp[0] &= 0x1f;
  fp xp, t; fp_from_be48(&xp, tmp);
  if (!fp_raw_sub(&t, &xp, &FP_P)) return 3;
  fp_from_plain(&r->x, &xp);
  fp rhs; fp_sqr(&t, &r->x); fp_mul(&rhs, &t, &r->x); fp_add(&rhs, &rhs, &FP_B);
  /* sqrt: rhs^((p+1)/4) */
  static const u64 E[NP] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL, 0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
  fp_pow(&r->y, &rhs, E, NP);
  fp_sqr(&t, &r->y);
  if (!fp_eq(&t, &rhs)) return 4;
  if (((b0 & 0x20) != 0) != (fp_lex_larger(&r->y) != 0)) fp_neg(&r->y, &r->y);
  r->inf = 0;
  return 0;
}
static int g1_in_subgroup(const g1a* a) { /* [r]P == O (blst_p1_affine_in_g1, src/bls.rs:522) */
  if (a->inf) return 1;
  g1j p, acc; g1j_from_affine(&p, a); g1j_set_inf(&acc);
  for (int i = 254; i >= 0; i--) { g1j_double(&acc, &acc); if ((FR_R.l[i >> 6] >> (i & 63)) & 1) g1j_add(&acc, &acc, &p); }
  return g1j_is_inf(&acc);
}

/* --------------------------------------------------------------- setup --- */
typedef struct {
  g1j g1_lagrange_brp[4096];   /* Box<[P1; 4096]> Jacobian, as the reference stores it (src/kzg/setup.rs:39) */
  fr roots_brp[4096];          /* Montgomery */
  int threads;
} cport_setup;

static void init_consts(void) {
  static int done = 0; if (done) return; done = 1;
  /* R mod p, R^2 mod p by repeated doubling */
  fp one = {{1, 0, 0, 0, 0, 0}}; fp x = one;
  for (int i = 0; i < 384; i++) fp_add(&x, &x, &x);
  FP_ONE = x;
  for (int i = 0; i < 384; i++) fp_add(&x, &x, &x);
  FP_R2 = x;
  fp four = {{4, 0, 0, 0, 0, 0}}; fp_from_plain(&FP_B, &four);
  fr ro = {{1, 0, 0, 0}}; fr y = ro;
  for (int i = 0; i < 256; i++) fr_add(&y, &y, &y);
  FR_ONE = y;
  for (int i = 0; i < 256; i++) fr_add(&y, &y, &y);
  FR_R2 = y;
}
static unsigned bitrev12(unsigned i) { unsigned r = 0; for (int b = 0; b < 12; b++) r |= ((i >> b) & 1u) << (11 - b); return r; }

/* Setup::load_json after JSON parsing (src/kzg/setup.rs:52-81).  Returns 0 or (index+1)*16 + code. */
int cport_setup_create(cport_setup** out, const uint8_t* g1_lagrange48, int subgroup_checks, int threads) {
  init_consts();
  cport_setup* s = (cport_setup*)calloc(1, sizeof *s);
  s->threads = threads > 0 ? threads : 1;
  for (unsigned i = 0; i < 4096; i++) {
    g1a a; int st = g1_uncompress(&a, g1_lagrange48 + 48 * i);
    if (!st && subgroup_checks && !g1_in_subgroup(&a)) st = 5;
    if (st) { free(s); return (int)((i + 1) * 16 + st); }
    g1j_from_affine(&s->g1_lagrange_brp[bitrev12(i)], &a);
  }
  /* roots_of_unity (src/math.rs:16-29), generator 7 */
  fr seven = {{7, 0, 0, 0}}, g; fr_from_plain(&g, &seven);
  /* exponent (r-1)/4096 */
  fr e = FR_R; e.l[0] -= 1;
  for (int k = 0; k < 12; k++) { for (int i = 0; i < NR - 1; i++) e.l[i] = (e.l[i] >> 1) | (e.l[i + 1] << 63); e.l[NR - 1] >>= 1; }
  fr w = FR_ONE;
  for (int i = 255; i >= 0; i--) { fr_mul(&w, &w, &w); if ((e.l[i >> 6] >> (i & 63)) & 1) fr_mul(&w, &w, &g); }
  fr cur = FR_ONE;
  for (unsigned i = 0; i < 4096; i++) { s->roots_brp[bitrev12(i)] = cur; fr_mul(&cur, &cur, &w); }
  *out = s;
  return 0;
}
void cport_setup_destroy(cport_setup* s) { free(s); }

/* ----------------------------------------------------- Pippenger MSM ----- */
#define WBITS 10
#define NWIN 26 /* ceil(255/10) = 26 */
typedef struct { const g1a* pts; const uint8_t* scalars_le; /* n*32 */ int n; int win; int lo, hi; g1x result; } tile;

/* Booth-style signed digit of window `win`: value in [-512, 512] */
static int booth_digit(const uint8_t* s_le, int win) {
  int bit = win * WBITS - 1; /* include the bit below the window */
  unsigned v = 0;
  for (int k = 0; k <= WBITS; k++) { int b = bit + k; unsigned x = 0; if (b >= 0 && b < 256) x = (s_le[b >> 3] >> (b & 7)) & 1u; v |= x << k; }
  /* v has WBITS+1 bits: booth recode: digit = ((v + 1) >> 1) - (top ? 2^WBITS : 0) */
  int top = (v >> WBITS) & 1;
  int d = (int)((v + 1) >> 1);
  if (top) d -= (1 << WBITS);
  return d;
}
static void tile_run(tile* t) {
  g1x* buckets = (g1x*)calloc(1 << (WBITS - 1), sizeof(g1x));
  for (int i = t->lo; i < t->hi; i++) {
    int d = booth_digit(t->scalars_le + 32 * i, t->win);
    if (d == 0 || t->pts[i].inf) continue;
    if (d > 0) g1x_madd(&buckets[d - 1], &t->pts[i].x, &t->pts[i].y);
    else { fp ny; fp_neg(&ny, &t->pts[i].y); g1x_madd(&buckets[-d - 1], &t->pts[i].x, &ny); }
  }
  g1x run, acc; g1x_set_inf(&run); g1x_set_inf(&acc);
  for (int b = (1 << (WBITS - 1)) - 1; b >= 0; b--) { g1x_add_xyzz(&run, &buckets[b]); g1x_add_xyzz(&acc, &run); }
  t->result = acc;
  free(buckets);
}
typedef struct { tile* tiles; int ntiles; volatile int* next; } worker_arg;
static void* worker(void* p) {
  worker_arg* a = (worker_arg*)p;
  for (;;) { int k = __sync_fetch_and_add(a->next, 1); if (k >= a->ntiles) break; tile_run(&a->tiles[k]); }
  return NULL;
}
/* P1::lincomb_pippenger (src/bls.rs:416-437) */
static void lincomb_pippenger(g1j* out, const g1j* points, const fr* scalars_mont, int n, int threads) {
  /* p1_affines::from: batch to-affine of all bases on every call (src/bls.rs:426) */
  g1a* aff = (g1a*)malloc(sizeof(g1a) * n);
  {
    fp* pre = (fp*)malloc(sizeof(fp) * n); fp acc = FP_ONE;
    for (int i = 0; i < n; i++) { pre[i] = acc; if (!g1j_is_inf(&points[i])) fp_mul(&acc, &acc, &points[i].z); }
    fp inv; fp_inv(&inv, &acc);
    for (int i = n - 1; i >= 0; i--) {
      if (g1j_is_inf(&points[i])) { memset(&aff[i], 0, sizeof(g1a)); aff[i].inf = 1; continue; }
      fp zi, zi2; fp_mul(&zi, &inv, &pre[i]); fp_mul(&inv, &inv, &points[i].z);
      fp_sqr(&zi2, &zi); fp_mul(&aff[i].x, &points[i].x, &zi2); fp_mul(&zi2, &zi2, &zi); fp_mul(&aff[i].y, &points[i].y, &zi2); aff[i].inf = 0;
    }
    free(pre);
  }
  /* Fr::to_le_bytes for every scalar (src/bls.rs:428-432) */
  uint8_t* sc = (uint8_t*)malloc(32 * (size_t)n);
  for (int i = 0; i < n; i++) { fr p; fr_to_plain(&p, &scalars_mont[i]); for (int k = 0; k < 32; k++) sc[32 * i + k] = (uint8_t)(p.l[k >> 3] >> (8 * (k & 7))); }
  /* tiles: NWIN windows x slices */
  int slices = 1;
  if (threads > 1) { while (NWIN * slices < 4 * threads && n / (slices * 2) >= 256) slices *= 2; }
  int ntiles = NWIN * slices;
  tile* tiles = (tile*)calloc(ntiles, sizeof(tile));
  for (int w = 0; w < NWIN; w++) for (int s = 0; s < slices; s++) {
    tile* t = &tiles[w * slices + s]; t->pts = aff; t->scalars_le = sc; t->n = n; t->win = w; t->lo = (int)((long)n * s / slices); t->hi = (int)((long)n * (s + 1) / slices);
  }
  volatile int next = 0;
  worker_arg wa = {tiles, ntiles, &next};
  if (threads <= 1) worker(&wa);
  else {
    pthread_t th[256]; int nt = threads > 256 ? 256 : threads;
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, worker, &wa);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
  }
  /* combine: Horner over windows */
  g1x total; g1x_set_inf(&total);
  g1j tj; g1j_set_inf(&tj);
  for (int w = NWIN - 1; w >= 0; w--) {
    for (int k = 0; k < WBITS; k++) g1j_double(&tj, &tj);
    g1x wsum; g1x_set_inf(&wsum);
    for (int s = 0; s < slices; s++) g1x_add_xyzz(&wsum, &tiles[w * slices + s].result);
    g1j wj; g1x_to_jac(&wj, &wsum);
    g1j_add(&tj, &tj, &wj);
  }
  (void)total;
  *out = tj;
  free(tiles); free(sc); free(aff);
}

/* ------------------------------------------------------------ blob API --- */
/* Blob::from_slice (src/blob.rs:26-37): 0 ok, 2 InvalidFieldElement */
static int blob_from_slice(fr* elements, const uint8_t* blob) {
  for (int i = 0; i < 4096; i++) if (fr_from_be_checked(&elements[i], blob + 32 * i)) return 2;
  return 0;
}
/* Setup::blob_to_commitment + compress (src/kzg/setup.rs:167-171, benches/kzg.rs:24-26) */
int cport_blob_to_commitment(const cport_setup* s, const uint8_t* blob, uint8_t* out48) {
  fr* el = (fr*)malloc(sizeof(fr) * 4096);
  int st = blob_from_slice(el, blob);
  if (st) { free(el); memset(out48, 0, 48); return st; }
  g1j c; lincomb_pippenger(&c, s->g1_lagrange_brp, el, 4096, s->threads);
  g1a a; g1j_to_affine(&a, &c); g1_compress(out48, &a);
  free(el);
  return 0;
}
/* times `reps` passes over n blobs; returns seconds.  compress == 0 mirrors the
   reference bench's timed region exactly (benches/kzg.rs:36 returns the P1). */
double cport_time_commitments(const cport_setup* s, const uint8_t* blobs, int n, int reps, int compress, uint8_t* out48) {
  struct timespec t0, t1;
  fr* el = (fr*)malloc(sizeof(fr) * 4096);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < reps; r++) for (int b = 0; b < n; b++) {
    blob_from_slice(el, blobs + (size_t)b * 131072);
    g1j c; lincomb_pippenger(&c, s->g1_lagrange_brp, el, 4096, s->threads);
    if (compress || r == reps - 1) { g1a a; g1j_to_affine(&a, &c); g1_compress(out48 + 48 * b, &a); }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(el);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
/* blob-parallel variant: `threads` workers, each commits whole blobs with a single-threaded MSM
   (the embarrassingly parallel way to use many cores on independent blobs; the reference itself
   loops over blobs sequentially and threads only inside blst's MSM). */
typedef struct { const cport_setup* s; const uint8_t* blobs; int n; int reps; uint8_t* out48; volatile int* next; } par_arg;
static void* par_worker(void* p) {
  par_arg* a = (par_arg*)p;
  fr* el = (fr*)malloc(sizeof(fr) * 4096);
  for (;;) {
    int k = __sync_fetch_and_add(a->next, 1);
    if (k >= a->n * a->reps) break;
    int b = k % a->n;
    blob_from_slice(el, a->blobs + (size_t)b * 131072);
    g1j c; lincomb_pippenger(&c, a->s->g1_lagrange_brp, el, 4096, 1);
    g1a af; g1j_to_affine(&af, &c); g1_compress(a->out48 + 48 * b, &af);
  }
  free(el);
  return NULL;
}
double cport_time_commitments_blob_parallel(const cport_setup* s, const uint8_t* blobs, int n, int reps, int threads, uint8_t* out48) {
  struct timespec t0, t1;
  volatile int next = 0;
  par_arg a = {s, blobs, n, reps, out48, &next};
  pthread_t th[256]; int nt = threads > 256 ? 256 : (threads < 1 ? 1 : threads);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, par_worker, &a);
  for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
/* ------------------------------------------------------------- SHA-256 (blst_sha256, src/bls.rs:194) --- */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_blocks(uint32_t h[8], const uint8_t* p, size_t nblocks) {
  for (size_t b = 0; b < nblocks; b++, p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) { uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10); w[i] = w[i - 16] + s0 + w[i - 7] + s1; }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
      uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
}
static void sha256(uint8_t out[32], const uint8_t* msg, size_t len) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t full = len / 64; sha256_blocks(h, msg, full);
  uint8_t tail[128] = {0}; size_t rem = len - 64 * full; memcpy(tail, msg + 64 * full, rem); tail[rem] = 0x80;
  size_t nb = (rem + 9 <= 64) ? 1 : 2; u64 bits = (u64)len * 8;
  for (int i = 0; i < 8; i++) tail[nb * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_blocks(h, tail, nb);
  for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}
/* Fr::hash_to (src/bls.rs:189-205): digest as big-endian integer, reduced mod r, to Montgomery */
static void fr_hash_to(fr* out_mont, const uint8_t* msg, size_t len) {
  uint8_t d[32]; sha256(d, msg, len);
  fr v, t; fr_from_be32(&v, d);
  if (!fr_raw_sub(&t, &v, &FR_R)) v = t;
  if (!fr_raw_sub(&t, &v, &FR_R)) v = t;
  fr_from_plain(out_mont, &v);
}
/* Blob::challenge (src/blob.rs:78-97): re-serialises every element (Fr::to_be_bytes) like the reference */
static void blob_challenge(fr* z_mont, const fr* elements_mont, const uint8_t* commitment48) {
  uint8_t* data = (uint8_t*)malloc(32 + 131072 + 48);
  memcpy(data, "FSBLOBVERIFY_V1_", 16); memset(data + 16, 0, 16); data[30] = 0x10;
  for (int i = 0; i < 4096; i++) { fr p; fr_to_plain(&p, &elements_mont[i]); fr_to_be32(data + 32 + 32 * i, &p); }
  memcpy(data + 32 + 131072, commitment48, 48);
  fr_hash_to(z_mont, data, 32 + 131072 + 48);
  free(data);
}
/* Polynomial::evaluate (src/kzg/poly.rs:10-33): one division (= one Euclidean inversion) per element */
static void poly_evaluate(fr* y_mont, const fr* e_mont, const fr* z_mont, const cport_setup* s, int batch_inverse) {
  for (int i = 0; i < 4096; i++) if (fr_eq(z_mont, &s->roots_brp[i])) { *y_mont = e_mont[i]; return; }
  fr acc; memset(&acc, 0, sizeof acc);
  if (!batch_inverse) {
    for (int i = 0; i < 4096; i++) {
      fr num, den, inv, term; fr_mul(&num, &e_mont[i], &s->roots_brp[i]); fr_sub(&den, z_mont, &s->roots_brp[i]);
      fr_eucl_inverse(&inv, &den); fr_mul(&term, &num, &inv); fr_add(&acc, &acc, &term);
    }
  } else { /* labelled variant: Montgomery batch inversion (not what the reference does) */
    fr* pre = (fr*)malloc(sizeof(fr) * 4096); fr run = FR_ONE;
    for (int i = 0; i < 4096; i++) { fr den; fr_sub(&den, z_mont, &s->roots_brp[i]); pre[i] = run; fr_mul(&run, &run, &den); }
    fr inv; fr_eucl_inverse(&inv, &run);
    for (int i = 4095; i >= 0; i--) {
      fr den, di, num, term; fr_sub(&den, z_mont, &s->roots_brp[i]); fr_mul(&di, &inv, &pre[i]); fr_mul(&inv, &inv, &den);
      fr_mul(&num, &e_mont[i], &s->roots_brp[i]); fr_mul(&term, &num, &di); fr_add(&acc, &acc, &term);
    }
    free(pre);
  }
  fr zn = *z_mont; for (int k = 0; k < 12; k++) fr_mul(&zn, &zn, &zn);
  fr_sub(&zn, &zn, &FR_ONE);
  fr n4096 = {{4096, 0, 0, 0}}, nm, ninv; fr_from_plain(&nm, &n4096); fr_eucl_inverse(&ninv, &nm);
  fr_mul(&zn, &zn, &ninv); fr_mul(y_mont, &acc, &zn);
}
/* Setup::verify_blob_proof_batch up to (not including) the pairing: src/kzg/setup.rs:223-245 and :115-156.
   Follows the reference literally: r from (domain, 4096, n) only (quirk Q1), rpowers via Fr::pow with the
   pow(x,0)=x quirk (Q2), three NAIVE lincombs (one 255-bit scalar multiplication per term) and n more for
   [-y_i]G.  Outputs: z (n x 32 B BE), y (n x 32 B BE), A = proof_lincomb (48 B), B = c_minus_y + proof_z (48 B).
   Returns 0 or the first error code in the reference's order. */
static void fr_pow_reference(fr* out, const fr* x, u64 power) { /* src/bls.rs:169-187 */
  fr o = *x, tmp = FR_ONE;
  while (power != 1 && power != 0) { if (power & 1) { fr_mul(&tmp, &o, &tmp); power -= 1; } fr_mul(&o, &o, &o); power >>= 1; }
  fr_mul(out, &o, &tmp);
}
int cport_verify_batch_prepairing(const cport_setup* s, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, int n, int batch_inverse,
                                  uint8_t* z_out, uint8_t* y_out, uint8_t* a48, uint8_t* b48) {
  fr* el = (fr*)malloc(sizeof(fr) * 4096 * (size_t)n);
  g1a* cs = (g1a*)malloc(sizeof(g1a) * n); g1a* ps = (g1a*)malloc(sizeof(g1a) * n);
  fr* zs = (fr*)malloc(sizeof(fr) * n); fr* ys = (fr*)malloc(sizeof(fr) * n);
  int rc = 0;
  for (int i = 0; i < n && !rc; i++) rc = blob_from_slice(el + 4096 * (size_t)i, blobs + (size_t)i * 131072);
  for (int i = 0; i < n && !rc; i++) { rc = g1_uncompress(&cs[i], commitments48 + 48 * i); if (!rc && !g1_in_subgroup(&cs[i])) rc = 5; }
  for (int i = 0; i < n && !rc; i++) { rc = g1_uncompress(&ps[i], proofs48 + 48 * i); if (!rc && !g1_in_subgroup(&ps[i])) rc = 5; }
  if (!rc) {
    for (int i = 0; i < n; i++) {  /* src/kzg/setup.rs:235-242 (sequential) */
      blob_challenge(&zs[i], el + 4096 * (size_t)i, commitments48 + 48 * i);
      poly_evaluate(&ys[i], el + 4096 * (size_t)i, &zs[i], s, batch_inverse);
      fr p; fr_to_plain(&p, &zs[i]); fr_to_be32(z_out + 32 * i, &p); fr_to_plain(&p, &ys[i]); fr_to_be32(y_out + 32 * i, &p);
    }
    uint8_t data[48]; memcpy(data, "RCKZGBATCH___V1_", 16); memset(data + 16, 0, 32); data[30] = 0x10;
    for (int k = 0; k < 8; k++) data[47 - k] = (uint8_t)((u64)n >> (8 * k));
    fr r; fr_hash_to(&r, data, 48);
    /* generator */
    g1a gen; { static const uint8_t G48[48] = {0x97,0xf1,0xd3,0xa7,0x31,0x97,0xd7,0x94,0x26,0x95,0x63,0x8c,0x4f,0xa9,0xac,0x0f,0xc3,0x68,0x8c,0x4f,0x97,0x74,0xb9,0x05,0xa1,0x4e,0x3a,0x3f,0x17,0x1b,0xac,0x58,0x6c,0x55,0xe8,0x3f,0xf9,0x7a,0x1a,0xef,0xfb,0x3a,0xf0,0x0a,0xdb,0x22,0xc6,0xbb}; g1_uncompress(&gen, G48); }
    g1j negG; { g1a ng = gen; fp_neg(&ng.y, &ng.y); g1j_from_affine(&negG, &ng); }
    g1j A, Bz, Bc; g1j_set_inf(&A); g1j_set_inf(&Bz); g1j_set_inf(&Bc);
    for (int i = 0; i < n; i++) {
      fr rp, zr, rpp, zrp, yp; fr_pow_reference(&rp, &r, (u64)i); fr_mul(&zr, &zs[i], &rp);
      fr_to_plain(&rpp, &rp); fr_to_plain(&zrp, &zr); fr_to_plain(&yp, &ys[i]);
      g1j pj, cj, t; g1j_from_affine(&pj, &ps[i]); g1j_from_affine(&cj, &cs[i]);
      g1j_mul(&t, &pj, &rpp); g1j_add(&A, &A, &t);           /* proof_lincomb        :152 */
      g1j_mul(&t, &pj, &zrp); g1j_add(&Bz, &Bz, &t);         /* proof_z_lincomb      :153 */
      g1j cmy; g1j_mul(&t, &negG, &yp); g1j_add(&cmy, &cj, &t); /* C_i + [-y_i]G     :149 */
      g1j_mul(&t, &cmy, &rpp); g1j_add(&Bc, &Bc, &t);        /* comm_minus_eval_lincomb :155 */
    }
    g1j B; g1j_add(&B, &Bc, &Bz);
    g1a aa, ba; g1j_to_affine(&aa, &A); g1j_to_affine(&ba, &B); g1_compress(a48, &aa); g1_compress(b48, &ba);
  }
  free(el); free(cs); free(ps); free(zs); free(ys);
  return rc;
}
double cport_time_verify_prepairing(const cport_setup* s, const uint8_t* blobs, const uint8_t* c48, const uint8_t* p48, int n, int batch_inverse) {
  struct timespec t0, t1; uint8_t a[48], b[48]; uint8_t* z = (uint8_t*)malloc(32 * (size_t)n); uint8_t* y = (uint8_t*)malloc(32 * (size_t)n);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  cport_verify_batch_prepairing(s, blobs, c48, p48, n, batch_inverse, z, y, a, b);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(z); free(y);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
void cport_fr_eucl_inv_plain(uint8_t* out32be, const uint8_t* a32be) {
  init_consts(); fr a, r; fr_from_be32(&a, a32be); fr_from_plain(&a, &a); fr_eucl_inverse(&r, &a); fr_to_plain(&r, &r); fr_to_be32(out32be, &r);
}
void cport_sha256(uint8_t* out32, const uint8_t* msg, size_t len) { sha256(out32, msg, len); }

void cport_set_threads(cport_setup* s, int threads) { s->threads = threads > 0 ? threads : 1; }

/* small exported helpers for tests/test_cport.py */
int cport_g1_decompress_status(const uint8_t* in48) { init_consts(); g1a a; int st = g1_uncompress(&a, in48); if (!st && !g1_in_subgroup(&a)) st = 5; return st; }
void cport_fp_mul_plain(uint8_t* out48be, const uint8_t* a48be, const uint8_t* b48be) {
  init_consts(); fp a, b, r; fp_from_be48(&a, a48be); fp_from_be48(&b, b48be); fp_from_plain(&a, &a); fp_from_plain(&b, &b); fp_mul(&r, &a, &b); fp_to_plain(&r, &r); fp_to_be48(out48be, &r);
}
void cport_fr_mul_plain(uint8_t* out32be, const uint8_t* a32be, const uint8_t* b32be) {
  init_consts(); fr a, b, r; fr_from_be32(&a, a32be); fr_from_be32(&b, b32be); fr_from_plain(&a, &a); fr_from_plain(&b, &b); fr_mul(&r, &a, &b); fr_to_plain(&r, &r); fr_to_be32(out32be, &r);
}
void cport_fr_inv_plain(uint8_t* out32be, const uint8_t* a32be) {
  init_consts(); fr a, r; fr_from_be32(&a, a32be); fr_from_plain(&a, &a); fr_inv(&r, &a); fr_to_plain(&r, &r); fr_to_be32(out32be, &r);
}
int cport_g1_mul(uint8_t* out48, const uint8_t* in48, const uint8_t* k32be) {
  init_consts(); g1a a; int st = g1_uncompress(&a, in48); if (st) return st;
  fr k; fr_from_be32(&k, k32be); g1j p, r; g1j_from_affine(&p, &a); g1j_mul(&r, &p, &k); g1a o; g1j_to_affine(&o, &r); g1_compress(out48, &o); return 0;
}
