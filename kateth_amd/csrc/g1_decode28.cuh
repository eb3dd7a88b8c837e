// P1::decompress (blst_p1_uncompress + blst_p1_affine_in_g1, src/bls.rs:505-531) with the field work in the
// carry-free radix-2^28 representation (fp28.cuh).  Same checks, same error codes and the same output format
// (canonical 2^384-Montgomery x, y in 12 x 32-bit limbs) as g1_decompress of g1.cuh, which stays as the reference
// implementation that tests/test_hostmath.py compares this one against.
//
// Cost per point: square root a^((p+1)/4) by a width-3 sliding window (378 squarings + 108 products; a squaring is
// 105 + 196 v_mad_u64_u32), subgroup test phi(P) == -[z^2]P by a Jacobian ladder over z^2 (127 doublings of 4 S + 3 M and
// 16 mixed additions of the affine input) -- about 0.65 M VALU instructions against 1.3 M for the 12 x 32-bit-limb path.
//
// Everything on the decoding path is INLINED into its kernel (round 4): out of line, the square root, the ladder and its
// adder took their operands by address -- i.e. through scratch -- and saved their callers' registers: 247 VGPRs + 720 B of
// scratch per lane, 2.76 GB of scratch write-through per 131,072 points.  Inlined, hipcc keeps the two loop bodies (one
// squaring + one product; one doubling + one mixed addition) in 224 VGPRs with no scratch at all, which also leaves a SIMD
// holding two decoder waves 64 registers for a third, small wave (the transcript / scalar / sorting kernels of batch
// verification run in the decoder's shadow: engine_verify.hip).
#pragma once
#include "fp28.cuh"
#include "issue_fair.cuh"

namespace kzg {

#define KZG_F28_TABLE(fn, MACRO)         \
  KZG_HD constexpr uint32_t fn(int i) {  \
    constexpr uint32_t t[F28_N] = MACRO; \
    return t[i];                         \
  }
KZG_F28_TABLE(f28_r2_limb, KZG_FP28_R2)
KZG_F28_TABLE(f28_b_limb, KZG_FP28_B)
KZG_F28_TABLE(f28_beta_limb, KZG_FP28_BETA)
KZG_F28_TABLE(f28_r400_limb, KZG_FP28_R400)
KZG_F28_TABLE(f28_24p_t8, KZG_FP28_24P_T8)
KZG_F28_TABLE(f28_32p_t1, KZG_FP28_32P_T1)
#undef KZG_F28_TABLE


// r = a^e for a fixed exponent given as a width-3 sliding-window schedule (tools/gen_consts.py): `first` is the
// leading digit's index, then (squarings, digit index | 255) pairs; digit index d stands for a^(2d+1).
// a: N-form.  Result N-form.
KZG_HD void f28_pow_sched(fp28& r, const fp28& a, const uint8_t* sched, int len, int first) {
  fp28 t1 = a, t3, t5, t7, a2;
  f28_sqr(a2, a);
  f28_mul(t3, a2, a);
  f28_mul(t5, t3, a2);
  f28_mul(t7, t5, a2);
  fp28 acc;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) acc.l[i] = first == 0 ? t1.l[i] : (first == 1 ? t3.l[i] : (first == 2 ? t5.l[i] : t7.l[i]));
#pragma unroll 1
  for (int s = 0; s < len; s++) {
    issue_fair_tick(18);  // one thread per point, two waves per SIMD: see issue_fair.cuh
    const int nsq = sched[2 * s], idx = sched[2 * s + 1];
#pragma unroll 1
    for (int q = 0; q < nsq; q++) f28_sqr(acc, acc);
    if (idx != 255) {
      fp28 m;
      KZG_UNROLL_FULL
      for (int i = 0; i < F28_N; i++) m.l[i] = idx == 0 ? t1.l[i] : (idx == 1 ? t3.l[i] : (idx == 2 ? t5.l[i] : t7.l[i]));
      f28_mul(acc, acc, m);
    }
  }
  r = acc;
}
// a^((p+1)/4): the square root of a when a is a square (378 squarings + 105 products)
KZG_HD void f28_sqrt_candidate(fp28& r, const fp28& a) {
  const uint8_t sched[2 * KZG_FP_SQRT_SCHED_LEN] = KZG_FP_SQRT_SCHED;
  f28_pow_sched(r, a, sched, KZG_FP_SQRT_SCHED_LEN, KZG_FP_SQRT_FIRST_DIGIT_INDEX);
}
// a^(p-2) = 1/a (0 -> 0) by the sliding-window power: kept as the cross-check of f28_inv (tests/test_hostmath.py)
KZG_HD_NOINLINE void f28_inv_fermat(fp28& r, const fp28& a) {
  const uint8_t sched[2 * KZG_FP_INV_SCHED_LEN] = KZG_FP_INV_SCHED;
  f28_pow_sched(r, a, sched, KZG_FP_INV_SCHED_LEN, KZG_FP_INV_FIRST_DIGIT_INDEX);
}
// 1/a in the 2^392-Montgomery domain (a: N-form; 0 -> 0): safegcd on the canonical residue (modinv30.cuh), then one
// product by R'^3:  (a R')^-1 R'^3 / R' = a^-1 R'.  ~20 k instead of ~200 k VALU instructions.
// FERMAT_INL: the fallback power (never taken: 45 batches exceed the proven bound of 37) is inlined as well, so that a kernel
// whose only out-of-line callee it would be needs no scratch memory (the encoder at the end of every single-item call).
template <bool FERMAT_INL = false>
KZG_HD void f28_inv(fp28& r, const fp28& a) {
  fp_t c, ci;
  f28_to_bn(c, a);
  canonicalize<FpParams>(c);
  bool ok;
  if constexpr (FERMAT_INL)
    ok = modinv30_inl<FpInv30>(ci, c);
  else
    ok = modinv30<FpInv30>(ci, c);
  if (!ok) {
    if constexpr (FERMAT_INL) {
      const uint8_t sched[2 * KZG_FP_INV_SCHED_LEN] = KZG_FP_INV_SCHED;
      f28_pow_sched(r, a, sched, KZG_FP_INV_SCHED_LEN, KZG_FP_INV_FIRST_DIGIT_INDEX);
    } else {
      f28_inv_fermat(r, a);
    }
    return;
  }
  fp28 t, k;
  f28_from_bn(t, ci);
  constexpr uint32_t r3[F28_N] = KZG_FP28_R3;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) k.l[i] = r3[i];
  f28_mul(r, t, k);
}

// blst_p1_compress (src/bls.rs:499) of an XYZZ point given in the 12 x 32-bit-limb format (canonical 2^384-Montgomery), as
// twelve 32-bit WORDS holding the 48 output bytes in memory order (word q = bytes 4q .. 4q + 3, i.e. the big-endian limbs
// byte-swapped for a little-endian store): same bytes as g1_compress_xyzz (g1.cuh); the inversion and the five products around it
// run in the radix-2^28 field.  Everything inline and no byte array: the encoder kernels keep the point, the intermediate values
// and the output in registers (round 5: out of line and through a 48-byte stack buffer they took 592 B of scratch per lane).
// `out_affine24` (optional): the same point as blst_p1_affine -- x || y, each 12 x 32-bit little-endian limbs of the
// canonical 2^384-Montgomery residue (the byte image of blst's 6 x 64-bit limbs); infinity = all zero.
template <bool WANT_AFFINE>
KZG_HD void g1_encode_xyzz28_words(uint32_t* out12, uint32_t* out_affine24, const g1_xyzz& p) {
  if (xyzz_is_inf(p)) {
    out12[0] = 0xC0u;  // byte 0 = 0xC0, the rest zero
    KZG_UNROLL_FULL
    for (int i = 1; i < 12; i++) out12[i] = 0;
    if (WANT_AFFINE) {
      KZG_UNROLL_FULL
      for (int i = 0; i < 24; i++) out_affine24[i] = 0;
    }
    return;
  }
  fp28 k, X, Y, ZZ, ZZZ, t, ti, a;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) k.l[i] = f28_r400_limb(i);
  // v * 2^384 read as an integer, times 2^400 / 2^392  ->  v * 2^392
  f28_from_bn(ZZ, p.zz);
  f28_mul(ZZ, ZZ, k);
  f28_from_bn(ZZZ, p.zzz);
  f28_mul(ZZZ, ZZZ, k);
  f28_mul(t, ZZ, ZZZ);
  f28_inv<true>(ti, t);
  f28_from_bn(X, p.x);
  f28_mul(X, X, k);
  f28_mul(a, ti, ZZZ);  // 1 / ZZ
  f28_mul(X, X, a);
  f28_from_bn(Y, p.y);
  f28_mul(Y, Y, k);
  f28_mul(a, ti, ZZ);   // 1 / ZZZ
  f28_mul(Y, Y, a);
  if (WANT_AFFINE) {
    fp_t xm, ym;
    f28_to_fp(xm, X);
    f28_to_fp(ym, Y);
    KZG_UNROLL_FULL
    for (int i = 0; i < 12; i++) {
      out_affine24[i] = xm.v[i];
      out_affine24[12 + i] = ym.v[i];
    }
  }
  fp28 one_plain;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) one_plain.l[i] = i == 0 ? 1u : 0u;
  f28_mul(X, X, one_plain);  // plain, N-form
  f28_mul(Y, Y, one_plain);
  fp_t xp, yp;
  f28_to_bn(xp, X);
  canonicalize<FpParams>(xp);
  f28_to_bn(yp, Y);
  canonicalize<FpParams>(yp);
  uint32_t top = xp.v[11] | 0x80000000u;                 // compressed flag (bit 7 of byte 0 = bit 31 of the top limb)
  if (fp_is_lex_larger_plain(yp)) top |= 0x20000000u;    // sign flag
  out12[0] = __builtin_bswap32(top);
  KZG_UNROLL_FULL
  for (int i = 1; i < 12; i++) out12[i] = __builtin_bswap32(xp.v[11 - i]);
}
// the same as 48 bytes (host tests, callers that want bytes)
KZG_HD_NOINLINE void g1_compress_xyzz28(uint8_t* out48, uint32_t* out_affine24, const g1_xyzz& p) {
  uint32_t w[12], aff[24];
  g1_encode_xyzz28_words<true>(w, aff, p);
  for (int q = 0; q < 12; q++) {
    out48[4 * q] = (uint8_t)w[q];
    out48[4 * q + 1] = (uint8_t)(w[q] >> 8);
    out48[4 * q + 2] = (uint8_t)(w[q] >> 16);
    out48[4 * q + 3] = (uint8_t)(w[q] >> 24);
  }
  if (out_affine24)
    for (int q = 0; q < 24; q++) out_affine24[q] = aff[q];
}

// ---- Jacobian ladder for the subgroup test ----------------------------------------------------------------------------
// 126 of the ~600 k v_mad_u64_u32 of a point decoding were XYZZ doublings (3 S + 4 M + a double product = 3,059 mads
// each); in Jacobian coordinates a doubling on y^2 = x^3 + 4 is 4 S + 3 M = 2,380 (dbl-2009-l with D = 4 X1 B as a
// product), needs no special case (the group order is odd: no point has Y = 0), and the ladder adds the AFFINE input
// point, so no general addition is needed: [z^2]P by one 127-step ladder over z^2 (Hamming weight 17).
// Invariant of the accumulator:  x: limbs <= 2^28 + 16, value < 26p;  y: limbs <= 2^28 + 16, value < 30p;
// z: limbs < 2^29, value < 4p.  Every bound below is re-checked at run time in the CPU test build (KZG_FP28_CHECK).
struct g1_jac28 {
  fp28 x, y, z;
  uint32_t inf;
};

// p = 2p   (A = X^2, B = Y^2, U = X B, C = B^2, E = 3A, F = E^2, X3 = F - 8U, Y3 = 3 A (4U - X3) - 8C, Z3 = 2 Y Z)
KZG_HD void jac28_dbl(g1_jac28& p) {
  if (p.inf) return;
  fp28 A, B, C, U, E, T;
  f28_sqr(A, p.x);        // 14 * (2^28+16)^2 ; 26^2 = 676 < 2^11
  f28_sqr(B, p.y);        // 30^2 = 900
  f28_mul(U, p.x, B);     // 26 * 2
  f28_sqr(C, B);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) E.l[i] = 3u * A.l[i];  // limbs < 3*2^28, value < 6p
  f28_sqr(E, E);          // F: 14 * 9 * 2^56 ; 36
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = 8u * U.l[i];                       // limbs <= 8 (2^28 - 1), value < 16p
    F28_SUBCHK(E.l[i], f28_24p_t8(i), s);
    p.x.l[i] = E.l[i] + f28_24p_t8(i) - s;                // X3 = F - 8U: limbs < 10*2^28, value < 26p
  }
  f28_carry_pass(p.x);    // limbs <= 2^28 + 9
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(4u * U.l[i], f28_32p_t1(i), p.x.l[i]);
    T.l[i] = 4u * U.l[i] + f28_32p_t1(i) - p.x.l[i];      // 4U - X3: limbs < 2^30 + 2^29, value < 40p
  }
  f28_mul(T, A, T);       // 14 * 2^28 * 1.5 * 2^30 = 2^62.4 ; 2 * 40
  f28_mul(U, p.y, p.z);   // Y Z (old Y): 14 * (2^28+16) * 2^29 ; 30 * 4
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = 8u * C.l[i];
    F28_SUBCHK(3u * T.l[i], f28_24p_t8(i), s);
    p.y.l[i] = 3u * T.l[i] + f28_24p_t8(i) - s;           // Y3: limbs < 12*2^28, value < 30p
    p.z.l[i] = 2u * U.l[i];                               // Z3: limbs < 2^29, value < 4p
  }
  f28_carry_pass(p.y);    // limbs <= 2^28 + 11
}

// p += (x2, y2), an affine point in N-form; complete (identity, P + P, P + (-P)).
// U2 = x2 Z^2, S2 = y2 Z^3, H = U2 - X, r = S2 - Y, X3 = r^2 - H^3 - 2 X H^2, Y3 = r (X H^2 - X3) - Y H^3, Z3 = Z H.
KZG_HD void jac28_madd(g1_jac28& p, const fp28& x2, const fp28& y2) {
  if (p.inf) {
    p.x = x2;
    p.y = y2;
    p.z = f28_one();
    p.inf = 0;
    return;
  }
  fp28 zz, zzz, h, r, hh, hhh, v, t;
  f28_sqr(zz, p.z);        // 14 * 2^58 ; 16
  f28_mul(zzz, p.z, zz);   // 8
  f28_mul(h, x2, zz);      // U2
  f28_mul(r, y2, zzz);     // S2
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(h.l[i], f28_32p_t1(i), p.x.l[i]);
    F28_SUBCHK(r.l[i], f28_32p_t1(i), p.y.l[i]);
    h.l[i] = h.l[i] + f28_32p_t1(i) - p.x.l[i];  // H: limbs < 3*2^28, value < 34p
    r.l[i] = r.l[i] + f28_32p_t1(i) - p.y.l[i];  // r
  }
  if (f28_is_zero(h)) {
    if (f28_is_zero(r)) {  // the same point: double it
      g1_jac28 q;
      q.x = x2;
      q.y = y2;
      q.z = f28_one();
      q.inf = 0;
      jac28_dbl(q);
      p = q;
    } else {
      p.inf = 1;
    }
    return;
  }
  f28_sqr(hh, h);          // 14 * 9 * 2^56 ; 34^2 = 1156 < 2^11
  f28_mul(hhh, h, hh);     // 68
  f28_mul(v, p.x, hh);     // X H^2: 52
  f28_sqr(t, r);           // 1156
  f28_mul(zz, p.z, h);     // Z3: 14 * 2^29 * 3 * 2^28 ; 4 * 34   (zz reused)
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    const uint32_t s = hhh.l[i] + 2u * v.l[i];   // limbs <= 3 (2^28 - 1), value < 6p
    F28_SUBCHK(t.l[i], f28_8p_t3(i), s);
    p.x.l[i] = t.l[i] + f28_8p_t3(i) - s;        // X3: limbs < 5*2^28, value < 10p
  }
  f28_carry_pass(p.x);     // limbs <= 2^28 + 4
  f28_sub_16p(v, v, p.x);  // X H^2 - X3: limbs < 3*2^28, value < 18p
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(0u, f28_32p_t1(i), p.y.l[i]);
    t.l[i] = f28_32p_t1(i) - p.y.l[i];           // -Y: limbs < 2^29, value <= 32p
  }
  f28_mul2(p.y, r, v, t, hhh);  // 14 * (9 * 2^56 + 2^57) ; 34 * 18 + 32 * 2 = 676
  p.z = zz;
}

// [z^2]P for the affine point (x, y) (N-form), z the BLS12-381 parameter: z^2 = 0xac45a4010001a4020000000100000000
KZG_HD void g1_mul_by_z2_jac28(g1_jac28& acc, const fp28& x, const fp28& y) {
  const uint64_t hi = 0xac45a4010001a402ull, lo = 0x0000000100000000ull;
  acc.x = x;
  acc.y = y;
  acc.z = f28_one();
  acc.inf = 0;
#pragma unroll 1
  for (int i = 126; i >= 0; i--) {
    issue_fair_tick(18);
    jac28_dbl(acc);  // inline: the accumulator stays in registers over the runs of doublings
    const uint64_t w = i >= 64 ? hi : lo;
    if ((w >> (i & 63)) & 1ull) jac28_madd(acc, x, y);
  }
}

// blst_p1_affine_in_g1 via the endomorphism (see g1_in_subgroup in g1.cuh for the argument):
// (x, y) in G1  <=>  (beta x, y) == -[z^2](x, y).   x, y: 2^392-Montgomery N-form.
KZG_HD bool g1_in_subgroup28(const fp28& x, const fp28& y) {
  g1_jac28 q;
  g1_mul_by_z2_jac28(q, x, y);
  if (q.inf) return false;
  fp28 beta, zz, t;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) beta.l[i] = f28_beta_limb(i);
  f28_sqr(zz, q.z);
  f28_mul(t, x, beta);
  f28_mul(t, t, zz);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) {
    F28_SUBCHK(t.l[i], f28_32p_t1(i), q.x.l[i]);
    t.l[i] = t.l[i] + f28_32p_t1(i) - q.x.l[i];  // beta x Z^2 - X: value < 34p
  }
  if (!f28_is_zero(t)) return false;
  f28_mul(zz, q.z, zz);    // Z^3
  f28_mul(t, y, zz);
  f28_add(t, t, q.y);      // y Z^3 + Y: value < 32p
  return f28_is_zero(t);
}

// Same contract as g1_decompress (g1.cuh): status code, canonical 2^384-Montgomery x and y (unless r392_out), *inf.
// r392_out: leave the coordinates in the 2^392-Montgomery domain (canonical, 12 x 32 limbs) -- the operand format of
// the radix-2^28 adders (k_var_buckets) -- instead of converting to 2^384.
KZG_HD int32_t g1_decompress28(fp_t& x, fp_t& y, bool& inf, const uint8_t* in48, bool r392_out = false) {
  inf = false;
  const uint8_t b0 = in48[0];
  if (!(b0 & 0x80)) return KZG_ERR_EC_INVALID_ENCODING;
  if (b0 & 0x40) {
    uint32_t o = b0 & 0x3F;
    for (int i = 1; i < 48; i++) o |= in48[i];
    if (o) return KZG_ERR_EC_INVALID_ENCODING;
    inf = true;
    bn_zero(x);
    bn_zero(y);
    return KZG_OK;
  }
  fp_t xp;
  fp_from_be_bytes_plain(xp, in48);
  xp.v[11] &= 0x1FFFFFFFu;
  if (bn_geq(xp, modulus<FpParams>())) return KZG_ERR_EC_INVALID_ENCODING;
  fp28 x28, y28, rhs, t, k;
  f28_from_bn(x28, xp);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) k.l[i] = f28_r2_limb(i);
  f28_mul(x28, x28, k);  // Montgomery, N-form
  f28_sqr(t, x28);
  f28_mul(rhs, t, x28);
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) rhs.l[i] += f28_b_limb(i);  // x^3 + 4: limbs < 2^29, value < 3p
  f28_mul(rhs, rhs, f28_one());  // back to N-form (the window table of the square root is built from it)
  f28_sqrt_candidate(y28, rhs);
  f28_sqr(t, y28);
  f28_sub_4p(t, t, rhs);
  if (!f28_is_zero(t)) return KZG_ERR_EC_NOT_ON_CURVE;
  // sign: compare the plain y with (p-1)/2; the flag says which root the encoder meant
  fp28 one_plain;
  KZG_UNROLL_FULL
  for (int i = 0; i < F28_N; i++) one_plain.l[i] = i == 0 ? 1u : 0u;
  f28_mul(t, y28, one_plain);  // plain y, N-form (< 2p)
  fp_t yp;
  f28_to_bn(yp, t);
  canonicalize<FpParams>(yp);
  const bool larger = fp_is_lex_larger_plain(yp);
  const bool flip = ((b0 & 0x20) != 0) != larger;
  if (flip) {
    fp28 ny;
    f28_neg_4p(ny, y28);            // 4p - y: limbs < 2^29, value <= 4p
    f28_mul(y28, ny, f28_one());    // N-form
  }
  if (!g1_in_subgroup28(x28, y28)) return KZG_ERR_EC_NOT_IN_GROUP;
  if (r392_out) {
    f28_to_bn(x, x28);  // N-form: limbs strictly normalised, value < 2p
    canonicalize<FpParams>(x);
    f28_to_bn(y, y28);
    canonicalize<FpParams>(y);
  } else {
    f28_to_fp(x, x28);
    f28_to_fp(y, y28);
  }
  return KZG_OK;
}

}  // namespace kzg
