"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- restatement of kateth `src/bls.rs`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product (`kateth_amd/`) never does.

What this file restates
-----------------------
kateth's `src/bls.rs` is a thin FFI wrapper over the third-party crate
`blst = "0.3.11"` (`/root/reference/Cargo.toml:7`), which is NOT vendored under
`/root/reference`, so the arithmetic below restates blst's *published*
behaviour (BLS12-381, ZCash serialisation) with plain Python integers, and is
anchored on kateth's own call sites:

* `Fr`                     -> `src/bls.rs:78-360`   (ints mod R here)
* `P1` (G1)                -> `src/bls.rs:362-552`  (affine tuples / None = infinity)
* `P2` (G2)                -> `src/bls.rs:554-570`
* `verify_pairings`        -> `src/bls.rs:572-598`

PARITY STATUS: "parity unpinned" against reference-owned golden vectors -- the
reference's only vectors for this path are consensus-spec-tests, an empty
submodule in this checkout (`/root/reference/.gitmodules:1-3`), and kateth
cannot be built here (no rustc/cargo, no blst).  The oracle is instead pinned
by public constants and by algebraic identities over the ceremony file the
reference's tests load (`trusted_setup_4096.json`, `src/kzg/setup.rs:299-303`);
see `tests/test_oracle_kat.py`.
"""
from __future__ import annotations

import hashlib

# --------------------------------------------------------------------------
# constants (public BLS12-381 parameters)
# --------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_X = 0xD201000000010000  # |z|; the curve parameter is z = -BLS_X
B1 = 4  # E : y^2 = x^3 + 4

G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G1_GEN = (G1_X, G1_Y)

G2_X = (
    0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
    0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
)
G2_Y = (
    0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
    0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
)
G2_GEN = (G2_X, G2_Y)


# --------------------------------------------------------------------------
# errors (src/bls.rs:21-50)
# --------------------------------------------------------------------------
class FiniteFieldError(Exception):
    """`bls::FiniteFieldError` (src/bls.rs:21-25). kind in {InvalidEncoding, NotInFiniteField}."""

    def __init__(self, kind: str):
        super().__init__(kind)
        self.kind = kind


class ECGroupError(Exception):
    """`bls::ECGroupError` (src/bls.rs:27-32). kind in {InvalidEncoding, NotInGroup, NotOnCurve}."""

    def __init__(self, kind: str):
        super().__init__(kind)
        self.kind = kind


# --------------------------------------------------------------------------
# Fr  (src/bls.rs:78-360).  Values are canonical ints in [0, R).
# --------------------------------------------------------------------------
FR_BYTES = 32


def fr_from_be_slice(b: bytes) -> int:
    """`Fr::from_be_slice` (src/bls.rs:130-139): length check, then `< r` check."""
    if len(b) != FR_BYTES:
        raise FiniteFieldError("InvalidEncoding")
    v = int.from_bytes(b, "big")
    if v >= R:  # blst_scalar_fr_check, src/bls.rs:113
        raise FiniteFieldError("NotInFiniteField")
    return v


def fr_to_be_bytes(v: int) -> bytes:
    """`Fr::to_be_bytes` (src/bls.rs:141-149)."""
    return int(v % R).to_bytes(32, "big")


def fr_hash_to(data: bytes) -> int:
    """`Fr::hash_to` (src/bls.rs:189-205): SHA-256, big-endian, reduced mod r
    (blst_fr_from_scalar reduces a >= r input)."""
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % R


def fr_pow_reference(x: int, power: int) -> int:
    """`Fr::pow` exactly as written (src/bls.rs:169-187), INCLUDING quirk Q2:
    for power == 0 the loop is skipped and the function returns x*1 = x."""
    out = x
    tmp = 1
    while power != 1 and power != 0:
        if power & 1:
            tmp = out * tmp % R
            power -= 1
        out = out * out % R
        power >>= 1
    return out * tmp % R


def fr_div(a: int, b: int) -> int:
    """`impl Div for Fr` (src/bls.rs:297-311): panics on zero divisor."""
    if b % R == 0:
        raise ZeroDivisionError("division by zero in finite field Fr")
    return a * pow(b, -1, R) % R


# --------------------------------------------------------------------------
# generic short-Weierstrass arithmetic over a field given by (add, sub, mul, inv)
# implemented twice (Fp ints, Fp2 tuples) for clarity rather than abstraction.
# --------------------------------------------------------------------------
def _fp_inv(a: int) -> int:
    return pow(a, -1, P)


# ---- G1: affine (x, y) ints, None = infinity --------------------------------
def g1_is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - (x * x * x + B1)) % P == 0


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    """Complete affine addition.  (kateth's `P1 + P1` goes through
    `blst_p1_add` (src/bls.rs:452-463) which is not doubling-safe -- SURVEY quirk
    Q3; equal operands never occur in the parity set, so the oracle uses the
    mathematically complete law.)"""
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * _fp_inv(2 * y1) % P
    else:
        lam = (y2 - y1) * _fp_inv(x2 - x1) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


# Jacobian helpers for speed (X, Y, Z); Z == 0 is infinity.
def _jac_double(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return _jac_double(p)
        return (1, 1, 0)
    H = (U2 - U1) % P
    I = 4 * H * H % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def _jac_from_affine(pt):
    return (1, 1, 0) if pt is None else (pt[0], pt[1], 1)


def _jac_to_affine(p):
    X, Y, Z = p
    if Z == 0:
        return None
    zi = _fp_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_mul(pt, k: int):
    """`impl Mul<Fr> for P1` (src/bls.rs:474-489): scalar multiplication."""
    k %= R
    acc = (1, 1, 0)
    base = _jac_from_affine(pt)
    while k:
        if k & 1:
            acc = _jac_add(acc, base)
        base = _jac_double(base)
        k >>= 1
    return _jac_to_affine(acc)


def g1_mul_unreduced(pt, k: int):
    """scalar multiplication WITHOUT reducing k mod r (for the subgroup check)."""
    acc = (1, 1, 0)
    base = _jac_from_affine(pt)
    while k:
        if k & 1:
            acc = _jac_add(acc, base)
        base = _jac_double(base)
        k >>= 1
    return _jac_to_affine(acc)


def g1_in_subgroup(pt) -> bool:
    """`blst_p1_affine_in_g1` (src/bls.rs:522): prime-order subgroup check,
    restated as the definition [r]P == O."""
    return g1_mul_unreduced(pt, R) is None


def g1_lincomb(points, scalars):
    """`P1::lincomb` (src/bls.rs:406-413): naive sum of scalar multiples."""
    acc = (1, 1, 0)
    for pt, s in zip(points, scalars):
        acc = _jac_add(acc, _jac_from_affine(g1_mul(pt, s)))
    return _jac_to_affine(acc)


def g1_lincomb_pippenger(points, scalars, window: int = 8):
    """`P1::lincomb_pippenger` (src/bls.rs:416-437) -> blst `p1_affines::mult`.
    blst's own window/recoding is an implementation detail of the missing
    dependency; the *result* is the unique group element sum_i s_i*P_i, which a
    plain unsigned-window bucket method reproduces."""
    n = min(len(points), len(scalars))
    jp = [_jac_from_affine(points[i]) for i in range(n)]
    sc = [scalars[i] % R for i in range(n)]
    nwin = (255 + window - 1) // window
    total = (1, 1, 0)
    for w in reversed(range(nwin)):
        for _ in range(window):
            total = _jac_double(total)
        buckets = [(1, 1, 0)] * (1 << window)
        shift = w * window
        mask = (1 << window) - 1
        for i in range(n):
            d = (sc[i] >> shift) & mask
            if d:
                buckets[d] = _jac_add(buckets[d], jp[i])
        run = (1, 1, 0)
        acc = (1, 1, 0)
        for d in range((1 << window) - 1, 0, -1):
            run = _jac_add(run, buckets[d])
            acc = _jac_add(acc, run)
        total = _jac_add(total, acc)
    return _jac_to_affine(total)


def _fp_sqrt(a: int):
    """p = 3 mod 4 -> candidate a^((p+1)/4); None if a is a non-residue."""
    a %= P
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None


def g1_compress(pt) -> bytes:
    """`Compress for P1` (src/bls.rs:491-503) -> `blst_p1_compress`.
    ZCash format: 48-B big-endian x; bit7 = compressed, bit6 = infinity,
    bit5 = y is the lexicographically larger root (y > (p-1)/2)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if y > (P - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g1_uncompress(b: bytes):
    """`blst_p1_uncompress` as used at src/bls.rs:514-521: returns the affine
    point or raises InvalidEncoding / NotOnCurve.  NO subgroup check."""
    if len(b) != 48:
        raise ECGroupError("InvalidEncoding")
    b0 = b[0]
    if not (b0 & 0x80):  # compressed flag must be set
        raise ECGroupError("InvalidEncoding")
    if b0 & 0x40:  # infinity: every other bit (including the sign bit) must be 0
        if (b0 & 0x3F) == 0 and not any(b[1:]):
            return None
        raise ECGroupError("InvalidEncoding")
    x = int.from_bytes(bytes([b0 & 0x1F]) + b[1:], "big")
    if x >= P:
        raise ECGroupError("InvalidEncoding")
    y = _fp_sqrt(x * x * x + B1)
    if y is None:
        raise ECGroupError("NotOnCurve")
    y_is_larger = y > (P - 1) // 2
    if bool(b0 & 0x20) != y_is_larger:
        y = (-y) % P
    return (x, y)


def g1_decompress(b: bytes):
    """`Decompress for P1` (src/bls.rs:505-531): uncompress + subgroup check."""
    pt = g1_uncompress(b)
    if not g1_in_subgroup(pt):
        raise ECGroupError("NotInGroup")
    return pt


# --------------------------------------------------------------------------
# Fp2 = Fp[u]/(u^2+1): tuples (c0, c1)
# --------------------------------------------------------------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return f2_mul(a, a)


def f2_muls(a, s: int):
    return (a[0] * s % P, a[1] * s % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


def f2_inv(a):
    d = _fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e: int):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)  # non-residue 1+u; Fp6 = Fp2[v]/(v^3 - XI), Fp12 = Fp6[w]/(w^2 - v)
B2 = (4, 4)  # E' : y^2 = x^3 + 4(1+u)


def f2_sqrt(a):
    """square root in Fp2 (p = 3 mod 4), or None."""
    if a == F2_ZERO:
        return F2_ZERO
    # Algorithm 9 of Adj & Rodriguez-Henriquez
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), a)
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        cand = (-x0[1] % P, x0[0])  # u * x0
    else:
        b = f2_pow(f2_add(F2_ONE, alpha), (P - 1) // 2)
        cand = f2_mul(b, x0)
    return cand if f2_sqr(cand) == (a[0] % P, a[1] % P) else None


# ---- G2: affine ((x0,x1),(y0,y1)), None = infinity --------------------------
def g2_is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B2)) == F2_ZERO


def g2_neg(pt):
    return None if pt is None else (pt[0], f2_neg(pt[1]))


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return None
        lam = f2_mul(f2_muls(f2_sqr(x1), 3), f2_inv(f2_muls(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_sqr(lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul_unreduced(pt, k: int):
    acc = None
    base = pt
    while k:
        if k & 1:
            acc = g2_add(acc, base)
        base = g2_add(base, base)
        k >>= 1
    return acc


def g2_mul(pt, k: int):
    """`impl Mul<Fr> for P2` (src/bls.rs:474-489, macro instance :554-570)."""
    return g2_mul_unreduced(pt, k % R)


def g2_in_subgroup(pt) -> bool:
    return g2_mul_unreduced(pt, R) is None


def g2_compress(pt) -> bytes:
    """`blst_p2_compress`: 96 B = x.c1 || x.c0 big-endian, flags in byte 0;
    sign = y lexicographically larger, compared on (c1, then c0)."""
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), (y0, y1) = pt
    out = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    out[0] |= 0x80
    larger = (y1 > (P - 1) // 2) if y1 != 0 else (y0 > (P - 1) // 2)
    if larger:
        out[0] |= 0x20
    return bytes(out)


def g2_uncompress(b: bytes):
    if len(b) != 96:
        raise ECGroupError("InvalidEncoding")
    b0 = b[0]
    if not (b0 & 0x80):
        raise ECGroupError("InvalidEncoding")
    if b0 & 0x40:
        if (b0 & 0x3F) == 0 and not any(b[1:]):
            return None
        raise ECGroupError("InvalidEncoding")
    x1 = int.from_bytes(bytes([b0 & 0x1F]) + b[1:48], "big")
    x0 = int.from_bytes(b[48:], "big")
    if x1 >= P or x0 >= P:
        raise ECGroupError("InvalidEncoding")
    x = (x0, x1)
    y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
    if y is None:
        raise ECGroupError("NotOnCurve")
    y0, y1 = y
    larger = (y1 > (P - 1) // 2) if y1 != 0 else (y0 > (P - 1) // 2)
    if bool(b0 & 0x20) != larger:
        y = f2_neg(y)
    return (x, y)


def g2_decompress(b: bytes):
    """`Decompress for P2` (src/bls.rs:505-531 via :554-570)."""
    pt = g2_uncompress(b)
    if not g2_in_subgroup(pt):
        raise ECGroupError("NotInGroup")
    return pt


# --------------------------------------------------------------------------
# Fp6 / Fp12 tower and the optimal-ate pairing (src/bls.rs:572-598 ->
# blst_miller_loop / blst_final_exp / blst_fp12_is_one)
# Fp6 element: (c0, c1, c2) of Fp2, v^3 = XI.  Fp12 element: (a, b) of Fp6, w^2 = v.
# --------------------------------------------------------------------------
def f2_mul_xi(a):
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))


def f6_sub(a, b):
    return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))


def f6_neg(a):
    return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = f2_mul(a0, b0)
    t1 = f2_mul(a1, b1)
    t2 = f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(t2))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a2, b0)), t1)
    return (c0, c1, c2)


def f6_mul_by_v(a):
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_inv(a):
    c0, c1, c2 = a
    t0 = f2_sub(f2_sqr(c0), f2_mul_xi(f2_mul(c1, c2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(c2)), f2_mul(c0, c1))
    t2 = f2_sub(f2_sqr(c1), f2_mul(c0, c2))
    d = f2_add(f2_mul(c0, t0), f2_mul_xi(f2_add(f2_mul(c2, t1), f2_mul(c1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))


F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0 = f6_mul(a0, b0)
    t1 = f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_by_v(t1))
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a0, a1), f6_add(b0, b1)), t0), t1)
    return (c0, c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    a0, a1 = a
    d = f6_sub(f6_mul(a0, a0), f6_mul_by_v(f6_mul(a1, a1)))
    di = f6_inv(d)
    return (f6_mul(a0, di), f6_neg(f6_mul(a1, di)))


def f12_pow(a, e: int):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_sqr(a)
        e >>= 1
    return r


def _f12_from_sparse(c_w0, c_w2, c_w3):
    """element  c_w0 + c_w2*w^2 + c_w3*w^3  with w^2 = v:
    w^0 -> a.c0 ; w^2 = v -> a.c1 ; w^3 = v*w -> b.c1."""
    return ((c_w0, c_w2, F2_ZERO), (F2_ZERO, c_w3, F2_ZERO))


def _line(T, lam, Pt):
    """line through T (on the twist E') with twist-slope lam, evaluated at the
    G1 point Pt and scaled by w^3 (a factor in the proper subfield Fp4, killed
    by the final exponentiation):
        y_P*w^3 - lam*x_P*w^2 + (lam*x_T - y_T)."""
    xT, yT = T
    xP, yP = Pt
    return _f12_from_sparse(f2_sub(f2_mul(lam, xT), yT), f2_neg(f2_muls(lam, xP)), (yP, 0))


def miller_loop(Q, Pt):
    """optimal-ate Miller loop f_{|z|,Q}(P) (`blst_miller_loop`, src/bls.rs:591-592).
    Q in G2 (affine over Fp2), Pt in G1 (affine).  Returns 1 if either is infinity."""
    if Q is None or Pt is None:
        return F12_ONE
    f = F12_ONE
    T = Q
    for bit in bin(BLS_X)[3:]:
        xT, yT = T
        lam = f2_mul(f2_muls(f2_sqr(xT), 3), f2_inv(f2_muls(yT, 2)))
        f = f12_mul(f12_sqr(f), _line(T, lam, Pt))
        T = g2_add(T, T)
        if bit == "1":
            lam = f2_mul(f2_sub(Q[1], T[1]), f2_inv(f2_sub(Q[0], T[0])))
            f = f12_mul(f, _line(T, lam, Pt))
            T = g2_add(T, Q)
    # z < 0: f_{z,Q} = 1/f_{|z|,Q} up to subfield factors -> conjugate after the
    # easy part; conjugating here is equivalent for the final result.
    return f12_conj(f)


def final_exp_is_one(f) -> bool:
    """`blst_final_exp` + `blst_fp12_is_one` (src/bls.rs:595-596).
    Computed straight from the definition f^((p^12-1)/r) == 1, no addition
    chain: easy part (p^6-1) via conjugate/inverse, the rest by plain pow."""
    f = f12_mul(f12_conj(f), f12_inv(f))  # f^(p^6 - 1)
    e = (P**6 + 1) // R
    assert (P**6 + 1) % R == 0
    return f12_pow(f, e) == F12_ONE


def verify_pairings(pair_a, pair_b) -> bool:
    """`bls::verify_pairings` (src/bls.rs:572-598):
    e(-a1, a2) * e(b1, b2) == 1."""
    (a1, a2), (b1, b2) = pair_a, pair_b
    e1 = miller_loop(a2, g1_neg(a1))
    e2 = miller_loop(b2, b1)
    return final_exp_is_one(f12_mul(e1, e2))
