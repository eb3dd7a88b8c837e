#!/usr/bin/env python3
"""Python model of the signed radix-2^30 Montgomery arithmetic planned for the fixed-base MSM's hot loop (fp30.cuh): 13 centred
limbs, centred quotient digits, column sums in a signed 64-bit accumulator.  Checks the algebra and the column bounds on random
and on extreme operands before any HIP is written (round 5)."""
import random

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
N, W = 13, 30
R = 1 << (N * W)
H = 1 << (W - 1)
MASK = (1 << W) - 1
INV = (-pow(P, -1, 1 << W)) % (1 << W)


def centred(v, top_free=True):
    out = []
    for _ in range(N - 1):
        l = v & MASK
        if l >= H:
            l -= 1 << W
        out.append(l)
        v = (v - l) >> W
    out.append(v)
    return out


PC = centred(P)


def val(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def sbfe(x):
    x &= MASK
    return x - (1 << W) if x >= H else x


worst = [0]


def chk(a):
    assert -(1 << 63) <= a < (1 << 63), "column overflow %d" % a.bit_length()
    worst[0] = max(worst[0], abs(a))
    return a


def mul(a, b, c=None, d=None):
    """(a*b [+ c*d]) / 2^390 mod p, centred limbs out"""
    q, r, A = [0] * N, [0] * N, 0
    for k in range(2 * N):
        lo, hi = max(0, k - N + 1), min(k, N - 1)
        for i in range(lo, hi + 1):
            A = chk(A + a[i] * b[k - i])
            if c is not None:
                A = chk(A + c[i] * d[k - i])
        if k < N:
            for i in range(k):
                A = chk(A + q[i] * PC[k - i])
            q[k] = sbfe((A & 0xFFFFFFFF) * INV)
            A = chk(A + q[k] * PC[0])
            assert A & MASK == 0
            A >>= W
        else:
            for i in range(lo, hi + 1):
                A = chk(A + q[i] * PC[k - i])
            if k < 2 * N - 1:
                r[k - N] = sbfe(A)
                A = chk(A + H) >> W  # rounding shift: (A - r) / 2^30
            else:
                r[N - 1] = A
    return r


def carry(a):
    """parallel carry pass: limbs 0..11 back to [-2^29, 2^29) + a carry of a few units"""
    out = list(a)
    c = [(x + H) >> W for x in a[:N - 1]]
    for i in range(N - 1):
        out[i] = sbfe(a[i])
    for i in range(N - 1):
        out[i + 1] += c[i]
    assert val(out) == val(a)
    return out


def check_mul(rnd, n=300):
    for _ in range(n):
        x, y = rnd.randrange(-2 * P, 2 * P), rnd.randrange(-2 * P, 2 * P)
        a, b = centred(x), centred(y)
        r = mul(a, b)
        assert (val(r) * R - x * y) % P == 0
        assert abs(val(r)) < 0.54 * P, abs(val(r)) / P
        assert all(-H <= l < H for l in r[:N - 1]) and abs(r[N - 1]) < 1 << 22


def extreme():
    """operands with every full limb at +-2^29 (C-form extreme) and at +-2^30 (one lazy operand)"""
    for sa in (1, -1):
        for sb in (1, -1):
            a = [sa * (H + 2)] * (N - 1) + [sa * (1 << 21)]
            b = [sb * (H + 2)] * (N - 1) + [sb * (1 << 21)]
            r = mul(a, b)
            assert (val(r) * R - val(a) * val(b)) % P == 0
            lazy = [sa * (2 * H + 4)] * (N - 1) + [sa * (1 << 22)]
            r = mul(lazy, b)
            assert (val(r) * R - val(lazy) * val(b)) % P == 0
            r = mul(a, b, b, a)  # two products, one reduction: all four operands C-form
            assert (val(r) * R - 2 * val(a) * val(b)) % P == 0


if __name__ == "__main__":
    rnd = random.Random(30)
    check_mul(rnd)
    extreme()
    import math

    print("centred p:", [hex(x) for x in PC])
    print("sum |p_j| = 2^%.3f; worst column seen 2^%.3f (limit 2^63)" % (math.log2(sum(abs(x) for x in PC)), math.log2(worst[0])))
    x = rnd.randrange(P)
    a = centred(x)
    a2 = [2 * l for l in a]
    assert val(carry(a2)) == 2 * x and all(-H - 2 <= l <= H + 2 for l in carry(a2)[:N - 1])
    print("ok")
