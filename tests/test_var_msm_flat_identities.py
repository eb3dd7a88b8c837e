"""The flat variable-base MSM path (kateth_amd/csrc/verify_kernels.cuh: VarGeom with top_n != 0, k_var_bitsums; geometry
chosen in engine_verify.hip) rests on integer identities that need no GPU:
  * the signed base-2^13 digits of var_digits() reproduce the scalar, lie in [1, 4096] in magnitude, and the top (20th)
    window of any scalar below r is non-negative and at most 232 -- the bound behind top_n = 256;
  * k_var_bitsums' "di-th number with bit b set" enumeration visits exactly the magnitudes with that bit, in increasing order;
  * sum_d d * B_d = sum_b 2^b * sum_{d: bit b set} B_d, which lets the host's Horner loop take the bit sums directly."""
import random

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
C, W, HALF = 13, 20, 1 << 12


def var_digits(s, c=C, w=W):
    """signed digits as the device computes them (verify_kernels.cuh var_digits): returns [(window, magnitude, negative)]"""
    half, mask, out, carry = 1 << (c - 1), (1 << c) - 1, [], 0
    for j in range(w):
        u = (s & mask) + carry
        s >>= c
        neg = u > half
        d = (1 << c) - u if neg else u
        carry = 1 if neg else 0
        if d:
            out.append((j, d, neg))
    assert carry == 0 and s == 0
    return out


def test_digits_reproduce_the_scalar_and_bound_the_top_window():
    rnd = random.Random(13)
    top_max = 0
    for s in [0, 1, R - 1, R - 2, (1 << 254) - 1, (1 << 247) - 1, 1 << 247, 0x73ED << 240] + [rnd.randrange(R) for _ in range(3000)]:
        if s >= R:
            continue
        digs = var_digits(s)
        assert sum((-d if neg else d) << (C * j) for j, d, neg in digs) == s
        assert all(1 <= d <= HALF for _, d, _ in digs)
        for j, d, neg in digs:
            if j == W - 1:
                assert not neg
                top_max = max(top_max, d)
    assert top_max <= 232 == ((R >> 247) + 1)


def nth_with_bit(di, b):
    return ((di >> b) << (b + 1)) | (1 << b) | (di & ((1 << b) - 1))


def test_bit_enumeration_is_exact_and_increasing():
    for limit in (HALF, 256):
        for b in range(C):
            seen, prev = [], 0
            for di in range(limit // 2 + 1):
                d = nth_with_bit(di, b)
                if d > limit:
                    break
                assert d > prev
                prev = d
                seen.append(d)
            assert seen == [d for d in range(1, limit + 1) if (d >> b) & 1], (limit, b)


def test_bit_sums_equal_the_weighted_bucket_sum():
    rnd = random.Random(7)
    buckets = [rnd.randrange(1 << 64) for _ in range(HALF)]  # B_d for d = 1..4096 (integers stand in for points)
    direct = sum(d * buckets[d - 1] for d in range(1, HALF + 1))
    by_bits = sum((1 << b) * sum(buckets[d - 1] for d in range(1, HALF + 1) if (d >> b) & 1) for b in range(C))
    assert direct == by_bits
