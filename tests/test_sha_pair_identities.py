"""The lane-pair SHA-256 rounds (kateth_amd/csrc/sha256.cuh, sha256_rounds_pair) rest on three identities; the kernel itself is
device-only (DPP lane exchange) and is covered by the GPU parity tests, the identities are checked here on the CPU:
  * Maj(a, b, c) = Ch(~(a ^ b), b, c);
  * the selector v_bitop3_b32(a0, a1, role, 0xD2) is a0 for role = 0 and ~(a0 ^ a1) for role = all-ones;
  * one round computed as (X: T1 + h-half, Y: T2-half) with the two exchanges gives the FIPS 180-4 round, so 64 of them give
    the reference compression function."""
import hashlib
import random
import struct

M = 0xFFFFFFFF
K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M


def ch(e, f, g):
    return ((e & f) ^ (~e & g)) & M


def maj(a, b, c):
    return ((a & b) ^ (a & c) ^ (b & c)) & M


def bitop3(a, b, c, table):
    """v_bitop3_b32: result bit = table[(a_bit << 2) | (b_bit << 1) | c_bit]"""
    out = 0
    for i in range(32):
        idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1)
        out |= ((table >> idx) & 1) << i
    return out


def test_maj_is_ch_with_a_xor_b_selector():
    rnd = random.Random(1)
    for _ in range(2000):
        a, b, c = (rnd.getrandbits(32) for _ in range(3))
        assert maj(a, b, c) == ch(~(a ^ b) & M, b, c)


def test_selector_truth_table_0xD2():
    rnd = random.Random(2)
    for _ in range(500):
        a0, a1 = rnd.getrandbits(32), rnd.getrandbits(32)
        assert bitop3(a0, a1, 0, 0xD2) == a0
        assert bitop3(a0, a1, M, 0xD2) == (~(a0 ^ a1)) & M
    assert bitop3(0xF0F0F0F0, 0xCCCCCCCC, 0xAAAAAAAA, 0xCA) == ch(0xF0F0F0F0, 0xCCCCCCCC, 0xAAAAAAAA)  # the Ch table the kernel uses


def pair_compress(state, w64):
    """the kernel's data flow: X = (e, f, g, h), Y = (a, b, c, d); same operations on both with role-dependent constants"""
    X = [state[4], state[5], state[6], state[7]]
    Y = [state[0], state[1], state[2], state[3]]
    for i in range(64):
        out = {}
        for role, s, rots, wk in (("X", X, (6, 11, 25), (w64[i] + K[i]) & M), ("Y", Y, (2, 13, 22), 0)):
            ymask = M if role == "Y" else 0
            sig = rotr(s[0], rots[0]) ^ rotr(s[0], rots[1]) ^ rotr(s[0], rots[2])
            sel = bitop3(s[0], s[1], ymask, 0xD2)
            c = bitop3(sel, s[1], s[2], 0xCA)
            t = (sig + c + wk + (s[3] & (~ymask & M))) & M
            out[role] = (t, s[3] if role == "Y" else t)  # (own t, what this lane sends)
        n0x = (out["X"][0] + out["Y"][1]) & M  # e' = T1 + d
        n0y = (out["Y"][0] + out["X"][1]) & M  # a' = T2 + T1
        X = [n0x, X[0], X[1], X[2]]
        Y = [n0y, Y[0], Y[1], Y[2]]
    return [(state[k] + Y[k]) & M for k in range(4)] + [(state[4 + k] + X[k]) & M for k in range(4)]


def test_pair_rounds_equal_sha256():
    rnd = random.Random(3)
    for n in (0, 1, 55, 56, 64, 131):
        msg = bytes(rnd.getrandbits(8) for _ in range(n))
        padded = msg + b"\x80" + b"\x00" * ((55 - n) % 64) + struct.pack(">Q", 8 * n)
        st = list(H0)
        for off in range(0, len(padded), 64):
            w = list(struct.unpack(">16I", padded[off:off + 64]))
            for i in range(16, 64):
                s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
                s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
                w.append((w[i - 16] + s0 + w[i - 7] + s1) & M)
            st = pair_compress(st, w)
        assert struct.pack(">8I", *st) == hashlib.sha256(msg).digest(), n
