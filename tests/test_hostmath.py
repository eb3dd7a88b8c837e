"""CPU unit tests of the product's single-source field / curve / hash headers
(kateth_amd/csrc/{field,g1,sha256}.cuh compiled for the host by g++), checked
against the Python oracle.  The same source is what the HIP kernels inline."""
import ctypes
import hashlib
import os
import random
import subprocess

import pytest

from oracle.pyref import bls

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostmath")
P, R = bls.P, bls.R


def _build(so, extra):
    src = os.path.join(HERE, "shim.cpp")
    hdr_dir = os.path.join(os.path.dirname(HERE), "..", "kateth_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(hdr_dir, f)) for f in os.listdir(hdr_dir) if f.endswith((".cuh", ".hpp")))
    if not os.path.exists(so) or os.path.getmtime(so) < max(newest, os.path.getmtime(src)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + extra + [src, "-o", so])
    return ctypes.CDLL(so)


@pytest.fixture(scope="module", params=["plain", "ubsan"])
def hm(request):
    """the host build of the device math, twice: plain, and under UndefinedBehaviorSanitizer (-fno-sanitize-recover: any
    shift / overflow / alignment / bounds UB in the single-source headers aborts the test process).  GPU sanitizers are
    not available on the pool; this is the CPU-side run."""
    if request.param == "plain":
        return _build(os.path.join(HERE, "libhostmath.so"), [])
    return _build(os.path.join(HERE, "libhostmath_ubsan.so"),
                  ["-g", "-fsanitize=undefined,bounds-strict,float-cast-overflow", "-fno-sanitize=vptr", "-fno-sanitize-recover=all", "-static-libubsan"])


def _op(lib, fn, op, a, b, nbytes):
    out = ctypes.create_string_buffer(nbytes)
    getattr(lib, fn)(op, out, int(a).to_bytes(nbytes, "little"), int(b).to_bytes(nbytes, "little"))
    return int.from_bytes(out.raw, "little")


@pytest.mark.parametrize("fn,mod,nb", [("hm_fp_op", P, 48), ("hm_fr_op", R, 32)])
def test_field_ops(hm, fn, mod, nb):
    rnd = random.Random(11)
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, (1 << (8 * nb - 8)) % mod]
    vals = edge + [rnd.randrange(mod) for _ in range(40)]
    for a in vals:
        for b in (vals[rnd.randrange(len(vals))], vals[rnd.randrange(len(vals))], a):
            assert _op(hm, fn, 0, a, b, nb) == a * b % mod
            assert _op(hm, fn, 1, a, b, nb) == (a + b) % mod
            assert _op(hm, fn, 2, a, b, nb) == (a - b) % mod
        assert _op(hm, fn, 3, a, 0, nb) == (-a) % mod
        assert _op(hm, fn, 4, a, 0, nb) == a * a % mod
    for a in vals[:12]:
        inv = _op(hm, fn, 5, a, 0, nb)
        assert inv == (pow(a, -1, mod) if a else 0)


def test_host_mulx_product_matches_portable(hm):
    """the generated mulx / adcx / adox Montgomery product the host's Fp arithmetic takes on x86-64 CPUs with BMI2 + ADX
    (kateth_amd/csrc/host_fp_mulx.hpp, tools/gen_host_mulx.py) against the portable unsigned-__int128 loop it replaces: canonical
    and lazy forms, 60,000 chained products with the edge operands 0 and p - 1 mixed in; and the committed header is what the
    generator writes"""
    import subprocess
    import sys

    bad = hm.hm_host_mulx_crosscheck(ctypes.c_uint64(0x5EED5), 60000)
    if bad == -1:
        pytest.skip("this CPU has no BMI2 + ADX (the portable path is the only one)")
    assert bad == 0
    root = os.path.dirname(os.path.dirname(HERE))
    hdr = os.path.join(root, "kateth_amd", "csrc", "host_fp_mulx.hpp")
    before = open(hdr).read()
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_host_mulx.py")])
    assert open(hdr).read() == before, "kateth_amd/csrc/host_fp_mulx.hpp is not what tools/gen_host_mulx.py generates"


def test_fp_sqrt(hm):
    rnd = random.Random(5)
    for _ in range(10):
        a = rnd.randrange(P)
        sq = a * a % P
        s = _op(hm, "hm_fp_op", 6, sq, 0, 48)
        assert s in (a, P - a)


def test_sha256_and_hash_to_fr(hm):
    rnd = random.Random(3)
    for ln in (0, 1, 20, 55, 56, 63, 64, 65, 119, 120, 128, 1000, 131152):
        msg = bytes(rnd.randrange(256) for _ in range(ln))
        out = ctypes.create_string_buffer(32)
        hm.hm_sha256(out, msg, ctypes.c_uint64(ln))
        assert out.raw == hashlib.sha256(msg).digest()
        hm.hm_hash_to_fr(out, msg, ctypes.c_uint64(ln))
        assert int.from_bytes(out.raw, "big") == bls.fr_hash_to(msg)


def _sum(hm, pts, check=0):
    out = ctypes.create_string_buffer(48)
    st = hm.hm_g1_sum(out, b"".join(pts), len(pts), check)
    return st, out.raw


def test_g1_madd_complete(hm):
    g = bls.G1_GEN
    pts = [bls.g1_mul(g, k) for k in (1, 2, 3, 5, 0xABCDEF123456789)]
    enc = [bls.g1_compress(p) for p in pts]
    st, out = _sum(hm, enc)
    want = None
    for p in pts:
        want = bls.g1_add(want, p)
    assert st == 0 and out == bls.g1_compress(want)
    # P + P (doubling branch), P + (-P) (cancellation), infinity operands
    st, out = _sum(hm, [enc[0], enc[0]])
    assert out == bls.g1_compress(bls.g1_mul(g, 2))
    st, out = _sum(hm, [enc[1], enc[2], bls.g1_compress(bls.g1_mul(g, 5))])  # (2G+3G) + 5G
    assert out == bls.g1_compress(bls.g1_mul(g, 10))
    st, out = _sum(hm, [enc[3], bls.g1_compress(bls.g1_neg(pts[3]))])
    assert out == bls.g1_compress(None)
    st, out = _sum(hm, [bls.g1_compress(None), enc[4], bls.g1_compress(None)])
    assert out == enc[4]
    st, out = _sum(hm, [])
    assert out == bls.g1_compress(None)


def test_g1_full_add_and_mul(hm):
    g = bls.G1_GEN
    a, b = bls.g1_mul(g, 77), bls.g1_mul(g, 1234567)
    out = ctypes.create_string_buffer(48)
    for x, y in ((a, b), (a, a), (a, bls.g1_neg(a)), (None, b), (a, None), (None, None)):
        assert hm.hm_g1_add_full(out, bls.g1_compress(x), bls.g1_compress(y)) == 0
        assert out.raw == bls.g1_compress(bls.g1_add(x, y))
    rnd = random.Random(9)
    for k in (0, 1, 2, R - 1, rnd.randrange(R), rnd.randrange(R)):
        assert hm.hm_g1_mul(out, bls.g1_compress(a), k.to_bytes(32, "big")) == 0
        assert out.raw == bls.g1_compress(bls.g1_mul(a, k))


def test_g1_decompress_status_codes(hm):
    gen = bls.g1_compress(bls.G1_GEN)
    assert hm.hm_g1_decompress_status(gen) == 0
    assert hm.hm_g1_decompress_status(bls.g1_compress(None)) == 0
    assert hm.hm_g1_decompress_status(bytes([gen[0] & 0x7F]) + gen[1:]) == 3
    assert hm.hm_g1_decompress_status(bytes([0x9A]) + bytes([0xFF] * 47)) == 3
    assert hm.hm_g1_decompress_status(bytes([0xE0]) + bytes(47)) == 3
    assert hm.hm_g1_decompress_status(bytes([0xC0]) + bytes(46) + b"\x01") == 3
    x = 1
    while bls._fp_sqrt(x**3 + 4) is not None:
        x += 1
    assert hm.hm_g1_decompress_status(bytes([0x80]) + x.to_bytes(48, "big")[1:]) == 4
    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    assert hm.hm_g1_decompress_status(bls.g1_compress((x, y))) == 5
    # both sign choices decode to the right y
    q = bls.g1_mul(bls.G1_GEN, 424242)
    for pt in (q, bls.g1_neg(q)):
        st, out = _sum(hm, [bls.g1_compress(pt)])
        assert st == 0 and out == bls.g1_compress(pt)


# ---- host pairing (kateth_amd/csrc/pairing.hpp), the once-per-call check of verify ----------
def test_host_pairing_against_oracle(hm, oracle_setup):
    g1, g2 = bls.G1_GEN, bls.G2_GEN
    a, b = 0x1234567, 0x89ABCDEF01
    c = lambda p: bls.g1_compress(p)
    c2 = lambda q: bls.g2_compress(q)
    # e(-(ab)G, H) * e(aG, bH) == 1
    assert hm.hm_pairing_check(c(bls.g1_mul(g1, a * b)), c2(g2), c(bls.g1_mul(g1, a)), c2(bls.g2_mul(g2, b))) == 1
    assert hm.hm_pairing_check(c(bls.g1_mul(g1, a * b + 1)), c2(g2), c(bls.g1_mul(g1, a)), c2(bls.g2_mul(g2, b))) == 0
    # infinity operands (reference Q4: two infinities verify as true)
    assert hm.hm_pairing_check(c(None), c2(g2), c(None), c2(bls.g2_mul(g2, b))) == 1
    assert hm.hm_pairing_check(c(None), c2(g2), c(g1), c2(g2)) == 0
    # ceremony consistency: e([tau]_1, G2) == e(G1, [tau]_2)
    from oracle.pyref import blob as oblob

    tau1 = oblob.commitment(oracle_setup.roots_of_unity_brp, oracle_setup)
    tau2 = c2(oracle_setup.g2_monomial[1])
    assert hm.hm_pairing_check(c(tau1), c2(g2), c(g1), tau2) == 1
    assert hm.hm_pairing_check(c(tau1), c2(g2), c(g1), c2(oracle_setup.g2_monomial[2])) == 0
    # same decisions as the oracle's pairing on a few mixed cases
    rnd = random.Random(4)
    for _ in range(3):
        x, y = rnd.randrange(R), rnd.randrange(R)
        good = rnd.random() < 0.5
        lhs = bls.g1_mul(g1, x * y % R if good else (x * y + 5) % R)
        got = hm.hm_pairing_check(c(lhs), c2(g2), c(bls.g1_mul(g1, x)), c2(bls.g2_mul(g2, y)))
        want = bls.verify_pairings((lhs, g2), (bls.g1_mul(g1, x), bls.g2_mul(g2, y)))
        assert got == int(want) == int(good)


def test_host_pairing_on_the_external_point_evaluation_vector(hm, oracle_setup):
    """the product's host pairing, G1 decoder and host lincomb on data produced outside this repository (the public EIP-4844
    point-evaluation precompile test, tests/golden/external-vectors/README.md): verify_proof_inner of src/kzg/setup.rs:84-94,
    e(-(C - [y]G), G2) * e(proof, [tau]_2 - [z]G2) == 1.  The G2 operand is the ceremony's g2_monomial[1] minus [z]G2 by the
    oracle's G2 arithmetic; everything in G1 and the pairing run in the product's code"""
    from test_oracle_kat import EXT_COMMITMENT, EXT_PROOF, EXT_Y, EXT_Z

    assert hm.hm_g1_decompress_status(EXT_COMMITMENT) == 0 and hm.hm_g1_decompress_status(EXT_PROOF) == 0
    out28 = ctypes.create_string_buffer(48)
    assert hm.hm_g1_decompress28(out28, EXT_COMMITMENT) == 0 and out28.raw == EXT_COMMITMENT
    z, y = int.from_bytes(EXT_Z, "big"), int.from_bytes(EXT_Y, "big")
    c2 = bls.g2_compress

    def decision(zz, yy):
        c_minus_y = ctypes.create_string_buffer(48)  # [0]proof - [y]G + C through host_lincomb.hpp
        assert hm.hm_single_item_lincomb(c_minus_y, bytes(32), EXT_PROOF, yy.to_bytes(32, "big"), EXT_COMMITMENT) == 0
        tau_minus_z = bls.g2_add(oracle_setup.g2_monomial[1], bls.g2_neg(bls.g2_mul(bls.G2_GEN, zz)))
        return hm.hm_pairing_check(c_minus_y.raw, c2(bls.G2_GEN), EXT_PROOF, c2(tau_minus_z))

    assert decision(z, y) == 1
    assert decision(z, (y + 1) % R) == 0
    assert decision((z + 1) % R, y) == 0


def test_host_final_exp_chain_matches_definition(hm):
    e = (P**12 - 1) // R
    nl = (e.bit_length() + 31) // 32
    eb = e.to_bytes(4 * nl, "little")
    res = hm.hm_final_exp_crosscheck(bls.g1_compress(bls.g1_mul(bls.G1_GEN, 77)), bls.g2_compress(bls.g2_mul(bls.G2_GEN, 99)), eb, nl)
    assert res == 1  # agree, and a non-degenerate pairing value is not one
    res = hm.hm_final_exp_crosscheck(bls.g1_compress(None), bls.g2_compress(bls.G2_GEN), eb, nl)
    assert res == 3  # agree, value is one
    assert hm.hm_frobenius_check(bls.g1_compress(bls.g1_mul(bls.G1_GEN, 5)), bls.g2_compress(bls.g2_mul(bls.G2_GEN, 7))) == 1


def test_host_g2_decompress(hm, oracle_setup):
    raw = __import__("json").load(open(__import__("conftest").TRUSTED_SETUP))
    for s in raw["g2_monomial"][:4] + raw["g2_monomial"][-2:]:
        assert hm.hm_g2_decompress_status(bytes.fromhex(s[2:])) == 0
    good = bytes.fromhex(raw["g2_monomial"][1][2:])
    assert hm.hm_g2_decompress_status(bytes([good[0] & 0x7F]) + good[1:]) == 3
    assert hm.hm_g2_decompress_status(bytes([0xC0]) + bytes(95)) == 0
    assert hm.hm_g2_decompress_status(bytes([0xE0]) + bytes(95)) == 3
    # flipped sign bit still decodes (the other root): status 0
    assert hm.hm_g2_decompress_status(bytes([good[0] ^ 0x20]) + good[1:]) == 0
    # an on-curve point outside G2
    x = (1, 0)
    while True:
        y = bls.f2_sqrt(bls.f2_add(bls.f2_mul(bls.f2_sqr(x), x), bls.B2))
        if y is not None and not bls.g2_in_subgroup((x, y)):
            break
        x = (x[0] + 1, 0)
    assert hm.hm_g2_decompress_status(bls.g2_compress((x, y))) == 5


def test_fast_subgroup_check_agrees_with_definition(hm):
    """endomorphism-based membership test == the definition [r]P == O, on members and non-members"""
    rnd = random.Random(12)
    for k in (1, 2, R - 1, rnd.randrange(R), rnd.randrange(R)):
        assert hm.hm_g1_subgroup_both(bls.g1_compress(bls.g1_mul(bls.G1_GEN, k))) == 3
    assert hm.hm_g1_subgroup_both(bls.g1_compress(None)) == 3
    found = 0
    x = 1
    while found < 6:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None:
            member = bls.g1_in_subgroup((x, y))
            got = hm.hm_g1_subgroup_both(bls.g1_compress((x, y)))
            assert got == (3 if member else 0)
            found += 0 if member else 1
        x += 1
    # a member plus a small-order cofactor point: multiply a non-member by r to land in the cofactor part
    t = bls.g1_mul_unreduced((x - 1, bls._fp_sqrt((x - 1) ** 3 + 4)), R) if bls._fp_sqrt((x - 1) ** 3 + 4) else None
    if t is not None:
        mixed = bls.g1_add(bls.g1_mul(bls.G1_GEN, 12345), t)
        assert hm.hm_g1_subgroup_both(bls.g1_compress(mixed)) == 0


def test_lazy_madd_matches_canonical(hm):
    """the lazy-reduction mixed add of the MSM hot loop (coordinates in [0, 2p)) gives the same point"""
    rnd = random.Random(21)
    g = bls.G1_GEN
    pts = [bls.g1_mul(g, rnd.randrange(1, R)) for _ in range(40)]
    enc = [bls.g1_compress(p) for p in pts]
    out = ctypes.create_string_buffer(48)
    assert hm.hm_g1_sum_lazy(out, b"".join(enc), len(enc)) == 0
    want = None
    for p in pts:
        want = bls.g1_add(want, p)
    assert out.raw == bls.g1_compress(want)
    # special cases through the lazy path: P + P, P - P, leading infinity
    assert hm.hm_g1_sum_lazy(out, enc[0] + enc[0], 2) == 0 and out.raw == bls.g1_compress(bls.g1_mul(pts[0], 2))
    assert hm.hm_g1_sum_lazy(out, enc[1] + enc[2] + bls.g1_compress(bls.g1_neg(bls.g1_add(pts[1], pts[2]))), 3) == 0 and out.raw == bls.g1_compress(None)
    assert hm.hm_g1_sum_lazy(out, enc[3] + enc[4] + bls.g1_compress(bls.g1_add(pts[3], pts[4])), 3) == 0
    assert out.raw == bls.g1_compress(bls.g1_mul(bls.g1_add(pts[3], pts[4]), 2))


# ---- radix-2^28 field of the fixed-base MSM hot loop (kateth_amd/csrc/fp28.cuh) -------------------
def _f28(hm, op, a, b=0, c=0, d=0):
    out = ctypes.create_string_buffer(48)
    hm.hm_f28_op(op, out, *(int(v).to_bytes(48, "little") for v in (a, b, c, d)))
    return out.raw


def test_fp28_field_ops(hm):
    """14 x 28-bit limb Montgomery arithmetic (radix 2^392) against Python integers; the CPU build re-computes every
    column sum in 128 bits and checks every limb subtraction, so a violated bound fails the test"""
    rnd = random.Random(28)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (1 << 380), (1 << 28) - 1, ((1 << 392) % P), P - ((1 << 392) % P)]
    vals = edge + [rnd.randrange(P) for _ in range(60)]
    for a in vals:
        assert int.from_bytes(_f28(hm, 3, a), "little") == a
        assert int.from_bytes(_f28(hm, 1, a), "little") == a * a % P
    for _ in range(300):
        a, b, c, d = (rnd.choice(vals) for _ in range(4))
        assert int.from_bytes(_f28(hm, 0, a, b), "little") == a * b % P
        assert int.from_bytes(_f28(hm, 2, a, b, c, d), "little") == (a * b + c * d) % P
        assert int.from_bytes(_f28(hm, 4, a, b), "little") == (a - b) % P
        assert int.from_bytes(_f28(hm, 5, a, b), "little") == (a - b) % P
        assert _f28(hm, 6, a, b)[0] == (1 if a == b else 0)
    for a in vals:
        assert _f28(hm, 6, a, a)[0] == 1
    assert hm.hm_f28_violations() == 0


def test_fp28_madd_complete(hm):
    """the hot-loop adder in the 2^28 representation: long random signed sums, P + P, P - P, leading/trailing cancellation"""
    rnd = random.Random(29)
    g = bls.G1_GEN
    pts = [bls.g1_mul(g, rnd.randrange(1, R)) for _ in range(48)]
    enc = [bls.g1_compress(p) for p in pts]
    out = ctypes.create_string_buffer(48)

    def run(idx, signs, force=0):
        assert hm.hm_g1_sum28(out, b"".join(enc[i] for i in idx), bytes(signs), len(idx), force) == 0
        want = None
        for i, s in zip(idx, signs):
            want = bls.g1_add(want, bls.g1_neg(pts[i]) if s else pts[i])
        assert out.raw == bls.g1_compress(want), (idx, signs)

    run(list(range(48)), [rnd.randrange(2) for _ in range(48)])
    assert hm.hm_f28_slow_calls() == 1  # only the first add (identity accumulator) left the hot path
    run(list(range(48)), [rnd.randrange(2) for _ in range(48)], force=1)  # generic case through the complete adder
    run([0, 0], [0, 0])  # P + P (first add after the identity)
    run([0, 0], [1, 1])  # (-P) + (-P)
    run([0, 0], [0, 1])  # P - P
    run([0, 0, 1], [0, 1, 0])  # identity again, then a point
    run([1, 2, 3, 3], [0, 0, 0, 0])
    run([1, 2, 1, 2], [0, 0, 1, 1])  # cancels through a non-trivial accumulator
    # accumulator equals the next entry with a non-trivial ZZ: (a + b) then + (a + b) as an affine point
    ab = bls.g1_add(pts[4], pts[5])
    assert hm.hm_g1_sum28(out, enc[4] + enc[5] + bls.g1_compress(ab), bytes([0, 0, 0]), 3, 0) == 0
    assert out.raw == bls.g1_compress(bls.g1_mul(ab, 2))
    assert hm.hm_g1_sum28(out, enc[4] + enc[5] + bls.g1_compress(ab), bytes([0, 0, 1]), 3, 0) == 0
    assert out.raw == bls.g1_compress(None)
    assert hm.hm_g1_sum28(out, enc[4] + enc[5] + bls.g1_compress(ab) + enc[6], bytes([1, 1, 1, 0]), 4, 0) == 0
    assert out.raw == bls.g1_compress(bls.g1_add(bls.g1_neg(bls.g1_mul(ab, 2)), pts[6]))
    for _ in range(6):
        n = rnd.randrange(1, 30)
        run([rnd.randrange(48) for _ in range(n)], [rnd.randrange(2) for _ in range(n)])
    assert hm.hm_f28_violations() == 0


# ---- signed radix-2^30 Fp (kateth_amd/csrc/fp30.cuh): the fixed-base MSM's field since round 5 -----------------------------------
F30_N, F30_H = 13, 1 << 29


def _c30(v):
    """centred digits of an integer: limbs 0..11 in [-2^29, 2^29), the rest in limb 12"""
    out = []
    for _ in range(F30_N - 1):
        l = v & ((1 << 30) - 1)
        if l >= F30_H:
            l -= 1 << 30
        out.append(l)
        v = (v - l) >> 30
    out.append(v)
    return out


def _v30(limbs):
    return sum(int(x) << (30 * i) for i, x in enumerate(limbs))


def _f30(hm, op, a, b=None, c=None, d=None):
    arr = lambda l: (ctypes.c_int32 * 13)(*(l if l is not None else [0] * 13))  # noqa: E731
    out = (ctypes.c_int32 * 13)()
    hm.hm_f30_op(op, out, arr(a), arr(b), arr(c), arr(d))
    return list(out)


def test_fp30_field_ops(hm):
    """13 centred 30-bit limbs, Montgomery radix 2^390, against Python integers -- with every column also formed in 128 bits by
    the checked build: products of C-forms, of one lazy (L-form) operand, squarings, double products; the extremes of every
    operand form (all limbs at +-(2^29 + 2), +-(2^30 + 4)); both carry passes; the packed table format; conversions"""
    rnd = random.Random(30)
    Rm = 1 << 390
    before = hm.hm_f28_violations()

    def ok_product(r, want_times_R):
        v = _v30(r)
        assert (v * Rm - want_times_R) % P == 0
        assert abs(v) < 0.54 * P and all(-F30_H <= x < F30_H for x in r[:12]) and abs(r[12]) < 1 << 22

    for _ in range(200):
        x, y, z, w = (rnd.randrange(-3 * P, 3 * P) for _ in range(4))
        a, b, c, d = _c30(x), _c30(y), _c30(z), _c30(w)
        ok_product(_f30(hm, 0, a, b), x * y)
        ok_product(_f30(hm, 1, a), x * x)
        ok_product(_f30(hm, 2, a, b, c, d), x * y + z * w)
        lazy = [p_ - q_ for p_, q_ in zip(a, c)]  # L-form: an unreduced difference
        ok_product(_f30(hm, 0, lazy, b), (x - z) * y)
        ok_product(_f30(hm, 0, b, lazy), (x - z) * y)
    for sa in (1, -1):
        for sb in (1, -1):
            a = [sa * (F30_H + 2)] * 12 + [sa * (1 << 21)]
            b = [sb * (F30_H + 2)] * 12 + [sb * (1 << 21)]
            lazy = [sa * (2 * F30_H + 4)] * 12 + [sa * (1 << 22)]
            ok_product(_f30(hm, 0, a, b), _v30(a) * _v30(b))
            ok_product(_f30(hm, 1, a), _v30(a) ** 2)
            ok_product(_f30(hm, 0, lazy, b), _v30(lazy) * _v30(b))
            ok_product(_f30(hm, 2, a, b, b, a), 2 * _v30(a) * _v30(b))
            ok_product(_f30(hm, 2, a, b, [-t for t in b], a), 0)
    # products with injected differences (the fast addition's P, R and V): the exact integer a b / 2^390 + k c [+ k' d], digits centred
    def ok_injected(r, want_times_R, bound):
        v = _v30(r)
        assert (v * Rm - want_times_R) % P == 0
        assert abs(v) < bound * P and all(-F30_H <= t < F30_H for t in r[:12]) and abs(r[12]) < 1 << 25

    for _ in range(200):
        x, y, z, w = (rnd.randrange(-3 * P, 3 * P) for _ in range(4))
        a, b, c, d = _c30(x), _c30(y), _c30(z), _c30(w)
        lazy_c = [p_ + q_ for p_, q_ in zip(c, d)]  # an L-form injected value
        ok_injected(_f30(hm, 10, a, b, c), x * y - z * Rm, 3.6)
        ok_injected(_f30(hm, 10, a, b, lazy_c), x * y - (z + w) * Rm, 6.6)
        ok_injected(_f30(hm, 11, a, a, c, d), x * x - (z + 3 * w) * Rm, 12.6)
        got = _f30(hm, 10, a, b, c)
        plain = _f30(hm, 0, a, b)
        assert _v30(got) == _v30(plain) - z  # the integer itself, not only its residue
    for sa in (1, -1):
        for sc in (1, -1):
            a = [sa * (F30_H + 2)] * 12 + [sa * (1 << 21)]
            c = [sc * (2 * F30_H + 4)] * 12 + [sc * (1 << 22)]
            ok_injected(_f30(hm, 10, a, a, c), _v30(a) ** 2 - _v30(c) * Rm, 40)
            ok_injected(_f30(hm, 11, a, a, c, c), _v30(a) ** 2 - 4 * _v30(c) * Rm, 160)
    # U-form results (floor digits: the accumulator's ZZ / ZZZ): the same integer as the centred result, limbs 0..11 in [0, 2^30);
    # a U-form operand with a C-form one, at the extremes of both
    for _ in range(200):
        x, y = rnd.randrange(-3 * P, 3 * P), rnd.randrange(-3 * P, 3 * P)
        a, b = _c30(x), _c30(y)
        u = _f30(hm, 12, a, b)
        assert _v30(u) == _v30(_f30(hm, 0, a, b)) and all(0 <= t < 1 << 30 for t in u[:12]) and abs(u[12]) < 1 << 22
        ok_product(_f30(hm, 0, u, b), _v30(u) * y)
        uu = _f30(hm, 12, u, b)  # U x C -> U, as ZZ3 = ZZ1 * PP
        assert (_v30(uu) * Rm - _v30(u) * y) % P == 0 and all(0 <= t < 1 << 30 for t in uu[:12])
    for sb in (1, -1):
        top = [(1 << 30) - 1] * 12 + [1 << 21]
        b = [sb * (F30_H + 2)] * 12 + [sb * (1 << 21)]
        ok_product(_f30(hm, 0, top, b), _v30(top) * _v30(b))
        ok_product(_f30(hm, 0, b, top), _v30(top) * _v30(b))
        ok_injected(_f30(hm, 10, top, b, b), _v30(top) * _v30(b) - _v30(b) * Rm, 8)
        uu = _f30(hm, 12, top, b)
        assert (_v30(uu) * Rm - _v30(top) * _v30(b)) % P == 0 and all(0 <= t < 1 << 30 for t in uu[:12])
    # carry passes: value preserved, limbs back in range
    for _ in range(100):
        l = [rnd.randrange(-(1 << 31) + (1 << 29) + 1, (1 << 31) - (1 << 29)) for _ in range(12)] + [rnd.randrange(-(1 << 24), 1 << 24)]
        r = _f30(hm, 3, l)
        assert _v30(r) == _v30(l) and all(abs(t) <= F30_H + 2 for t in r[:12])
        wide = [rnd.choice([(1 << 31) - 1, -(1 << 31) + 3, rnd.randrange(-(1 << 31) + 3, 1 << 31)]) for _ in range(12)] + [rnd.randrange(-(1 << 24), 1 << 24)]
        r = _f30(hm, 4, wide)
        assert _v30(r) == _v30(wide) and all(abs(t) <= F30_H + 2 for t in r[:12])
    # packed table format, conversions
    for v in [0, 1, P - 1, (P - 1) // 2, (P + 1) // 2] + [rnd.randrange(P) for _ in range(60)]:
        c = _c30(v)
        assert _f30(hm, 5, c) == c
        words = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
        as_i32 = [w - (1 << 32) if w >= 1 << 31 else w for w in words] + [0]
        assert _v30(_f30(hm, 7, as_i32)) == v
        x = v  # read v as x * 2^384 (canonical Montgomery form of field.cuh): the table format holds x * 2^390
        got = _v30(_f30(hm, 9, as_i32))
        assert 0 <= got < P and (got - v * (1 << 6)) % P == 0
        back = _f30(hm, 6, _c30(got))  # x * 2^390 -> canonical x * 2^384
        assert sum((t & 0xFFFFFFFF) << (32 * i) for i, t in enumerate(back[:12])) == x
        neg = _f30(hm, 6, _c30(got - 2 * P))  # any representative
        assert sum((t & 0xFFFFFFFF) << (32 * i) for i, t in enumerate(neg[:12])) == x
    for k in (-3, -1, 0, 1, 2):
        assert _f30(hm, 8, _c30(k * P))[0] == 1
        assert _f30(hm, 8, _c30(k * P + 1))[0] == 0
    assert hm.hm_f28_violations() == before


def test_fp30_madd_complete(hm):
    """the hot-loop adder in the signed 2^30 representation, fed with packed table entries: long random signed sums, P + P, P - P,
    leading / trailing cancellation, an accumulator that meets its own affine image, Horner doublings"""
    rnd = random.Random(31)
    g = bls.G1_GEN
    pts = [bls.g1_mul(g, rnd.randrange(1, R)) for _ in range(48)]
    enc = [bls.g1_compress(p) for p in pts]
    out = ctypes.create_string_buffer(48)
    before = hm.hm_f28_violations()

    def run(idx, signs, force=0, dbl=0):
        assert hm.hm_g1_sum30(out, b"".join(enc[i] for i in idx), bytes(signs), len(idx), force, dbl) == 0
        want = None
        for i, s in zip(idx, signs):
            want = bls.g1_add(want, bls.g1_neg(pts[i]) if s else pts[i])
        want = bls.g1_mul(want, 1 << dbl) if want is not None else None
        assert out.raw == bls.g1_compress(want), (idx, signs)

    slow0 = hm.hm_f30_slow_calls()
    run(list(range(48)), [rnd.randrange(2) for _ in range(48)])
    assert hm.hm_f30_slow_calls() == slow0 + 1  # only the first add (identity accumulator) left the hot path
    run(list(range(48)), [rnd.randrange(2) for _ in range(48)], force=1)  # generic case through the complete adder
    run(list(range(48)), [rnd.randrange(2) for _ in range(48)], dbl=31)   # 31 Horner doublings, as a lane of the G = 8 comb does
    run([0, 0], [0, 0])
    run([0, 0], [1, 1])
    run([0, 0], [0, 1])
    run([0, 0, 1], [0, 1, 0])
    run([0, 0, 1], [0, 1, 0], dbl=3)
    run([1, 2, 3, 3], [0, 0, 0, 0])
    run([1, 2, 1, 2], [0, 0, 1, 1])
    ab = bls.g1_add(pts[4], pts[5])
    assert hm.hm_g1_sum30(out, enc[4] + enc[5] + bls.g1_compress(ab), bytes([0, 0, 0]), 3, 0, 0) == 0
    assert out.raw == bls.g1_compress(bls.g1_mul(ab, 2))
    assert hm.hm_g1_sum30(out, enc[4] + enc[5] + bls.g1_compress(ab), bytes([0, 0, 1]), 3, 0, 0) == 0
    assert out.raw == bls.g1_compress(None)
    assert hm.hm_g1_sum30(out, enc[4] + enc[5] + bls.g1_compress(ab) + enc[6], bytes([1, 1, 1, 0]), 4, 0, 2) == 0
    assert out.raw == bls.g1_compress(bls.g1_mul(bls.g1_add(bls.g1_neg(bls.g1_mul(ab, 2)), pts[6]), 4))
    for _ in range(8):
        n = rnd.randrange(1, 40)
        run([rnd.randrange(48) for _ in range(n)], [rnd.randrange(2) for _ in range(n)], dbl=rnd.randrange(4))
    assert hm.hm_f28_violations() == before


def test_glv_split_of_a_scalar(hm):
    """glv.cuh: k = k1 + k2 z^2 with k1 < z^2 < 2^128 and k2 = floor(k / z^2) < 2^128, for random scalars, the extremes of the field,
    multiples of z^2 and their neighbours (the Barrett estimate is corrected up to twice); and [z^2]P = (beta x, -y) on the curve"""
    rnd = random.Random(0x61F)
    Z = 0xD201000000010000
    Z2 = Z * Z
    out = ctypes.create_string_buffer(64)
    cases = [0, 1, Z2 - 1, Z2, Z2 + 1, R - 1, R - 2, (R - 1) // 2, 5 * Z2 - 1, (Z2 - 1) * Z2, (Z2 - 2) * Z2 + Z2 - 1]
    cases += [rnd.randrange(R) for _ in range(3000)] + [rnd.randrange(Z2) * Z2 + rnd.choice([0, 1, Z2 - 1]) for _ in range(300)]
    for k in cases:
        if k >= R:  # the engine's scalars are canonical (k2 = floor(k / z^2) < z^2 needs k < r < z^4)
            continue
        hm.hm_glv_split(out, k.to_bytes(32, "little"))
        k1, k2 = int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little")
        assert k1 == k % Z2 and k2 == k // Z2, hex(k)
        assert k1 < 1 << 128 and k2 < 1 << 128
    beta = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE
    pt = bls.g1_mul(bls.G1_GEN, 0xABCDEF)
    assert bls.g1_mul(pt, Z2) == (beta * pt[0] % P, (-pt[1]) % P)


def _f29(hm, op, a, b=0, c=0, d=0):
    out = ctypes.create_string_buffer(32)
    hm.hm_f29_op(op, out, *(int(v).to_bytes(32, "little") for v in (a, b, c, d)))
    return out.raw


def test_fr29_field_ops(hm):
    """9 x 29-bit limb Montgomery arithmetic for Fr (radix 2^261) against Python integers, with the 128-bit column and
    limb-subtraction checks of the CPU build armed"""
    rnd = random.Random(2929)
    edge = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, 1 << 254, (1 << 29) - 1, (1 << 261) % R, R - ((1 << 261) % R), 4096, pow(4096, -1, R)]
    vals = edge + [rnd.randrange(R) for _ in range(60)]
    for a in vals:
        assert int.from_bytes(_f29(hm, 3, a), "little") == a
        assert int.from_bytes(_f29(hm, 1, a), "little") == a * a % R
    for _ in range(300):
        a, b, c, d = (rnd.choice(vals) for _ in range(4))
        assert int.from_bytes(_f29(hm, 0, a, b), "little") == a * b % R
        assert int.from_bytes(_f29(hm, 2, a, b, c, d), "little") == (a * b + c * d) % R
        assert int.from_bytes(_f29(hm, 4, a, b), "little") == (a - b) % R
        assert _f29(hm, 5, a, b)[0] == (1 if a == b else 0)
    for a in vals:
        assert _f29(hm, 5, a, a)[0] == 1
    assert hm.hm_f28_violations() == 0


def test_g1_decompress_radix28_matches_reference_path(hm):
    """g1_decompress28 (square root, sign choice and subgroup test in the radix-2^28 field) against g1_decompress
    (12 x 32-bit limbs) and the oracle: same status codes, same decoded coordinates, on members, non-members of every
    kind, both signs, infinity and malformed encodings; the CPU build's 128-bit bound checks stay armed"""
    rnd = random.Random(2828)
    out = ctypes.create_string_buffer(48)

    def both(enc):
        rc = hm.hm_g1_decompress28(out, enc)
        assert rc >> 16 == 0, "decoded coordinates differ"
        assert rc & 0xFF == (rc >> 8) & 0xFF, "status codes differ: %x" % rc
        return rc & 0xFF

    for k in (1, 2, 3, R - 1, R - 2) + tuple(rnd.randrange(R) for _ in range(12)):
        pt = bls.g1_mul(bls.G1_GEN, k)
        for q in (pt, bls.g1_neg(pt)):
            enc = bls.g1_compress(q)
            assert both(enc) == 0 and out.raw == enc
    assert both(bls.g1_compress(None)) == 0 and out.raw == bls.g1_compress(None)
    gen = bls.g1_compress(bls.G1_GEN)
    assert both(bytes([gen[0] & 0x7F]) + gen[1:]) == 3
    assert both(bytes([0x9A]) + bytes([0xFF] * 47)) == 3
    assert both(bytes([0xE0]) + bytes(47)) == 3
    assert both(bytes([0xC0]) + bytes(46) + b"\x01") == 3
    pb = P.to_bytes(48, "big")
    assert both(bytes([0x80 | pb[0]]) + pb[1:]) == 3  # x == p is not canonical
    # x with no y; points on the curve outside the subgroup (several, both signs); member + cofactor component
    found_nc = found_ns = 0
    x = 1
    last_ns = None
    while found_nc < 4 or found_ns < 6:
        y = bls._fp_sqrt(x**3 + 4)
        if y is None:
            if found_nc < 4:
                assert both(bytes([0x80]) + x.to_bytes(48, "big")[1:]) == 4
                found_nc += 1
        elif not bls.g1_in_subgroup((x, y)):
            if found_ns < 6:
                assert both(bls.g1_compress((x, y))) == 5
                assert both(bls.g1_compress((x, P - y))) == 5
                found_ns += 1
                last_ns = (x, y)
        x += 1
    t = bls.g1_mul_unreduced(last_ns, R)  # lands in the cofactor part
    if t is not None:
        assert both(bls.g1_compress(bls.g1_add(bls.g1_mul(bls.G1_GEN, 777), t))) == 5
        assert both(bls.g1_compress(t)) == 5
        # small-order points: multiply further by cofactor factors until the order is tiny
        for f in (3, 11, 11 * 3, 10177, 859267):
            s = bls.g1_mul_unreduced(t, f)
            if s is not None:
                assert both(bls.g1_compress(s)) == 5
    # random x: whatever class each falls in (not on curve / outside the subgroup), both decoders must agree
    for _ in range(300):
        xr = rnd.randrange(P)
        code = both(bytes([0x80 | (xr >> 376) | (0x20 if rnd.random() < 0.5 else 0)]) + (xr & ((1 << 376) - 1)).to_bytes(47, "big"))
        assert code in (4, 5)
    # points of SMALL prime order q | h (h = 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2): the ladder meets P + P, P + (-P)
    # and the identity accumulator (order 3: [2]P = -P)
    h = 0x396C8C005555E1568C00AAAB0000AAAB
    assert h == 3 * 11**2 * 10177**2 * 859267**2 * 52437899**2
    seen = set()
    x = 5
    while len(seen) < 3 and x < 60:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            cof = bls.g1_mul_unreduced((x, y), R)
            for q in (3, 11, 10177):
                s_q = bls.g1_mul_unreduced(cof, h // q) if cof is not None else None
                if s_q is not None:
                    assert bls.g1_mul_unreduced(s_q, q) is None  # order exactly q
                    assert both(bls.g1_compress(s_q)) == 5
                    assert both(bls.g1_compress(bls.g1_neg(s_q))) == 5
                    seen.add(q)
        x += 1
    assert 3 in seen
    assert hm.hm_f28_violations() == 0


def test_g1_compress_radix28_matches_reference_path(hm):
    """g1_compress_xyzz28 (inversion by a sliding-window power in the radix-2^28 field) gives the bytes of
    g1_compress_xyzz and of the oracle, on sums with non-trivial ZZ, on both y signs and on infinity"""
    rnd = random.Random(4848)
    out = ctypes.create_string_buffer(48)
    pts = [bls.g1_mul(bls.G1_GEN, rnd.randrange(1, R)) for _ in range(12)]
    enc = [bls.g1_compress(p) for p in pts]
    for k in range(1, 12):
        assert hm.hm_g1_sum_compress28(out, b"".join(enc[:k]), k) == 1
        want = None
        for p in pts[:k]:
            want = bls.g1_add(want, p)
        assert out.raw == bls.g1_compress(want)
        neg = bls.g1_compress(bls.g1_neg(want))
        assert hm.hm_g1_sum_compress28(out, neg, 1) == 1 and out.raw == neg
    assert hm.hm_g1_sum_compress28(out, enc[0] + bls.g1_compress(bls.g1_neg(pts[0])), 2) == 1 and out.raw == bls.g1_compress(None)
    assert hm.hm_f28_violations() == 0


def test_affine_output_is_the_blst_p1_affine_image(hm):
    """the 96-byte point form of the *_affine entry points (x || y, 2^384-Montgomery residues as little-endian limbs,
    infinity = zeros) equals the oracle's affine point; the 48-byte encoding produced beside it is unchanged"""
    rnd = random.Random(96)
    out96 = ctypes.create_string_buffer(96)
    out48 = ctypes.create_string_buffer(48)
    pts = [bls.g1_mul(bls.G1_GEN, rnd.randrange(1, R)) for _ in range(5)]
    enc = [bls.g1_compress(p) for p in pts]
    want = None
    for k in range(1, 6):
        want = bls.g1_add(want, pts[k - 1])
        assert hm.hm_g1_sum_affine96(out96, out48, b"".join(enc[:k]), k) == 0
        x, y = want
        assert out96.raw == (x * (1 << 384) % P).to_bytes(48, "little") + (y * (1 << 384) % P).to_bytes(48, "little")
        assert out48.raw == bls.g1_compress(want)
    assert hm.hm_g1_sum_affine96(out96, out48, enc[0] + bls.g1_compress(bls.g1_neg(pts[0])), 2) == 0
    assert out96.raw == bytes(96) and out48.raw == bls.g1_compress(None)


def test_safegcd_inversion(hm):
    """modinv30 (Bernstein-Yang divsteps on signed 30-bit limbs) against pow(a, -1, m) for Fp and Fr: edge values,
    values with long runs of zero bits, random values; 0 -> 0"""
    rnd = random.Random(30)
    for which, m, nb in ((0, P, 48), (1, R, 32)):
        out = ctypes.create_string_buffer(nb)
        vals = [1, 2, 3, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, 1 << 30, (1 << 30) - 1, 1 << 60, (1 << (m.bit_length() - 1)), (1 << 200) + 1,
                m - (1 << 100), 0x5555555555555555555555555555555555555555 % m]
        vals += [rnd.randrange(1, m) for _ in range(400)]
        vals += [rnd.randrange(1, 1 << k) for k in (1, 5, 29, 30, 31, 59, 60, 61, 90, 128) for _ in range(5)]
        for a in vals:
            hm.hm_modinv30(which, out, int(a).to_bytes(nb, "little"))
            got = int.from_bytes(out.raw, "little")
            assert got == pow(a, -1, m), (which, hex(a))
        hm.hm_modinv30(which, out, bytes(nb))
        assert int.from_bytes(out.raw, "little") == 0


def test_single_item_host_lincomb(hm):
    """host_lincomb.hpp (the n = 1 ending of verification: commitment - [y]G + [z]proof as one double-scalar multiplication on the
    host, points taken from the decoder's 2^392-domain output) against the oracle's group arithmetic: random scalars and points,
    the zero scalars, scalars r - 1, a proof at infinity (constant polynomials), a commitment at infinity, and the cancellation
    [z]P = [y]G"""
    rnd = random.Random(0x51A61E)
    out = ctypes.create_string_buffer(48)

    def check(k1, pt, k2, com):
        want = bls.g1_add(bls.g1_add(bls.g1_mul(pt, k1) if pt is not None else None, bls.g1_neg(bls.g1_mul(bls.G1_GEN, k2))), com)
        rc = hm.hm_single_item_lincomb(out, k1.to_bytes(32, "big"), bls.g1_compress(pt), k2.to_bytes(32, "big"), bls.g1_compress(com))
        assert rc == 0 and out.raw == bls.g1_compress(want), (k1, k2)

    pts = [bls.g1_mul(bls.G1_GEN, rnd.randrange(1, R)) for _ in range(3)]
    for _ in range(6):
        check(rnd.randrange(R), rnd.choice(pts), rnd.randrange(R), rnd.choice(pts))
    check(0, pts[0], 0, pts[1])
    check(R - 1, pts[0], R - 1, pts[1])
    check(rnd.randrange(R), None, rnd.randrange(R), pts[2])       # proof = infinity
    check(rnd.randrange(R), pts[1], rnd.randrange(R), None)       # commitment = infinity
    check(1, bls.G1_GEN, 1, None)                                  # [1]G - [1]G: the identity
    check(0xF, bls.G1_GEN, 0xF0, bls.g1_mul(bls.G1_GEN, 0xE1))     # every step of one window adds a point and its negative
