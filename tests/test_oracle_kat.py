"""Pins the CPU oracle (oracle/pyref) with every oracle-free known answer
available for this path plus identities over
the ceremony file the reference's own tests load (src/kzg/setup.rs:299-303).
The reference's golden vectors (consensus-spec-tests) are an empty submodule in
this checkout, hence "parity unpinned" in the oracle header."""
import hashlib
import json

import pytest

from oracle.pyref import blob as oblob
from oracle.pyref import bls, domain, poly
from oracle.pyref.setup import KzgError, Setup

from conftest import TRUSTED_SETUP

R = bls.R
P = bls.P
GEN48 = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
)
INF48 = bytes([0xC0]) + bytes(47)


def be32(v):
    return int(v).to_bytes(32, "big")


# ---- reference unit tests restated (src/bls.rs:604-612, src/math.rs:80-113) ---
def test_fr_one_and_max_constants():
    # Fr::ONE / Fr::MAX limbs are 1 and r-1 in Montgomery form (R = 2^256)
    one = [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]
    mx = [18446744060824649731, 18102478225614246908, 11073656695919314959, 6613806504683796440]
    val = lambda l: sum(x << (64 * i) for i, x in enumerate(l))
    assert val(one) == (1 << 256) % R
    assert val(mx) == (R - 1) * (1 << 256) % R
    assert (R - 1 + 1) % R == 0


def test_bit_reversal_is_involution_and_rejects_non_pow2():
    import random

    rnd = random.Random(1)
    xs = [rnd.randrange(1 << 16) for _ in range(4096)]
    assert domain.bit_reversal_permutation(domain.bit_reversal_permutation(xs)) == xs
    with pytest.raises(AssertionError):
        domain.bit_reversal_permutation([0] * 4095)


def test_primitive_root_of_unity():
    w = domain.primitive_root_of_unity(4096)
    assert w == 0x564C0A11A0F704F4FC3E8ACFE0F8245F0AD1347B378FBF96E206DA11A5D36306
    assert w * pow(w, 4095, R) % R == 1
    assert pow(w, 2048, R) == R - 1
    brp = domain.bit_reversal_permutation(domain.roots_of_unity(4096))
    assert brp[0] == 1 and brp[1] == R - 1
    assert brp[2] == 0x8D51CCCE760304D0EC030002760300000001000000000000
    digest = hashlib.sha256(b"".join(be32(x) for x in brp)).hexdigest()
    assert digest == "1d815dd2fcaae4382dad24b89c046c2ece81a6554455ce76eaf754b285ed0792"


def test_fr_pow_reference_quirk_q2():
    assert bls.fr_pow_reference(5, 0) == 5  # src/bls.rs:169-187 returns x for power 0
    assert bls.fr_pow_reference(5, 1) == 5
    for e in (2, 3, 7, 4096, 65537):
        assert bls.fr_pow_reference(5, e) == pow(5, e, R)


# ---- fixture file -----------------------------------------------------------
def test_trusted_setup_digests():
    raw = open(TRUSTED_SETUP, "rb").read()
    assert hashlib.sha256(raw).hexdigest() == "0229b43f4fac9b17374809520eb621b5ee1a7f74547e7d36918e7d4b122e178d"
    d = json.loads(raw)
    g1 = b"".join(bytes.fromhex(s[2:]) for s in d["g1_lagrange"])
    assert hashlib.sha256(g1).hexdigest() == "52c7615a9bd3eb20df67eb5a81ee701c96787c82a5ff638740b54fbadfde960b"
    assert bytes.fromhex(d["g2_monomial"][0][2:]) == bls.g2_compress(bls.G2_GEN)


def test_generators_and_encoding_roundtrip():
    assert bls.g1_is_on_curve(bls.G1_GEN) and bls.g1_in_subgroup(bls.G1_GEN)
    assert bls.g2_is_on_curve(bls.G2_GEN) and bls.g2_in_subgroup(bls.G2_GEN)
    assert bls.g1_compress(bls.G1_GEN) == GEN48
    assert bls.g1_decompress(GEN48) == bls.G1_GEN
    assert bls.g1_compress(None) == INF48 and bls.g1_decompress(INF48) is None
    q = bls.g1_mul(bls.G1_GEN, 0xDEADBEEF)
    assert bls.g1_decompress(bls.g1_compress(q)) == q
    nq = bls.g1_neg(q)
    assert bls.g1_decompress(bls.g1_compress(nq)) == nq
    assert bls.g1_compress(q)[0] & 0x20 != bls.g1_compress(nq)[0] & 0x20
    h = bls.g2_mul(bls.G2_GEN, 0xC0FFEE)
    assert bls.g2_decompress(bls.g2_compress(h)) == h


def test_setup_sample_subgroup_and_sum_is_generator(oracle_setup):
    pts = oracle_setup.g1_lagrange_brp
    for i in (0, 1, 2, 3, 1234, 4095):
        assert bls.g1_in_subgroup(pts[i])
    for q in oracle_setup.g2_monomial[:3]:
        assert bls.g2_is_on_curve(q) and bls.g2_in_subgroup(q)
    # sum_i L_i(tau) = 1  =>  sum of all Lagrange points = G1 generator (8c item 2)
    acc = (1, 1, 0)
    for pt in pts:
        acc = bls._jac_add(acc, bls._jac_from_affine(pt))
    assert bls._jac_to_affine(acc) == bls.G1_GEN


def test_pairing_bilinear_and_ceremony_consistency(oracle_setup):
    a, b = 0x1234567, 0x89ABCDEF01
    assert bls.verify_pairings(
        (bls.g1_mul(bls.G1_GEN, a * b), bls.G2_GEN), (bls.g1_mul(bls.G1_GEN, a), bls.g2_mul(bls.G2_GEN, b))
    )
    assert not bls.verify_pairings(
        (bls.g1_mul(bls.G1_GEN, a * b + 1), bls.G2_GEN), (bls.g1_mul(bls.G1_GEN, a), bls.g2_mul(bls.G2_GEN, b))
    )
    # [tau]_1 = commitment to p(x) = x (blob e_i = omega_brp[i]); the ceremony's
    # G2 side must agree:  e([tau]_1, G2) == e(G1, [tau]_2).  This ties the MSM,
    # the BRP convention, the roots of unity and the pairing to independent data.
    tau1 = oblob.commitment(oracle_setup.roots_of_unity_brp, oracle_setup)
    assert bls.verify_pairings((tau1, bls.G2_GEN), (bls.G1_GEN, oracle_setup.g2_monomial[1]))
    assert not bls.verify_pairings((tau1, bls.G2_GEN), (bls.G1_GEN, oracle_setup.g2_monomial[2]))


# ---- SURVEY 8(c) items 2-6 ---------------------------------------------------
# Public constants the oracle did not produce: the compressed G1 points [2]G and [3]G are the Ethereum consensus-layer BLS
# public keys of the secret keys 2 and 3 (every eth2 BLS test suite carries them); -G is G with the sign bit flipped.
G2X48 = bytes.fromhex("a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e")
G3X48 = bytes.fromhex("89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224")


def test_small_multiples_of_the_generator_match_public_constants(oracle_setup):
    """[2]G and [3]G against the well-known public keys of secret keys 2 and 3 -- pins doubling, addition and the encoding to
    data outside this repository; and, through the Lagrange basis summing to one, the commitments of the constant blobs 2 and 3
    (SURVEY.md section 8(c) item 2 extended): commit(all elements = c) = [c]G"""
    assert bls.g1_compress(bls.g1_mul(bls.G1_GEN, 2)) == G2X48
    assert bls.g1_compress(bls.g1_add(bls.g1_mul(bls.G1_GEN, 2), bls.G1_GEN)) == G3X48
    assert bls.g1_compress(bls.g1_neg(bls.G1_GEN)) == bytes([GEN48[0] ^ 0x20]) + GEN48[1:]
    assert bls.g1_compress(oracle_setup.blob_to_commitment((2).to_bytes(32, "big") * 4096)) == G2X48
    assert bls.g1_compress(oracle_setup.blob_to_commitment((3).to_bytes(32, "big") * 4096)) == G3X48


def test_commitment_known_answers(oracle_setup):
    d = json.load(open(TRUSTED_SETUP))
    ones = be32(1) * 4096
    assert bls.g1_compress(oracle_setup.blob_to_commitment(ones)) == GEN48
    assert bls.g1_compress(oracle_setup.blob_to_commitment(bytes(131072))) == INF48
    for i in (0, 1, 2, 3, 4095):
        blob = bytearray(131072)
        blob[32 * i + 31] = 1
        want = bytes.fromhex(d["g1_lagrange"][domain.bit_reversal_permutation_index(i, 4096)][2:])
        assert bls.g1_compress(oracle_setup.blob_to_commitment(bytes(blob))) == want


def test_pippenger_matches_naive_sum(oracle_setup):
    import random

    rnd = random.Random(7)
    pts = oracle_setup.g1_lagrange_brp[:24]
    sc = [rnd.randrange(R) for _ in pts]
    assert bls.g1_lincomb_pippenger(pts, sc) == bls.g1_lincomb(pts, sc)


def test_challenge_known_answers():
    z0 = oblob.challenge([0] * 4096, None)
    assert z0 == 0x04B7B22AF63D2B2F1CED8D550560E5D1E4B01E355903DEE22781E87826856096
    z1 = oblob.challenge([1] * 4096, bls.G1_GEN)
    assert z1 == 0x1240EE945BA588D3E81CE99DC1395E712C2C230DAEDAC7276EB31A371F17B564


def test_constant_and_identity_polynomials(oracle_setup):
    z = 0x1234567890ABCDEF
    c = 0x55AA
    y, pi = poly.prove([c] * 4096, z, oracle_setup)
    assert y == c and pi is None  # constant blob: proof = infinity
    roots = oracle_setup.roots_of_unity_brp
    assert poly.evaluate(roots, z, oracle_setup) == z  # p(x) = x
    assert poly.evaluate(roots, roots[5], oracle_setup) == roots[5]  # in-domain shortcut


# ---- rejection cases -------------------------------------
def test_rejections(oracle_setup):
    blob = bytearray(131072)
    blob[0:32] = be32(R)
    with pytest.raises(oblob.BlobError) as e:
        oracle_setup.blob_to_commitment(bytes(blob))
    assert e.value.kind == "InvalidFieldElement"
    for ln in (131071, 131073, 0):
        with pytest.raises(oblob.BlobError) as e:
            oracle_setup.blob_to_commitment(bytes(ln))
        assert e.value.kind == "InvalidLen"
    with pytest.raises(bls.ECGroupError) as e:  # compressed bit clear
        bls.g1_decompress(bytes([GEN48[0] & 0x7F]) + GEN48[1:])
    assert e.value.kind == "InvalidEncoding"
    with pytest.raises(bls.ECGroupError) as e:  # x >= p
        bls.g1_decompress(bytes([0x80 | 0x1A]) + bytes([0xFF] * 47))
    assert e.value.kind == "InvalidEncoding"
    with pytest.raises(bls.ECGroupError) as e:  # infinity with the sign bit set
        bls.g1_decompress(bytes([0xE0]) + bytes(47))
    assert e.value.kind == "InvalidEncoding"
    # find an x with no square root -> NotOnCurve, and an on-curve point outside G1
    x = 1
    while bls._fp_sqrt(x**3 + 4) is not None:
        x += 1
    with pytest.raises(bls.ECGroupError) as e:
        bls.g1_decompress(bytes([0x80]) + x.to_bytes(48, "big")[1:])
    assert e.value.kind == "NotOnCurve"
    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    with pytest.raises(bls.ECGroupError) as e:
        bls.g1_decompress(bls.g1_compress((x, y)))
    assert e.value.kind == "NotInGroup"


# ---- algebraic closure ------------------------------------
def test_prove_verify_closure_small_batch(oracle_setup):
    from oracle.pyref import synth

    blobs = [synth.blob_bytes(synth.DEFAULT_SEED, b) for b in range(2)]
    cs = [bls.g1_compress(oracle_setup.blob_to_commitment(b)) for b in blobs]
    ps = [bls.g1_compress(oracle_setup.blob_proof(b, c)) for b, c in zip(blobs, cs)]
    assert oracle_setup.verify_blob_proof(blobs[0], cs[0], ps[0])
    assert oracle_setup.verify_blob_proof_batch(blobs, cs, ps)
    assert not oracle_setup.verify_blob_proof_batch(blobs, cs, [ps[1], ps[0]])
    assert not oracle_setup.verify_blob_proof(blobs[0], cs[0], ps[1])
    assert oracle_setup.verify_blob_proof_batch([], [], [])  # pairing of two infinities == 1
    bad = bytearray(blobs[0])
    bad[31] ^= 1
    assert not oracle_setup.verify_blob_proof(bytes(bad), cs[0], ps[0])
    with pytest.raises(KzgError):
        oracle_setup.verify_blob_proof_batch([bytes(5)], cs[:1], ps[:1])


# ---- ceremony pins beyond k = 1 -------------------------------------------------------------
# The trusted setup's 65 G2 monomial points [tau^k]_2 are data the oracle did not produce.  commit(x^k) must be
# [tau^k]_1, so  e(commit(x^k), G2) == e(G1, [tau^k]_2)  ties the oracle's MSM, BRP convention, roots of unity, G2
# decoding and pairing to all of them.
def _monomial_blob(k, roots):
    return [pow(w, k, R) for w in roots]


def test_ceremony_identity_for_higher_monomials(oracle_setup):
    roots = oracle_setup.roots_of_unity_brp
    for k in (2, 3, 17, 64):
        ck = oblob.commitment(_monomial_blob(k, roots), oracle_setup)
        assert bls.verify_pairings((ck, bls.G2_GEN), (bls.G1_GEN, oracle_setup.g2_monomial[k])), k
        assert not bls.verify_pairings((ck, bls.G2_GEN), (bls.G1_GEN, oracle_setup.g2_monomial[k - 1])), k


def test_ceremony_identity_random_combination_of_all_65_monomials(oracle_setup):
    """one commitment against ALL 65 G2 points: p(x) = sum_k rho_k x^k with pseudo-random rho_k;
    e(commit(p), G2) == e(G1, sum_k rho_k [tau^k]_2).  A wrong [tau^k] relation for any k would break it."""
    import random

    rnd = random.Random(65)
    roots = oracle_setup.roots_of_unity_brp
    rho = [rnd.randrange(1, R) for _ in range(65)]
    evals = []
    for w in roots:
        acc = 0
        for c in reversed(rho):  # Horner
            acc = (acc * w + c) % R
        evals.append(acc)
    cp = oblob.commitment(evals, oracle_setup)
    q = None
    for k, c in enumerate(rho):
        q = bls.g2_add(q, bls.g2_mul(oracle_setup.g2_monomial[k], c))
    assert bls.verify_pairings((cp, bls.G2_GEN), (bls.G1_GEN, q))
    q2 = bls.g2_add(q, oracle_setup.g2_monomial[40])  # coefficient 40 off by one
    assert not bls.verify_pairings((cp, bls.G2_GEN), (bls.G1_GEN, q2))


def test_prove_on_monomials_matches_closed_form(oracle_setup):
    """p(x) = x^k:  y = z^k  and the quotient is sum_{j<k} z^(k-1-j) x^j, so the proof is
    sum_j z^(k-1-j) [tau^j]_1 with [tau^j]_1 = commit(x^j) (pinned to the ceremony's G2 side above).
    Off-domain z and an in-domain z (the reference's special branch, src/kzg/poly.rs:50-64)."""
    roots = oracle_setup.roots_of_unity_brp
    k = 5
    taus = [oblob.commitment(_monomial_blob(j, roots), oracle_setup) for j in range(k)]
    assert taus[0] == bls.G1_GEN
    for z in (0x1234567890ABCDEF, roots[77]):
        y, pi = poly.prove(_monomial_blob(k, roots), z, oracle_setup)
        assert y == pow(z, k, R)
        want = None
        for j in range(k):
            want = bls.g1_add(want, bls.g1_mul(taus[j], pow(z, k - 1 - j, R)))
        assert pi == want


# ---- the one EXTERNAL vector: public EIP-4844 point-evaluation precompile test (tests/golden/external-vectors/README.md) ------
EXT_COMMITMENT = bytes.fromhex("8f59a8d2a1a625a17f3fea0fe5eb8c896db3764f3185481bc22f91b4aaffcca25f26936857bc3a7c2539ea8ec3a952b7")
EXT_Z = bytes.fromhex("564c0a11a0f704f4fc3e8acfe0f8245f0ad1347b378fbf96e206da11a5d36306")
EXT_Y = bytes.fromhex("24d25032e67a7e6a4910df5834b8fe70e6bcfeeac0352434196bdf4b2485d5a1")
EXT_PROOF = bytes.fromhex("873033e038326e87ed3e1276fd140253fa08e9fc25fb2d9a98527fc22a2c9612fbeafdad446cbc7bcdbdcd780af2c16a")
EXT_VERSIONED_HASH = bytes.fromhex("01e798154708fe7789429634053cbf9f99b619f9f084048927333fce637f549b")


def test_external_point_evaluation_precompile_vector(oracle_setup):
    """verify_proof_inner (src/kzg/setup.rs:84-94) on data an independent implementation produced: the vector authenticates
    itself (versioned hash = 0x01 || sha256(commitment)[1:]; the pairing equation holds against the ceremony's [tau]_2), z is the
    primitive 4096th root of unity -- a point ON the domain -- and the neighbours y + 1, z + 1 are rejected"""
    assert b"\x01" + hashlib.sha256(EXT_COMMITMENT).digest()[1:] == EXT_VERSIONED_HASH
    assert int.from_bytes(EXT_Z, "big") == domain.primitive_root_of_unity(4096) == pow(7, (R - 1) // 4096, R)
    assert oracle_setup.verify_proof(EXT_PROOF, EXT_COMMITMENT, EXT_Z, EXT_Y) is True
    assert oracle_setup.verify_proof(EXT_PROOF, EXT_COMMITMENT, EXT_Z, be32(int.from_bytes(EXT_Y, "big") + 1)) is False
    assert oracle_setup.verify_proof(EXT_PROOF, EXT_COMMITMENT, be32(int.from_bytes(EXT_Z, "big") + 1), EXT_Y) is False
    assert oracle_setup.verify_proof(EXT_COMMITMENT, EXT_PROOF, EXT_Z, EXT_Y) is False  # roles swapped
    # both points decode, are in the group and re-encode to the same bytes
    for enc in (EXT_COMMITMENT, EXT_PROOF):
        pt = bls.g1_uncompress(enc)
        assert bls.g1_in_subgroup(pt) and bls.g1_compress(pt) == enc
