"""The C port of the reference CPU path (oracle/cport) checked against the
Python oracle and the committed golden vectors, so that the cpu_baseline that
bench.py times is a correct implementation of the same function."""
import ctypes
import json
import os
import random

import pytest

from oracle.cport import binding
from oracle.pyref import bls, synth

from conftest import GOLDEN, TRUSTED_SETUP

P, R = bls.P, bls.R


@pytest.fixture(scope="module")
def lib():
    return binding.load()


@pytest.fixture(scope="module")
def csetup(lib):
    s = binding.CSetup(lib, TRUSTED_SETUP, subgroup_checks=False, threads=2)
    yield s
    s.close()


def test_field_primitives(lib):
    rnd = random.Random(2)
    out48 = ctypes.create_string_buffer(48)
    out32 = ctypes.create_string_buffer(32)
    for _ in range(50):
        a, b = rnd.randrange(P), rnd.randrange(P)
        lib.cport_fp_mul_plain(out48, a.to_bytes(48, "big"), b.to_bytes(48, "big"))
        assert int.from_bytes(out48.raw, "big") == a * b % P
        x, y = rnd.randrange(R), rnd.randrange(R)
        lib.cport_fr_mul_plain(out32, x.to_bytes(32, "big"), y.to_bytes(32, "big"))
        assert int.from_bytes(out32.raw, "big") == x * y % R
    for x in (1, 2, R - 1, rnd.randrange(R)):
        lib.cport_fr_inv_plain(out32, x.to_bytes(32, "big"))
        assert int.from_bytes(out32.raw, "big") == pow(x, -1, R)
    for a, b in ((P - 1, P - 1), (0, 5), (1, P - 1)):
        lib.cport_fp_mul_plain(out48, a.to_bytes(48, "big"), b.to_bytes(48, "big"))
        assert int.from_bytes(out48.raw, "big") == a * b % P


def test_g1_mul_and_decompress(lib):
    out = ctypes.create_string_buffer(48)
    g = bls.g1_compress(bls.G1_GEN)
    for k in (1, 2, 3, R - 1, 0xDEADBEEFCAFEBABE123456789):
        assert lib.cport_g1_mul(out, g, k.to_bytes(32, "big")) == 0
        assert out.raw == bls.g1_compress(bls.g1_mul(bls.G1_GEN, k))
    assert lib.cport_g1_decompress_status(g) == 0
    assert lib.cport_g1_decompress_status(bytes([g[0] & 0x7F]) + g[1:]) == 3
    assert lib.cport_g1_decompress_status(bytes([0xC0]) + bytes(47)) == 0


def test_commitment_known_answers_and_golden(csetup):
    gen = bls.g1_compress(bls.G1_GEN)
    st, out = csetup.blob_to_commitment((1).to_bytes(32, "big") * 4096)
    assert st == 0 and out == gen
    st, out = csetup.blob_to_commitment(bytes(131072))
    assert st == 0 and out == bytes([0xC0]) + bytes(47)
    blob = bytearray(131072)
    blob[0:32] = R.to_bytes(32, "big")
    st, out = csetup.blob_to_commitment(bytes(blob))
    assert st == 2
    golden = json.load(open(os.path.join(GOLDEN, "kzg_vectors.json")))
    for rec in golden["blobs"][:3]:
        st, out = csetup.blob_to_commitment(synth.blob_bytes(golden["seed"], rec["index"]))
        assert st == 0 and out.hex() == rec["commitment"]


def test_threaded_equals_single(csetup):
    blob = synth.blob_bytes(0x99, 3)
    csetup.set_threads(1)
    _, a = csetup.blob_to_commitment(blob)
    csetup.set_threads(4)
    _, b = csetup.blob_to_commitment(blob)
    assert a == b


def test_verify_prepairing_matches_oracle_and_pairs(lib, csetup, oracle_setup):
    """the C port's reference-literal verify path: z, y equal the golden vectors; its two lincomb points
    satisfy the pairing equation (checked with the Python oracle's pairing); a swapped proof does not."""
    import ctypes as ct

    golden = json.load(open(os.path.join(GOLDEN, "kzg_vectors.json")))
    recs = golden["blobs"][:3]
    blobs = b"".join(synth.blob_bytes(golden["seed"], r["index"]) for r in recs)
    cs = b"".join(bytes.fromhex(r["commitment"]) for r in recs)
    ps = b"".join(bytes.fromhex(r["proof"]) for r in recs)
    for batch_inverse in (False, True):
        rc, z, y, a48, b48 = csetup.verify_batch_prepairing(blobs, cs, ps, 3, batch_inverse)
        assert rc == 0
        for k, r in enumerate(recs):
            assert z[32 * k:32 * k + 32].hex() == r["challenge_z"]
            assert y[32 * k:32 * k + 32].hex() == r["eval_y"]
        A, B = bls.g1_decompress(a48), bls.g1_decompress(b48)
        assert bls.verify_pairings((A, oracle_setup.g2_monomial[1]), (B, bls.G2_GEN))
    rc, z, y, a48, b48 = csetup.verify_batch_prepairing(blobs, cs, ps[48:96] + ps[:48] + ps[96:], 3, False)
    assert rc == 0
    assert not bls.verify_pairings((bls.g1_decompress(a48), oracle_setup.g2_monomial[1]), (bls.g1_decompress(b48), bls.G2_GEN))
    # errors in the reference's order
    bad = bytearray(blobs)
    bad[0:32] = R.to_bytes(32, "big")
    assert csetup.verify_batch_prepairing(bytes(bad), bytes([cs[0] & 0x7F]) + cs[1:], ps, 3)[0] == 2
    assert csetup.verify_batch_prepairing(blobs, bytes([cs[0] & 0x7F]) + cs[1:], ps, 3)[0] == 3
    # primitives
    out = ct.create_string_buffer(32)
    rnd = random.Random(8)
    for x in (1, 2, R - 1, rnd.randrange(R), rnd.randrange(R)):
        lib.cport_fr_eucl_inv_plain(out, x.to_bytes(32, "big"))
        assert int.from_bytes(out.raw, "big") == pow(x, -1, R)
    import hashlib

    for ln in (0, 55, 56, 64, 1000):
        msg = bytes(rnd.randrange(256) for _ in range(ln))
        lib.cport_sha256(out, msg, ln)
        assert out.raw == hashlib.sha256(msg).digest()
