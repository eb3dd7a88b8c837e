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
