"""Host-side mirror types of the reference's public API that need no GPU: `Blob<4096>` (src/blob.rs:18-76) and the
point type the producers return (`Commitment = Proof = P1`, src/kzg/mod.rs:9-10) -- checked against the oracle."""
import hashlib
import random

import pytest

import kateth_amd
from oracle.pyref import blob as oblob
from oracle.pyref import bls

R = bls.R


def test_blob_from_slice_to_bytes_and_errors():
    raw = b"".join(((i * 0x9E3779B97F4A7C15) % R).to_bytes(32, "big") for i in range(4096))
    b = kateth_amd.Blob.from_slice(raw)
    assert b.to_bytes() == raw and len(b) == kateth_amd.Blob.BYTES == 131072
    assert oblob.from_slice(raw) == [int.from_bytes(raw[i:i + 32], "big") for i in range(0, 131072, 32)]
    for ln in (0, 131071, 131073):
        with pytest.raises(kateth_amd.BlobError) as e:
            kateth_amd.Blob.from_slice(bytes(ln))
        assert e.value.kind == "InvalidLen"
    for bad in (R, R + 1, (1 << 256) - 1):
        data = bytearray(raw)
        data[32 * 4095:] = bad.to_bytes(32, "big")
        with pytest.raises(kateth_amd.BlobError) as e:
            kateth_amd.Blob.from_slice(bytes(data))
        assert e.value.kind == "InvalidFieldElement"
        with pytest.raises(oblob.BlobError):
            oblob.from_slice(bytes(data))
    data = bytearray(raw)
    data[0:32] = (R - 1).to_bytes(32, "big")  # the largest valid element
    kateth_amd.Blob.from_slice(bytes(data))


def test_blob_random_is_hash_to_of_512_random_bytes():
    """src/blob.rs:66-76: element i = Fr::hash_to(512 bytes drawn from the generator) = SHA-256(...) mod r"""
    b = kateth_amd.Blob.random(random.Random(4844))
    ref = random.Random(4844)
    raw = b.to_bytes()
    for i in range(0, 64):
        want = bls.fr_hash_to(ref.randbytes(512))
        assert int.from_bytes(raw[32 * i:32 * i + 32], "big") == want
    kateth_amd.Blob.from_slice(raw)  # every element canonical
    assert hashlib.sha256(raw).digest() != hashlib.sha256(kateth_amd.Blob.random(random.Random(1)).to_bytes()).digest()


def test_p1_image_compresses_like_the_oracle():
    rnd = random.Random(11)
    for _ in range(8):
        pt = bls.g1_mul(bls.G1_GEN, rnd.randrange(1, R))
        for q in (pt, bls.g1_neg(pt)):
            x, y = q
            img = (x * (1 << 384) % bls.P).to_bytes(48, "little") + (y * (1 << 384) % bls.P).to_bytes(48, "little")
            assert kateth_amd.P1(img).compress() == bls.g1_compress(q)
    assert kateth_amd.P1(bytes(96)).compress() == bls.g1_compress(None) and kateth_amd.P1(bytes(96)).is_inf()


def test_batch_verification_names_the_first_failing_blob(oracle_setup):
    """src/kzg/setup.rs:259-262: the blobs are parsed in index order and `collect` stops at the FIRST one that fails -- so
    (blob 0 non-canonical, blob 1 short) is InvalidFieldElement, (blob 0 fine, blob 1 short) InvalidLen, (blob 0 short, blob 1
    non-canonical) InvalidLen.  A short blob never reaches the engine (the C ABI takes n full blobs): the host mirror decides, and
    must decide like the reference (the oracle restates it).  No GPU needed: the decision falls before any library call."""
    from oracle.pyref.setup import KzgError as OracleKzgError

    good = (7).to_bytes(32, "big") * 4096
    noncanonical = R.to_bytes(32, "big") + good[32:]
    short = good[:-1]
    c = bls.g1_compress(bls.G1_GEN)
    mirror = object.__new__(kateth_amd.Setup)  # no context: every case below is decided on the host
    mirror._h, mirror._lib = None, None

    def kind_of(fn):
        try:
            fn()
        except (kateth_amd.KzgError, OracleKzgError) as err:
            inner = err
            while hasattr(inner, "inner"):
                inner = inner.inner
            return getattr(inner, "kind", None) or str(inner)
        return "no error"

    for blobs, want in (([noncanonical, short], "InvalidFieldElement"), ([good, short], "InvalidLen"), ([short, noncanonical], "InvalidLen"),
                        ([good, noncanonical, short], "InvalidFieldElement"), ([short], "InvalidLen")):
        cs, ps = [c] * len(blobs), [c] * len(blobs)
        assert want in kind_of(lambda: mirror.verify_blob_proof_batch(blobs, cs, ps)), (want, len(blobs))
        assert want in kind_of(lambda: oracle_setup.verify_blob_proof_batch(blobs, cs, ps)), (want, len(blobs))
