"""Oracle restatement of kateth `src/blob.rs`."""
from . import bls
from .bls import FiniteFieldError

FIELD_ELEMENTS_PER_BLOB = 4096
BYTES_PER_BLOB = 32 * FIELD_ELEMENTS_PER_BLOB


class BlobError(Exception):
    """`blob::Error` (src/blob.rs:6-10). kind in {InvalidFieldElement, InvalidLen}."""

    def __init__(self, kind: str):
        super().__init__(kind)
        self.kind = kind


def from_slice(data: bytes, n: int = FIELD_ELEMENTS_PER_BLOB):
    """`Blob::from_slice` (src/blob.rs:26-37)."""
    if len(data) != 32 * n:
        raise BlobError("InvalidLen")
    out = []
    for i in range(n):
        try:
            out.append(bls.fr_from_be_slice(data[32 * i:32 * i + 32]))
        except FiniteFieldError:
            raise BlobError("InvalidFieldElement")  # src/blob.rs:12-16
    return out


def to_bytes(elements) -> bytes:
    """`Blob::to_bytes` (src/blob.rs:39-46)."""
    return b"".join(bls.fr_to_be_bytes(e) for e in elements)


def commitment(elements, setup):
    """`Blob::commitment` (src/blob.rs:48-53)."""
    return bls.g1_lincomb_pippenger(setup.g1_lagrange_brp, elements)


def challenge(elements, commitment_pt) -> int:
    """`Blob::challenge` (src/blob.rs:78-97)."""
    n = len(elements)
    data = b"FSBLOBVERIFY_V1_" + n.to_bytes(16, "big") + to_bytes(elements) + bls.g1_compress(commitment_pt)
    return bls.fr_hash_to(data)


def proof(elements, commitment_pt, setup):
    """`Blob::proof` (src/blob.rs:55-64)."""
    from . import poly

    z = challenge(elements, commitment_pt)
    _, pi = poly.prove(elements, z, setup)
    return pi
