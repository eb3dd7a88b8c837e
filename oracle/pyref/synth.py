"""Seeded synthetic blobs.

The reference draws each element as SHA-256(512 random bytes) mod r from an
unseeded RNG (`Blob::random`, src/blob.rs:66-76) -- uniform canonical field
elements.  The build's counterpart is counter based so CPU and GPU can produce
the same blob:  element(b, i) = SHA-256(seed_le64 || b_le64 || i_le32) mod r.
"""
import hashlib
import struct

from .bls import R

DEFAULT_SEED = 0x4844


def element(seed: int, b: int, i: int) -> int:
    return int.from_bytes(hashlib.sha256(struct.pack("<QQI", seed, b, i)).digest(), "big") % R


def blob_bytes(seed: int, b: int, n: int = 4096) -> bytes:
    return b"".join(element(seed, b, i).to_bytes(32, "big") for i in range(n))
