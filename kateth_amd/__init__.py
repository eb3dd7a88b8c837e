"""kateth_amd -- MI355X-native EIP-4844 KZG engine (host-side mirror).

This package mirrors kateth's public Rust API (`src/lib.rs:5-7`,
`src/kzg/mod.rs:9-35`, `src/kzg/setup.rs:37-275`, `src/blob.rs:6-46`) over the
C ABI in `include/kateth_amd.h`.  All arithmetic happens in the HIP library
`libkateth_amd.so`; there is no Python or CPU compute path, and importing the
engine on a machine without the built library or without a GPU fails loudly.
"""
from .kzg import (  # noqa: F401
    BYTES_PER_BLOB,
    P1,
    Blob,
    BlobError,
    BlsError,
    ECGroupError,
    FiniteFieldError,
    KzgError,
    LoadSetupError,
    Setup,
    library_path,
)

__all__ = [
    "Setup",
    "Blob",
    "P1",
    "BlobError",
    "BlsError",
    "ECGroupError",
    "FiniteFieldError",
    "KzgError",
    "LoadSetupError",
    "BYTES_PER_BLOB",
    "library_path",
]
