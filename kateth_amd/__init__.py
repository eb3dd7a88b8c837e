"""kateth_amd -- MI355X-native EIP-4844 KZG engine (host-side mirror).

This package mirrors kateth's public Rust API (`src/lib.rs:5-7`,
`src/kzg/mod.rs:9-35`, `src/kzg/setup.rs:37-275`, `src/blob.rs:6-46`) over the
C ABI in `include/kateth_amd.h`.  All arithmetic happens in the HIP library
`libkateth_amd.so`; there is no Python or CPU compute path, and importing the
engine on a machine without the built library or without a GPU fails loudly.
"""
import os as _os

# The engine overlaps kernels on several HIP streams (point decoding beside evaluation, the two lincombs, copies beside
# compute, calls kept in flight by the caller); the runtime multiplexes ALL streams of the process onto GPU_MAX_HW_QUEUES
# hardware queues -- 4 by default -- and two streams that share a queue run strictly one after the other.  Takes effect when
# set before HIP initialises (the library's own load-time constructor does the same for non-Python callers).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .kzg import (  # noqa: F401,E402
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
