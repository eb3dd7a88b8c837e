"""Blob-sharded multi-GPU orchestration.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Blobs are independent, so ranks own contiguous
global index ranges and the data path has NO collective; the only exchanges are
  * commitments / proofs : one all-gather of 48 bytes per blob,
  * batch verification    : an all-gather of 32-byte transcript roots + the
    first-error records, then an all-gather of 192 bytes of partial sums.
torch is plumbing here (device memory, streams, process groups), not compute.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous range [first, first+count) of rank `rank`: ceil(n/world) blobs per rank."""
    per = (n_total + world - 1) // world
    first = min(n_total, rank * per)
    return first, max(0, min(per, n_total - first))


def merge_first_error(err6_by_rank: Sequence[Sequence[int]], first_index_by_rank: Sequence[int]) -> Tuple[int, int]:
    """Rebuilds the reference's first-error-wins order (src/kzg/setup.rs:259-271:
    every blob is parsed before any commitment, every commitment before any
    proof) from the per-rank records of kzg_verify_phase1_dev.
    Returns (code, global_index) or (0, -1)."""
    for kind in (0, 2, 4):  # blobs, commitments, proofs
        best = None
        for err6, first in zip(err6_by_rank, first_index_by_rank):
            if err6[kind] >= 0:
                g = first + err6[kind]
                if best is None or g < best[1]:
                    best = (err6[kind + 1], g)
        if best is not None:
            return best
    return 0, -1


def all_gather_bytes(local, world: int, group=None):
    """all-gather equal-sized uint8 tensors into one flat tensor ordered by rank."""
    import torch
    import torch.distributed as dist

    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def all_gather_host_bytes(payload: bytes, world: int, device, group=None) -> List[bytes]:
    """small fixed-size host records (roots, error records, partial points)."""
    import torch

    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    flat = all_gather_bytes(t, world, group).cpu().numpy().tobytes()
    k = len(payload)
    return [flat[i * k:(i + 1) * k] for i in range(world)]


def verify_blob_proof_batch_sharded(setup, d_blobs: int, d_commitments: int, d_proofs: int, n_local: int, first_index: int, n_total: int,
                                    rank: int, world: int, device, stream: int = 0, group=None) -> bool:
    """`Setup::verify_blob_proof_batch` over blobs sharded across `world` ranks.
    Every rank returns the same boolean (or raises the same KzgError)."""
    import struct

    from .kzg import _kzg_error

    sess, root, err6 = setup.verify_phase1_dev(d_blobs, d_commitments, d_proofs, n_local, stream)
    try:
        rec = root + struct.pack("<6iq", *err6, first_index)
        recs = all_gather_host_bytes(rec, world, device, group)
        roots = b"".join(r[:32] for r in recs)
        errs = [struct.unpack("<6iq", r[32:]) for r in recs]
        code, _ = merge_first_error([e[:6] for e in errs], [e[6] for e in errs])
        if code:
            raise _kzg_error(code)
        partial = setup.verify_phase2_dev(sess, roots, first_index, n_total)
    finally:
        setup.verify_session_destroy(sess)
    partials = all_gather_host_bytes(partial, world, device, group)
    return setup.verify_batch_finish(b"".join(partials))
