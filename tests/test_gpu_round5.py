"""GPU tests added in round 5 (through the C ABI, `-m gpu`):
  * the one EXTERNAL vector (tests/golden/external-vectors/README.md) through every way a single (commitment, z, y, proof)
    reaches the engine -- src/kzg/setup.rs:84-113;
  * device-resident sharded calls on a group context (kzg_*_group_dev) against the single-device calls;
  * the verification call's round-5 tail (device Horner, GLV split) against the round-4 path kept as cross-check."""
import os

import pytest

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, TRUSTED_SETUP  # noqa: E402
from test_oracle_kat import EXT_COMMITMENT, EXT_PROOF, EXT_Y, EXT_Z  # noqa: E402

INF48 = bytes([0xC0]) + bytes(47)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def be32(v):
    return int(v).to_bytes(32, "big")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def engine():
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    yield s
    s.close()


@pytest.fixture(scope="module")
def group2():
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0, 0])
    yield s
    s.close()


# ---------------------------------------------------------------------------------------------------------------------------
# external vector
# ---------------------------------------------------------------------------------------------------------------------------
def test_external_point_evaluation_vector(engine, group2, monkeypatch):
    """verify_kzg_proof on data an independent implementation produced (the public EIP-4844 point-evaluation precompile test):
    true; y + 1, z + 1 and swapped roles false -- through the host double-scalar ending of a single item, through the batch
    machinery (KATETH_AMD_SINGLE_VIA_BATCH: transcript, two variable-base MSMs), and through a group context; and both points
    through the public decoder with the oracle's affine coordinates"""
    import kateth_amd
    from oracle.pyref import bls

    monkeypatch.setenv("KATETH_AMD_SINGLE_VIA_BATCH", "1")
    batch = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    monkeypatch.delenv("KATETH_AMD_SINGLE_VIA_BATCH")
    try:
        y1 = be32((int.from_bytes(EXT_Y, "big") + 1) % R)
        z1 = be32((int.from_bytes(EXT_Z, "big") + 1) % R)
        for s in (engine, batch, group2):
            assert s.verify_proof(EXT_PROOF, EXT_COMMITMENT, EXT_Z, EXT_Y) is True
            assert s.verify_proof(EXT_PROOF, EXT_COMMITMENT, EXT_Z, y1) is False
            assert s.verify_proof(EXT_PROOF, EXT_COMMITMENT, z1, EXT_Y) is False
            assert s.verify_proof(EXT_COMMITMENT, EXT_PROOF, EXT_Z, EXT_Y) is False
            assert s.verify_proof(INF48, EXT_COMMITMENT, EXT_Z, EXT_Y) is False
        pts, st = engine.decompress_g1_batch([EXT_COMMITMENT, EXT_PROOF])
        assert st == [0, 0]
        for p1, enc in zip(pts, (EXT_COMMITMENT, EXT_PROOF)):
            assert p1.compress() == enc
            x, y = bls.g1_uncompress(enc)
            mont = lambda v: (v << 384) % bls.P  # noqa: E731 -- blst_p1_affine: 2^384-Montgomery, little-endian limbs
            assert p1.affine == mont(x).to_bytes(48, "little") + mont(y).to_bytes(48, "little")
    finally:
        batch.close()


# ---------------------------------------------------------------------------------------------------------------------------
# device-resident sharded calls on a group context (kzg_*_group_dev)
# ---------------------------------------------------------------------------------------------------------------------------
def _triples(engine, torch, n, seed):
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(seed, 0, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    return d_blobs, d_c, d_p


def _cut(t, counts, item_bytes):
    """contiguous member shares of a device tensor (views: every share is resident on the one card the members share)"""
    out, first = [], 0
    for c in counts:
        out.append(t[first * item_bytes:(first + c) * item_bytes])
        first += c
    return out


def _ptrs(parts, counts):
    return [p.data_ptr() if c else 0 for p, c in zip(parts, counts)]


@pytest.fixture(scope="module")
def group3():
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0, 0, 0])
    yield s
    s.close()


@pytest.mark.parametrize("counts", [(150, 150), (299, 1), (0, 300), (300, 0), (100, 120, 80), (7, 0, 293)])
def test_group_dev_commit_and_proof_match_the_single_device_calls(counts, engine, group2, group3, torch_cuda):
    """kzg_blob_to_commitment_batch_group_dev / kzg_compute_blob_proof_batch_group_dev: every member's share resident on its
    device (here: members on one card), enqueue-only; results and statuses byte for byte those of the single-device *_dev calls
    over the concatenated batch -- even, ragged and empty shares, a rejected blob in the last member's share"""
    torch = torch_cuda
    grp = group2 if len(counts) == 2 else group3
    n = sum(counts)
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0x6D0 + n)
    bad = n - 3  # a non-canonical element: InvalidFieldElement for that blob only (src/blob.rs:26-37)
    d_blobs[bad * 131072: bad * 131072 + 32] = 0xFF
    want_c = torch.empty_like(d_c)
    want_p = torch.empty_like(d_p)
    want_st = torch.empty(n, dtype=torch.int32, device="cuda")
    want_st2 = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, want_c.data_ptr(), want_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, want_p.data_ptr(), want_st2.data_ptr())
    torch.cuda.synchronize()
    assert want_st[bad].item() == 2 and int(want_st.abs().sum()) == 2
    got_c = torch.zeros_like(d_c)
    got_p = torch.zeros_like(d_p)
    got_st = torch.full((n,), -9, dtype=torch.int32, device="cuda")
    got_st2 = torch.full((n,), -9, dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in counts]
    torch.cuda.synchronize()
    blobs_k, c_k = _ptrs(_cut(d_blobs, counts, 131072), counts), _ptrs(_cut(d_c, counts, 48), counts)
    grp.blob_to_commitment_batch_group_dev(blobs_k, list(counts), _ptrs(_cut(got_c, counts, 48), counts), _ptrs(_cut(got_st, counts, 1), counts))
    grp.compute_blob_proof_batch_group_dev(blobs_k, c_k, list(counts), _ptrs(_cut(got_p, counts, 48), counts), _ptrs(_cut(got_st2, counts, 1), counts),
                                           streams=[s.cuda_stream for s in streams])
    torch.cuda.synchronize()
    assert torch.equal(got_c, want_c) and torch.equal(got_st, want_st)
    assert torch.equal(got_p, want_p) and torch.equal(got_st2, want_st2)


@pytest.mark.parametrize("counts", [(700, 700), (1399, 1), (0, 1400), (500, 400, 500), (16500, 16500)])
def test_group_dev_verify_matches_the_single_device_call(counts, engine, group2, group3, torch_cuda):
    """kzg_verify_blob_proof_batch_group_dev (src/kzg/setup.rs:223-275 over the members' resident shares, global order = member
    order): true on valid triples; a wrong proof in the LAST member's share -> false; then rejected inputs on different members --
    the code is the single-device call's (all blobs before any commitment before any proof, lowest global index:
    src/kzg/setup.rs:259-271).  33,000 triples: the members take the flat variable-base MSM."""
    import kateth_amd

    torch = torch_cuda
    grp = group2 if len(counts) == 2 else group3
    n = sum(counts)
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0x7E0 + n)

    def both():
        """(group result, single-device result) with errors as their kind names"""
        out = []
        for fn in (lambda: grp.verify_blob_proof_batch_group_dev(_ptrs(_cut(d_blobs, counts, 131072), counts), _ptrs(_cut(d_c, counts, 48), counts),
                                                                 _ptrs(_cut(d_p, counts, 48), counts), list(counts)),
                   lambda: engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)):
            try:
                out.append(fn())
            except kateth_amd.KzgError as err:
                inner = err.inner
                while hasattr(inner, "inner"):
                    inner = inner.inner
                out.append(type(inner).__name__ + ":" + inner.kind)
        return out

    assert both() == [True, True]
    saved = d_p[48 * (n - 2):48 * (n - 1)].clone()
    d_p[48 * (n - 2):48 * (n - 1)] = d_p[0:48]
    assert both() == [False, False]
    # a proof that is not a valid encoding on the FIRST busy member, a non-canonical blob on the LAST: the blob's error wins
    first_busy = 0 if counts[0] else counts[0]
    keep_p0 = d_p[48 * first_busy].item()
    d_p[48 * first_busy] = keep_p0 & 0x7F
    assert both() == ["ECGroupError:InvalidEncoding"] * 2
    keep_blob = d_blobs[(n - 1) * 131072:(n - 1) * 131072 + 32].clone()
    d_blobs[(n - 1) * 131072:(n - 1) * 131072 + 32] = 0xFF
    assert both() == ["BlobError:InvalidFieldElement"] * 2
    # two rejected commitments on different members: the lower global index decides (their codes differ)
    d_blobs[(n - 1) * 131072:(n - 1) * 131072 + 32] = keep_blob
    lo, hi = 1 if counts[0] > 1 else counts[0] + 1, n - 1
    keep_lo, keep_hi = d_c[48 * lo:48 * lo + 48].clone(), d_c[48 * hi].item()
    d_c[48 * lo:48 * lo + 48] = torch.tensor(list(bytes([0x80]) + bytes(46) + bytes([5])), dtype=torch.uint8, device="cuda")  # x = 5: not on the curve
    d_c[48 * hi] = keep_hi & 0x7F                                                                                               # compressed flag clear
    got = both()  # x = 5 decodes to a curve point outside the group (or to no point): either way not the other member's InvalidEncoding
    assert got[0] == got[1] and got[0] in ("ECGroupError:NotOnCurve", "ECGroupError:NotInGroup"), got
    d_c[48 * lo:48 * lo + 48] = keep_lo
    assert both() == ["ECGroupError:InvalidEncoding"] * 2
    d_c[48 * hi] = keep_hi
    d_p[48 * first_busy] = keep_p0
    d_p[48 * (n - 2):48 * (n - 1)] = saved
    assert both() == [True, True]


def test_group_dev_calls_on_a_single_device_context_and_bad_arguments(engine, group2, torch_cuda):
    """on a single-device context the group calls are the *_dev calls with one-element arrays; wrong array lengths are refused
    before anything is enqueued"""
    torch = torch_cuda
    n = 40
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0x515)
    out = torch.zeros_like(d_c)
    st = torch.full((n,), -9, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_group_dev([d_blobs.data_ptr()], [n], [out.data_ptr()], [st.data_ptr()])
    torch.cuda.synchronize()
    assert torch.equal(out, d_c) and int(st.abs().sum()) == 0
    assert engine.verify_blob_proof_batch_group_dev([d_blobs.data_ptr()], [d_c.data_ptr()], [d_p.data_ptr()], [n]) is True
    assert engine.verify_blob_proof_batch_group_dev([0], [0], [0], [0]) is True  # the empty batch (reference quirk Q4)
    assert group2.verify_blob_proof_batch_group_dev([0, 0], [0, 0], [0, 0], [0, 0]) is True
    with pytest.raises(ValueError):
        group2.verify_blob_proof_batch_group_dev([d_blobs.data_ptr()], [d_c.data_ptr()], [d_p.data_ptr()], [n])


# ---------------------------------------------------------------------------------------------------------------------------
# workspace slots: a caller that runs one call at a time grows ONE slot
# ---------------------------------------------------------------------------------------------------------------------------
def test_workspace_slots_grow_only_when_calls_are_in_flight(torch_cuda):
    import kateth_amd

    torch = torch_cuda
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    try:
        n = 600
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        s.synth_blobs_dev(0x77, 0, n, d_blobs.data_ptr())
        d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        assert s.workspace_bytes() == [0, 0, 0]
        for _ in range(4):  # one call at a time: commit, wait, prove, wait
            s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
            s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
        ws = s.workspace_bytes()
        assert ws[0] > 0 and ws[1] == 0 and ws[2] == 0, ws
        want = d_p.clone()
        # three calls in flight on three streams: the other slots appear, results unchanged
        lanes = [(torch.cuda.Stream(), torch.empty_like(d_p), torch.empty_like(d_st)) for _ in range(3)]
        torch.cuda.synchronize()
        for rep in range(2):
            for st, o, stt in lanes:
                s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, o.data_ptr(), stt.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        ws2 = s.workspace_bytes()
        assert ws2[0] >= ws[0] and ws2[1] > 0, ws2
        for _, o, stt in lanes:
            assert torch.equal(o, want) and int(stt.abs().sum()) == 0
    finally:
        s.close()
