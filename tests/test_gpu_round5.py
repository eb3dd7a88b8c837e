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
