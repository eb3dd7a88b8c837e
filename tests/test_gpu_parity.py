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
    import subprocess

    import kateth_amd

    assert os.path.exists(kateth_amd.library_path())
    assert engine.window_bits == 8 and engine.table_bytes > 0
    # the product library carries ONE fixed-base MSM kernel (the 32-bit-limb one lives in the test-only build)
    syms = subprocess.check_output(["strings", kateth_amd.library_path()], text=True)
    assert "k_msm_comb30" in syms and "k_msm_fixedILb" not in syms


def test_commitment_known_answers(engine):
    d = json.load(open(TRUSTED_SETUP))
    # [2]G and [3]G: the public BLS keys of the secret keys 2 and 3 (constants from outside this repository, tests/test_oracle_kat.py);
    # r - 1 everywhere = -G
    g2x = bytes.fromhex("a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e")
    g3x = bytes.fromhex("89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224")
    blobs = [be32(1) * 4096, bytes(131072), be32(2) * 4096, be32(3) * 4096, be32(R - 1) * 4096]
    want = [GEN48, INF48, g2x, g3x, bytes([GEN48[0] ^ 0x20]) + GEN48[1:]]
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
    """two small table classes (blocks of 4 and of 8 points) give identical bytes."""
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


def test_batch_transcript_and_lincombs_against_a_python_restatement(engine, torch_cuda):
    """verify_blob_kzg_proof_batch up to the pairing, restated with hashlib and the oracle's group law: the SHA-256 transcript
    tree (leaves H(C || z || y || pi), two levels of fan-out 16, ragged at the end), the challenge r, the powers r^i
    (src/kzg/setup.rs:138-150 with the spec's exponents) and both random linear combinations
        A = sum r^i pi_i        B = sum r^i C_i - (sum r^i y_i) G + sum r^i z_i pi_i      (setup.rs:151-160)
    for 37 triples (mid digests over 16 + 16 + 5 leaves) and for 300 (two node digests, the second over three mids)."""
    import hashlib

    from oracle.pyref import bls

    torch = torch_cuda
    sha = lambda b: hashlib.sha256(b).digest()  # noqa: E731
    for n, full in ((37, True), (300, False)):
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        engine.synth_blobs_dev(0x7A5C, 9, n, d_blobs.data_ptr())
        engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
        engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert int(d_st.abs().sum()) == 0
        com, prf = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
        sess, root, err = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
        assert err[0] == err[2] == err[4] == -1
        zs, ys = engine.verify_session_zy(sess, 0, n)
        out = engine.verify_phase2_dev(sess, root, 0, n)
        engine.verify_session_destroy(sess)
        # the transcript
        level = [sha(com[48 * i:48 * i + 48] + zs[32 * i:32 * i + 32] + ys[32 * i:32 * i + 32] + prf[48 * i:48 * i + 48]) for i in range(n)]
        for _ in range(2):
            level = [sha(b"".join(level[k:k + 16])) for k in range(0, len(level), 16)]
        assert len(level) == (n + 255) // 256
        assert sha(b"".join(level)) == root
        if not full:
            continue
        # the challenge and the two linear combinations
        r = bls.fr_hash_to(b"RCKZGBATCH___V1_" + (4096).to_bytes(16, "big") + n.to_bytes(16, "big") + root)
        A = B = None
        ysum, ri = 0, 1
        for i in range(n):
            C, pi = bls.g1_decompress(com[48 * i:48 * i + 48]), bls.g1_decompress(prf[48 * i:48 * i + 48])
            z, y = int.from_bytes(zs[32 * i:32 * i + 32], "big"), int.from_bytes(ys[32 * i:32 * i + 32], "big")
            A = bls.g1_add(A, bls.g1_mul(pi, ri))
            B = bls.g1_add(B, bls.g1_add(bls.g1_mul(C, ri), bls.g1_mul(pi, ri * z % R)))
            ysum = (ysum + ri * y) % R
            ri = ri * r % R
        B = bls.g1_add(B, bls.g1_neg(bls.g1_mul(bls.G1_GEN, ysum)))
        want = b"".join(v.to_bytes(48, "big") for v in (A[0], A[1], B[0], B[1]))
        assert out == want


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


def test_evaluation_kernel_on_and_off_the_domain(engine, oracle_setup, monkeypatch):
    """Polynomial::evaluate (src/kzg/poly.rs:10-33) through the VERIFICATION path's kernel (k_eval_frac, 16 and 64 lanes per
    blob): random points, 0, 1, r - 1, and -- poly.rs:14-18 -- every one of the sixteen positions of a bit-reversed group of
    roots, in the first, the last and interior groups (all lanes of both shapes are hit).  In verify_blob_kzg_proof_batch z
    is a hash output, so only this entry point reaches the kernel's on-domain case."""
    import random

    from oracle.pyref import blob as oblob
    from oracle.pyref import poly

    rng = random.Random(0xE7A1)
    blob = synth_blob(23, 0x3C)
    elements = oblob.from_slice(blob)
    roots = oracle_setup.roots_of_unity_brp
    zs, want = [], []
    for z in [0, 1, R - 1] + [rng.randrange(R) for _ in range(5)]:
        zs.append(z)
        want.append(poly.evaluate(elements, z, oracle_setup))
    on_domain = []
    for octet in (0, 1, 15, 16, 17, 63, 64, 255, 300, 510, 511):
        on_domain += [8 * octet + j for j in range(8)]
    on_domain += [rng.randrange(4096) for _ in range(16)]
    for i in on_domain:
        zs.append(roots[i])
        want.append(elements[i])  # poly.rs:14-18: the evaluation at a root is that element
    assert poly.evaluate(elements, roots[on_domain[13]], oracle_setup) == elements[on_domain[13]]  # the oracle agrees with the shortcut
    n = len(zs)
    blobs = blob * n
    points = b"".join(be32(z) for z in zs)
    for group in ("16", "64"):
        e2 = _engine_with_env(monkeypatch, {"KATETH_AMD_EVAL_GROUP": group})
        try:
            ys, status = e2.evaluate_blobs(blobs, points)
            assert status == [0] * n
            for k in range(n):
                assert ys[32 * k:32 * k + 32] == be32(want[k]), (group, k, hex(zs[k]))
            # a non-canonical z and a non-canonical blob element are reported per item
            bad_blob = bytearray(blob)
            bad_blob[32 * 77:32 * 77 + 32] = b"\xff" * 32
            ys2, st2 = e2.evaluate_blobs(blob + bytes(bad_blob) + blob, be32(5) + be32(5) + b"\xff" * 32)
            assert st2[0] == 0 and st2[1] == 2 and st2[2] == 7
            assert ys2[:32] == be32(poly.evaluate(elements, 5, oracle_setup)) and ys2[32:] == bytes(64)
        finally:
            e2.close()


def test_two_evaluation_kernels_agree_at_scale(engine, torch_cuda):
    """Polynomial::evaluate lives twice on the device: in the proof path (k_poly: batch inversion, 8 x 32-bit limbs, one
    workgroup per blob) and in the verification path (k_eval_frac: inversion-free fraction sums over hexes of roots, radix
    2^29, 16 lanes per blob at this size).  Independent code, same answers on 4,100 (blob, z) pairs -- random points plus a
    sprinkling of points on the domain."""
    import random

    from oracle.pyref import domain

    torch = torch_cuda
    n = 4100  # >= 4,096: the 16-lane shape of k_eval_frac, ragged last wave
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xE7A2, 5, n, d_blobs.data_ptr())
    blobs = d_blobs.cpu().numpy().tobytes()
    rng = random.Random(0xE7A2)
    roots = domain.bit_reversal_permutation(domain.roots_of_unity(4096))
    zs = [rng.randrange(R) for _ in range(n)]
    for k in range(0, n, 97):
        zs[k] = roots[rng.randrange(4096)]
    points = b"".join(be32(z) for z in zs)
    _, ys_proof, st_proof = engine.compute_proof_batch(blobs, points)
    ys_eval, st_eval = engine.evaluate_blobs(blobs, points)
    assert st_proof == [0] * n and st_eval == [0] * n
    assert ys_eval == ys_proof


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


def _engine_with_env(monkeypatch, env, window_bits=8, lib_path=None):
    """environment knobs are read once, at kzg_ctx_create: a context per setting"""
    import kateth_amd

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=window_bits, lib_path=lib_path)


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
        e2 = _engine_with_env(monkeypatch, {"KATETH_AMD_PROOF_CHUNK": chunk, "KATETH_AMD_PROOF_OVERLAP": overlap})
        try:
            got_p = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
            got_st = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            e2.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, got_p.data_ptr(), got_st.data_ptr())
            torch.cuda.synchronize()
            assert got_st.cpu().tolist() == st, (chunk, overlap)
            assert got_p.cpu().numpy().tobytes() == want_p.cpu().numpy().tobytes(), (chunk, overlap)
        finally:
            e2.close()


def test_three_challenge_kernels_agree(engine, torch_cuda, monkeypatch):
    """Blob::challenge (src/blob.rs:78-97) has three device kernels: rounds on lane pairs (small batches), the
    producer/consumer wave pair (mid-size batches) and one lane per blob (chip-filling batches).  Host-buffer verification
    in chunks of 64 blobs hashes every chunk with the kernel the batch-size rule picks, and KATETH_AMD_CHALLENGE_SPLIT_MAX
    overrides that rule: the same 130 triples (three chunks, a ragged last one) through all three kernels must verify,
    and must reject a swapped proof -- a wrong z_i from any kernel breaks the batch equation"""
    torch = torch_cuda
    n = 130
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x5AA5, 1, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    blobs, cs, ps = (t.cpu().numpy().tobytes() for t in (d_blobs, d_c, d_p))
    swapped = ps[48:96] + ps[48:]
    for env in ({}, {"KATETH_AMD_CHALLENGE_SPLIT_MAX": "1000000"}, {"KATETH_AMD_CHALLENGE_SPLIT_MAX": "1"}):
        e2 = _engine_with_env(monkeypatch, dict(env, KATETH_AMD_VERIFY_CHUNK="64"))
        try:
            assert e2.verify_blob_proof_batch_host(blobs, cs, ps, n) is True, env
            assert e2.verify_blob_proof_batch_host(blobs, cs, swapped, n) is False, env
        finally:
            e2.close()


def test_flat_and_classic_variable_base_msm_agree(engine, torch_cuda, monkeypatch):
    """the three lincombs of verify_proof_batch (src/kzg/setup.rs:152-155) run as two variable-base MSMs; from 32,768 terms
    on they take the flat path (c = 13, one thread per bucket, bit sums), below that -- and with KATETH_AMD_VAR_MSM=classic --
    c = 8 with per-bucket partials.  33,000 triples (A: 33,000 terms, B: 66,001), a repeated point and an infinity among them:
    the two partial sums of phase 2 must be the same 192 bytes from both paths, and both must accept / reject alike"""
    torch = torch_cuda
    n = 33000
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x7A57, 5, n, d_blobs.data_ptr())
    d_blobs[131072 * 3: 131072 * 4] = d_blobs[131072 * 2: 131072 * 3]  # blob 3 == blob 2: a repeated commitment/proof pair
    d_blobs[131072 * 7: 131072 * 8] = 0  # zero blob: commitment and proof are the point at infinity
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    bad_p = d_p.clone()
    bad_p[48 * 11: 48 * 12] = d_p[48 * 12: 48 * 13]
    classic = _engine_with_env(monkeypatch, {"KATETH_AMD_VAR_MSM": "classic"})
    # round 5: KATETH_AMD_VAR_GLV=1 splits every scalar at z^2 (GLV: twice the terms on 128-bit scalars, the second half on the
    # points' [z^2]-images (beta x, -y)) -- a third, independent decomposition of the same two sums (measured, not the default)
    monkeypatch.delenv("KATETH_AMD_VAR_MSM")
    noglv = _engine_with_env(monkeypatch, {"KATETH_AMD_VAR_GLV": "1"})
    monkeypatch.delenv("KATETH_AMD_VAR_GLV")
    # the default flat path gives every LANE the same number of entries (k_var_buckets_seg + k_var_seg_fixup); KATETH_AMD_VAR_SEG=0 is
    # the flat path with one thread per bucket: a fourth way to the same two sums
    perbucket = _engine_with_env(monkeypatch, {"KATETH_AMD_VAR_SEG": "0"})
    monkeypatch.delenv("KATETH_AMD_VAR_SEG")
    try:
        sums = []
        for e in (engine, classic, noglv, perbucket):
            sess, root, err = e.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
            assert err[0] == err[2] == err[4] == -1
            sums.append(e.verify_phase2_dev(sess, root, 0, n))
            e.verify_session_destroy(sess)
            assert e.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
            assert e.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), bad_p.data_ptr(), n) is False
        assert sums[0] == sums[1] == sums[2] == sums[3] and len(sums[0]) == 192
        # every triple the same: all buckets collect multiples of ONE point, so the complete adder's P == Q branch is taken in
        # the bucket chains (second entry of every list) and between equal bucket sums in the bit-sum trees
        d_blobs.view(n, 131072)[:] = d_blobs.view(n, 131072)[0].clone()
        d_c.view(n, 48)[:] = d_c.view(n, 48)[0].clone()
        d_p.view(n, 48)[:] = d_p.view(n, 48)[0].clone()
        torch.cuda.synchronize()
        sums = []
        for e in (engine, classic, noglv, perbucket):
            sess, root, err = e.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
            assert err[0] == err[2] == err[4] == -1
            sums.append(e.verify_phase2_dev(sess, root, 0, n))
            e.verify_session_destroy(sess)
            assert e.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
        assert sums[0] == sums[1] == sums[2] == sums[3]
    finally:
        classic.close()
        noglv.close()
        perbucket.close()


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
    zy = {}
    for group in ("16", "64"):
        e2 = _engine_with_env(monkeypatch, {"KATETH_AMD_EVAL_GROUP": group})
        try:
            for m in (1, 2, 3, 4, 5, 67, n):
                assert e2.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), m) is True, (group, m)
            assert e2.verify_blob_proof_batch_dev(flipped.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False
            assert e2.verify_blob_proof_batch_dev(flipped.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 77) is True
            with pytest.raises(kateth_amd.KzgError) as err:
                e2.verify_blob_proof_batch_dev(bad_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
            assert isinstance(err.value.inner, kateth_amd.BlobError) and err.value.inner.kind == "InvalidFieldElement"
            assert e2.verify_blob_proof_batch_dev(bad_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 129) is True  # the bad blob is item 129
            sess, _, _ = e2.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
            zy[group] = e2.verify_session_zy(sess, 0, n)
            e2.verify_session_destroy(sess)
        finally:
            e2.close()
    assert zy["16"] == zy["64"]  # both shapes: the same challenges and evaluations, byte for byte


def test_comb_and_window_table_kernels_agree_at_scale(torch_cuda, monkeypatch, window_msm_lib):
    """4,096 random blobs through three independent fixed-base MSMs must give identical commitments: the product's
    subset-sum comb (msm_comb.cuh), and -- from the TEST-ONLY build (tests/window_msm) -- round 1's
    window-table kernels on radix-2^28 limbs and on 12 x 32-bit limbs (tests/window_msm/window_msm.hip, hooked in through
    the MsmOverride extension point).  Different tables, different recodings (signed bits
    vs signed windows), different field representations: 2e8 - 5e8 mixed additions each, including the few thousand that
    take the out-of-line complete adder."""
    import kateth_amd

    torch = torch_cuda
    n = 4096
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    outs = []
    for env, lib in (({}, None), ({"KATETH_AMD_MSM": "window"}, window_msm_lib), ({"KATETH_AMD_MSM_RADIX": "32"}, window_msm_lib)):
        for k in ("KATETH_AMD_MSM", "KATETH_AMD_MSM_RADIX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, lib_path=lib)
        try:
            assert s.msm_kernel_name == {0: "k_msm_comb30", 1: "k_msm_fixed28", 2: "k_msm_fixed"}[len(outs)]
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
    assert outs[0] == outs[1] == outs[2]
    assert len(set(outs[0][48 * i:48 * i + 48] for i in range(n))) == n  # all distinct: nothing degenerate was compared


def test_comb_geometries_agree(golden, torch_cuda, monkeypatch):
    """every comb geometry -- blocks of 4, 8, 16 points and 1 to 16 plane groups, whole-blob waves and every split the
    geometry allows (a single blob uses the largest) -- gives the golden commitments and proofs"""
    import kateth_amd

    recs = golden["blobs"][:3]
    blobs = b"".join(synth_blob(r["index"]) for r in recs)
    cs = b"".join(bytes.fromhex(r["commitment"]) for r in recs)
    for wb, groups in ((4, 16), (4, 1), (8, 2), (8, 4), (8, 8), (8, 16), (16, 4)):
        monkeypatch.setenv("KATETH_AMD_COMB_GROUPS", str(groups))
        for splits in (None, 1, 2, 4):
            if splits is None:
                monkeypatch.delenv("KATETH_AMD_MSM_SPLITS", raising=False)
            else:
                monkeypatch.setenv("KATETH_AMD_MSM_SPLITS", str(splits))
            s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=wb)
            try:
                assert s.plane_groups == groups and s.window_bits == wb
                out, st = s.blob_to_commitment_batch(blobs)
                assert st == [0, 0, 0] and out == cs, (wb, groups, splits)
                assert s.blob_to_commitment(blobs[:131072]) == cs[:48]
                proofs, st = s.compute_blob_proof_batch(blobs[:131072], cs[:48])
                assert st == [0] and proofs.hex() == recs[0]["proof"], (wb, groups, splits)
            finally:
                s.close()


def test_host_buffer_commitment_pipeline_matches_device_path(engine, torch_cuda):
    """kzg_blob_to_commitment_batch streams host blobs in chunks of at least 512 blobs (copy of chunk k+1 beside the MSM of
    chunk k, one reduce/compress per group): two full chunks and a ragged one, an invalid blob in the last chunk, against the
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


def test_host_buffer_commitment_chunk_plans(engine, torch_cuda):
    """the larger chunk plans of kzg_blob_to_commitment_batch: 8,400 blobs = chunks of 2,048 in a group of 8,192 + a tail group
    of 208 with splits of its own; 16,500 blobs = four half-wave chunks of 4,096 (two blobs per wave, each its own group) + a
    tail of 116; an invalid blob in a half-wave chunk; against the device-pointer entry point"""
    torch = torch_cuda
    n = 16500
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0xC4A2, 3, n, d_blobs.data_ptr())
    d_blobs[9000 * 131072 + 32 * 77: 9000 * 131072 + 32 * 77 + 32] = 0xFF  # blob 9000, element 77: not canonical
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    want, want_st = d_c.cpu().numpy().tobytes(), d_st.cpu().tolist()
    assert want_st[9000] == 2 and sum(1 for v in want_st if v) == 1
    host = d_blobs.cpu().numpy()
    del d_blobs
    for m in (8400, n):
        got, got_st = engine.blob_to_commitment_batch(host[: m * 131072].tobytes(), m)
        assert got_st == want_st[:m], m
        assert got == want[: 48 * m], m


def test_host_buffer_proof_passes_match_the_device_path(engine, torch_cuda):
    """kzg_compute_blob_proof_batch above 5,120 blobs walks the batch in double-buffered passes (1,024, then 4,096s, then the
    rest; the copy of pass k+1 beside the device path on pass k): 5,300 blobs = passes of 1,024 + 4,096 + 180, an invalid blob
    and an undecodable commitment in different passes, against the device-pointer entry point; also compute_kzg_proof (z given)
    through the same passes"""
    torch = torch_cuda
    n = 5300
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x9A55, 5, n, d_blobs.data_ptr())
    d_blobs[3000 * 131072 + 32 * 9: 3000 * 131072 + 32 * 9 + 32] = 0xFF  # blob 3000 (second pass), element 9: not canonical
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    d_c[3000 * 48: 3001 * 48] = d_c[0:48]          # a decodable commitment for the invalid blob
    d_c[5250 * 48] = d_c[5250 * 48] & 0x7F         # third pass: compression bit cleared -> InvalidEncoding
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    want, want_st = d_p.cpu().numpy().tobytes(), d_st.cpu().tolist()
    assert want_st[3000] == 2 and want_st[5250] == 3 and sum(1 for v in want_st if v) == 2
    host_blobs, host_c = d_blobs.cpu().numpy().tobytes(), d_c.cpu().numpy().tobytes()
    got, got_st = engine.compute_blob_proof_batch(host_blobs, host_c)
    assert got_st == want_st
    assert got == want
    # proofs at given points through the same passes: y and proof equal the blob-proof values when z is the blob's challenge
    sess, _, _ = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 64)
    z64, y64 = engine.verify_session_zy(sess, 0, 64)
    engine.verify_session_destroy(sess)
    m = 5200
    zs = (z64 * (m // 64 + 1))[: 32 * m]
    proofs, ys, st = engine.compute_proof_batch(host_blobs[: m * 131072], zs)
    assert [k for k, v in enumerate(st) if v] == [3000] and st[3000] == 2
    assert proofs[: 48 * 64] == want[: 48 * 64] and ys[: 32 * 64] == y64
    # a blob in the last pass, checked through the single-item path
    one_p, one_y, one_st = engine.compute_proof_batch(host_blobs[5199 * 131072: 5200 * 131072], zs[32 * 5199: 32 * 5200])
    assert one_st == [0] and proofs[48 * 5199: 48 * 5200] == one_p and ys[32 * 5199: 32 * 5200] == one_y


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


def test_radix32_msm_kernel_is_bit_exact(golden, monkeypatch, window_msm_lib):
    """the test-only library (tests/window_msm) with KATETH_AMD_MSM_RADIX=32 runs round 1's 12 x 32-bit-limb
    window-table MSM kernel over a 2^384-Montgomery table -- an independent implementation of the same sum; the product
    library carries only the comb kernel (msm_comb.cuh).  Same bytes (the rest of this file runs the product)."""
    import kateth_amd

    monkeypatch.setenv("KATETH_AMD_MSM_RADIX", "32")
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=7, lib_path=window_msm_lib)
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


# ---------------------------------------------------------------------------
# production configurations (BASELINE.json configs[1]-[3]): the windows the library and bench.py really use, and
# batch verification at 65,536 -- the only size at which the one-lane-per-blob SHA-256 kernel (k_challenge) and the
# variable-base MSM over 131,073 terms run
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cport_setup():
    from oracle.cport import binding

    cs = binding.CSetup(binding.load(), TRUSTED_SETUP, subgroup_checks=False, threads=1)
    yield cs
    cs.close()


@pytest.mark.parametrize("window_bits", [16, 22])
def test_production_windows_against_golden_and_c_port(window_bits, golden, torch_cuda, cport_setup):
    """the table classes the library and bench.py really use -- 16 (kzg_ctx_create's default: comb blocks of 16 points, 16 plane
    groups, 12.9 GB) and 22 (bench.py: blocks of 22 + 21 + 21, 8 plane groups, 192 GiB): commitments and proofs of the six
    golden blobs equal the golden vectors; commitments of 96 synthetic blobs equal the C port of the reference's CPU path
    (oracle/cport, Pippenger c = 10 -- a different algorithm over the same points); for all 96 the challenge z and the
    evaluation y the engine derives equal the C port's, and commit -> prove -> verify closes, so the proofs are the
    unique points that satisfy the pairing equation for the C port's (z, y)."""
    import kateth_amd
    from oracle.cport import binding

    torch = torch_cuda
    torch.cuda.empty_cache()
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=window_bits)
    try:
        assert s.window_bits == window_bits
        assert s.plane_groups in {16: (16,), 22: (8, 4)}[window_bits]  # class 22: 8 groups when 192 GiB + headroom are free, else 4
        assert s.table_bytes == s.plane_groups * 64 * {16: 4 << 15, 22: 1 << 22}[window_bits] * 96  # 12.9 GB / 192 GiB
        n = 96
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        s.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
        d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
        s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert int(d_st.abs().sum()) == 0
        cs, ps, host = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes(), d_blobs.cpu().numpy().tobytes()
        for rec in golden["blobs"]:
            b = rec["index"]
            assert cs[48 * b:48 * b + 48].hex() == rec["commitment"], (window_bits, b)
            assert ps[48 * b:48 * b + 48].hex() == rec["proof"], (window_bits, b)
        cport_setup.set_threads(binding.host_cores())
        _, want = cport_setup.time_commitments_blob_parallel(host, n, 1, binding.host_cores())
        assert cs == want
        # z, y of every item against the C port of Blob::challenge / Polynomial::evaluate (batch-inversion variant: same values)
        rc, z_ref, y_ref, _, _ = cport_setup.verify_batch_prepairing(host, cs, ps, n, batch_inverse=True)
        assert rc == 0
        sess, _, err6 = s.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
        z_gpu, y_gpu = s.verify_session_zy(sess, 0, n)
        s.verify_session_destroy(sess)
        assert err6[0] == err6[2] == err6[4] == -1
        assert z_gpu == z_ref and y_gpu == y_ref
        for rec in golden["blobs"]:
            b = rec["index"]
            assert z_gpu[32 * b:32 * b + 32].hex() == rec["challenge_z"] and y_gpu[32 * b:32 * b + 32].hex() == rec["eval_y"]
        assert s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
        # the single-blob shape of the same window (many split units per blob; 256 on the latency comb) gives the same bytes
        assert s.blob_to_commitment(host[:131072]).hex() == golden["blobs"][0]["commitment"]
    finally:
        s.close()
        torch.cuda.empty_cache()


def test_verify_batch_65536(engine, golden, torch_cuda):
    """BASELINE configs[3]: verify_blob_kzg_proof_batch at n = 65,536 (8 GiB of blobs resident).  The batch is valid ->
    true; one corrupted proof -> false; one non-canonical element in a blob beyond index 32,768 -> InvalidFieldElement
    with that blob as the first error; challenges and evaluations of this size's kernels (one lane per blob SHA-256,
    16 lanes per blob evaluation) equal the golden vectors for blobs 0..5 and the small-batch kernels' values for a
    window of blobs near the end of the batch."""
    import kateth_amd

    torch = torch_cuda
    torch.cuda.empty_cache()
    n = 65536
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    cs, ps = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    for rec in golden["blobs"]:
        b = rec["index"]
        assert cs[48 * b:48 * b + 48].hex() == rec["commitment"] and ps[48 * b:48 * b + 48].hex() == rec["proof"]
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    # z, y out of the 65,536-batch kernels
    sess, root_full, err6 = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
    assert err6[0] == err6[2] == err6[4] == -1
    z_head, y_head = engine.verify_session_zy(sess, 0, 6)
    first = 65000
    z_tail, y_tail = engine.verify_session_zy(sess, first, 64)
    engine.verify_session_destroy(sess)
    for rec in golden["blobs"]:
        b = rec["index"]
        assert z_head[32 * b:32 * b + 32].hex() == rec["challenge_z"] and y_head[32 * b:32 * b + 32].hex() == rec["eval_y"]
    sess, _, _ = engine.verify_phase1_dev(d_blobs.data_ptr() + first * 131072, d_c.data_ptr() + first * 48, d_p.data_ptr() + first * 48, 64)
    z_small, y_small = engine.verify_session_zy(sess, 0, 64)
    engine.verify_session_destroy(sess)
    assert z_tail == z_small and y_tail == y_small
    # proofs computed with the one-lane-per-blob hash (a chunk above 32,768) equal the small-batch proofs
    e2 = _engine_with_env_plain({"KATETH_AMD_PROOF_CHUNK": "65536"})
    try:
        d_p2 = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        e2.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p2.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert int(d_st.abs().sum()) == 0
        assert torch.equal(d_p2, d_p)
        del d_p2
    finally:
        e2.close()
    # one corrupted proof
    keep = d_p[48 * 40000:48 * 40001].clone()
    d_p[48 * 40000:48 * 40001] = d_p[0:48]
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False
    d_p[48 * 40000:48 * 40001] = keep
    # one non-canonical element beyond index 32,768, and a later one: the earlier is the first error
    bad = 50001
    keep_b = d_blobs[bad * 131072 + 32 * 77: bad * 131072 + 32 * 78].clone()
    d_blobs[bad * 131072 + 32 * 77: bad * 131072 + 32 * 78] = 0xFF
    d_blobs[60000 * 131072: 60000 * 131072 + 32] = 0xFF
    with pytest.raises(kateth_amd.KzgError) as err:
        engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
    assert isinstance(err.value.inner, kateth_amd.BlobError) and err.value.inner.kind == "InvalidFieldElement"
    sess, _, err6 = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
    engine.verify_session_destroy(sess)
    assert err6[0] == bad and err6[1] == 2 and err6[2] == -1 and err6[4] == -1
    d_blobs[bad * 131072 + 32 * 77: bad * 131072 + 32 * 78] = keep_b
    del d_blobs, d_c, d_p
    torch.cuda.empty_cache()


def _engine_with_env_plain(env, window_bits=8):
    import kateth_amd

    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=window_bits)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_affine_outputs_are_the_compressed_points(engine, golden, oracle_setup):
    """kzg_*_affine: the producers return POINTS (Commitment = Proof = P1, src/kzg/mod.rs:9-10) as 96-byte blst_p1_affine
    images -- x || y, little-endian limbs of the 2^384-Montgomery residue.  Checked against the oracle's decompression
    of the golden 48-byte encodings; infinity is 96 zero bytes; a rejected blob gives zeros and its status."""
    from oracle.pyref import bls

    def image(p48):
        pt = bls.g1_uncompress(p48)
        if pt is None:
            return bytes(96)
        x, y = pt
        return (x * (1 << 384) % bls.P).to_bytes(48, "little") + (y * (1 << 384) % bls.P).to_bytes(48, "little")

    recs = golden["blobs"][:3]
    blobs = b"".join(synth_blob(r["index"]) for r in recs)
    cs = b"".join(bytes.fromhex(r["commitment"]) for r in recs)
    bad = bytearray(synth_blob(0))
    bad[0:32] = be32(R)
    aff, st = engine.blob_to_commitment_batch_affine(blobs + bytes(131072) + bytes(bad))
    assert st == [0, 0, 0, 0, 2]
    for k, r in enumerate(recs):
        assert aff[96 * k:96 * k + 96] == image(bytes.fromhex(r["commitment"])), k
    assert aff[288:384] == bytes(96) and aff[384:480] == bytes(96)
    paff, st = engine.compute_blob_proof_batch_affine(blobs, cs)
    assert st == [0, 0, 0]
    for k, r in enumerate(recs):
        assert paff[96 * k:96 * k + 96] == image(bytes.fromhex(r["proof"])), k
    zs = b"".join(bytes.fromhex(r["kzg_proof_at"]["z"]) for r in recs)
    qaff, ys, st = engine.compute_proof_batch_affine(blobs, zs)
    assert st == [0, 0, 0]
    for k, r in enumerate(recs):
        assert qaff[96 * k:96 * k + 96] == image(bytes.fromhex(r["kzg_proof_at"]["proof"])) and ys[32 * k:32 * k + 32].hex() == r["kzg_proof_at"]["y"]


def test_host_buffer_verify_pipeline_matches_device_path(engine, torch_cuda):
    """kzg_verify_blob_proof_batch streams host blobs through the staging arena in chunks (copy of chunk k+1 beside the
    hash + evaluation of chunk k, rotating compute streams, slot reuse beyond 4 chunks): same decisions and the same first
    error as the device-pointer entry point, ragged last chunk included; with a tiny chunk size the slot-reuse path runs."""
    import kateth_amd

    torch = torch_cuda
    n = 1111
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x7E57, 5, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    hb, hc, hp = d_blobs.cpu().numpy().tobytes(), d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    blobs = [hb[i * 131072:(i + 1) * 131072] for i in range(n)]
    cs = [hc[i * 48:(i + 1) * 48] for i in range(n)]
    ps = [hp[i * 48:(i + 1) * 48] for i in range(n)]
    small = _engine_with_env_plain({"KATETH_AMD_VERIFY_CHUNK": "37"})  # 31 chunks: more than the 4 staging slots
    try:
        for e in (engine, small):
            for m in (n, 512, 513, 1):
                assert e.verify_blob_proof_batch(blobs[:m], cs[:m], ps[:m]) is True, m
            assert e.verify_blob_proof_batch(blobs, cs, ps[:700] + [ps[0]] + ps[701:]) is False
            badb = bytearray(blobs[1000])
            badb[32 * 9:32 * 10] = b"\xff" * 32
            with pytest.raises(kateth_amd.KzgError) as err:
                e.verify_blob_proof_batch(blobs[:1000] + [bytes(badb)] + blobs[1001:], cs, ps[:3] + [bytes([0xE0]) + bytes(47)] + ps[4:])
            assert isinstance(err.value.inner, kateth_amd.BlobError)  # the blob error wins over the earlier proof error
            with pytest.raises(kateth_amd.KzgError) as err:
                e.verify_blob_proof_batch(blobs, cs, ps[:3] + [bytes([0xE0]) + bytes(47)] + ps[4:])
            assert err.value.inner.inner.kind == "InvalidEncoding"
    finally:
        small.close()


def test_concurrent_callers_on_one_context(engine, golden, torch_cuda):
    """`Setup` is shared behind an Arc and every method takes &self (src/kzg/setup.rs:323): four host threads drive ONE
    context at the same time, mixing commitments, proofs and verifications (host-buffer and device-pointer forms, each
    thread on its own stream); every result must equal the serial run's."""
    import threading

    torch = torch_cuda
    n = 48
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    hb, hc, hp = d_blobs.cpu().numpy().tobytes(), d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    blobs = [hb[i * 131072:(i + 1) * 131072] for i in range(n)]
    cs = [hc[i * 48:(i + 1) * 48] for i in range(n)]
    ps = [hp[i * 48:(i + 1) * 48] for i in range(n)]
    rec0 = golden["blobs"][0]
    at = rec0["kzg_proof_at"]
    errors = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                my_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
                my_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
                my_st = torch.empty(n, dtype=torch.int32, device="cuda")
                for it in range(6):
                    k = (tid + it) % 4
                    if k == 0:
                        engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, my_c.data_ptr(), my_st.data_ptr(), stream.cuda_stream)
                        stream.synchronize()
                        assert my_c.cpu().numpy().tobytes() == hc
                        out, st = engine.blob_to_commitment_batch(hb[: 5 * 131072])
                        assert out == hc[: 5 * 48] and st == [0] * 5
                    elif k == 1:
                        engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, my_p.data_ptr(), my_st.data_ptr(), stream.cuda_stream)
                        stream.synchronize()
                        assert my_p.cpu().numpy().tobytes() == hp
                        assert engine.blob_proof(blobs[3], cs[3]) == ps[3]
                    elif k == 2:
                        assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n, stream.cuda_stream) is True
                        assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr() + 48, n - 1, stream.cuda_stream) is False
                        assert engine.verify_proof(bytes.fromhex(at["proof"]), bytes.fromhex(rec0["commitment"]), bytes.fromhex(at["z"]), bytes.fromhex(at["y"])) is True
                    else:
                        assert engine.verify_blob_proof_batch(blobs[:9], cs[:9], ps[:9]) is True
                        assert engine.verify_blob_proof_batch(blobs[:9], cs[:9], ps[1:10]) is False
                        assert engine.verify_blob_proof(blobs[7], cs[7], ps[7]) is True
        except BaseException as err:  # noqa: BLE001
            errors.append((tid, repr(err)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


# ---------------------------------------------------------------------------
# pins that do not pass through the oracle's MSM / quotient code: the ceremony's G2 points and closed forms
# ---------------------------------------------------------------------------
def _blob_of(evals):
    return b"".join(be32(v) for v in evals)


def test_gpu_commitment_against_all_65_ceremony_g2_points(engine, oracle_setup):
    """p(x) = sum_k rho_k x^k (k = 0..64): the ENGINE's commitment C must satisfy e(C, G2) == e(G1, sum_k rho_k [tau^k]_2)
    with the trusted setup's own G2 monomial points -- data neither the engine nor the oracle produced (only the oracle's
    pairing and G2 arithmetic take part in the check, not its MSM)."""
    import random

    from oracle.pyref import bls

    rnd = random.Random(6565)
    roots = oracle_setup.roots_of_unity_brp
    rho = [rnd.randrange(1, R) for _ in range(65)]
    evals = []
    for w in roots:
        acc = 0
        for c in reversed(rho):
            acc = (acc * w + c) % R
        evals.append(acc)
    c48 = engine.blob_to_commitment(_blob_of(evals))
    q = None
    for k, c in enumerate(rho):
        q = bls.g2_add(q, bls.g2_mul(oracle_setup.g2_monomial[k], c))
    assert bls.verify_pairings((bls.g1_uncompress(c48), bls.G2_GEN), (bls.G1_GEN, q))
    assert not bls.verify_pairings((bls.g1_uncompress(c48), bls.G2_GEN), (bls.G1_GEN, bls.g2_add(q, oracle_setup.g2_monomial[33])))


def test_gpu_proofs_on_monomials_match_the_closed_form(engine, oracle_setup):
    """p(x) = x^k: y = z^k and proof = sum_{j<k} z^(k-1-j) [tau^j]_1, with [tau^j]_1 the engine's own commitment of x^j
    (tied to the ceremony by the test above); the combination uses only the oracle's G1 add / scalar-mul.  z off the
    domain and on it (src/kzg/poly.rs:50-64)."""
    from oracle.pyref import bls

    roots = oracle_setup.roots_of_unity_brp
    k = 7
    blobs = b"".join(_blob_of([pow(w, j, R) for w in roots]) for j in range(k + 1))
    cs, st = engine.blob_to_commitment_batch(blobs)
    assert st == [0] * (k + 1) and cs[:48] == GEN48
    taus = [bls.g1_uncompress(cs[48 * j:48 * j + 48]) for j in range(k)]
    xk = blobs[k * 131072:]
    for z in (0xFEEDFACECAFEBEEF, roots[3000], roots[1]):
        proof, y = engine.proof(xk, be32(z))
        assert y == be32(pow(z, k, R))
        want = None
        for j in range(k):
            want = bls.g1_add(want, bls.g1_mul(taus[j], pow(z, k - 1 - j, R)))
        assert proof == bls.g1_compress(want), hex(z)
        assert engine.verify_proof(proof, cs[48 * k:48 * k + 48], be32(z), y) is True


def test_point_returning_methods_match_the_byte_returning_ones(engine, golden):
    """`Setup::blob_to_commitment / blob_proof / proof` return points in the reference (src/kzg/setup.rs:167,177,185) and
    every caller compresses next (benches/kzg.rs:24-32): the mirror's *_point methods + P1.compress give the golden bytes."""
    import random

    import kateth_amd

    rec = golden["blobs"][1]
    blob = kateth_amd.Blob.from_slice(synth_blob(rec["index"]))
    c = engine.blob_to_commitment_point(blob)
    assert isinstance(c, kateth_amd.P1) and c.compress().hex() == rec["commitment"]
    p = engine.blob_proof_point(blob, c.compress())
    assert p.compress().hex() == rec["proof"]
    q, y = engine.proof_point(blob, bytes.fromhex(rec["kzg_proof_at"]["z"]))
    assert q.compress().hex() == rec["kzg_proof_at"]["proof"] and y.hex() == rec["kzg_proof_at"]["y"]
    assert engine.blob_to_commitment_point(bytes(131072)).is_inf()
    # Blob::random -> commit -> prove -> verify (the bench's input pipeline, benches/kzg.rs:17-33)
    rb = kateth_amd.Blob.random(random.Random(7))
    c48 = engine.blob_to_commitment_point(rb).compress()
    p48 = engine.blob_proof_point(rb, c48).compress()
    assert engine.verify_blob_proof(rb.to_bytes(), c48, p48) is True


def test_public_g1_decompress_matches_the_oracle(engine, golden, oracle_setup):
    """kzg_g1_decompress_batch = `P1::decompress` (src/bls.rs:505-531) for a batch: golden commitments and proofs, the
    generator, infinity and each rejection class against the oracle's decoder -- the accepted points as blst_p1_affine images
    (x || y, little-endian limbs of the 2^384-Montgomery residue), the rejected ones with the reference's error code"""
    import kateth_amd
    from oracle.pyref import bls

    P = bls.P
    good = [bytes.fromhex(r["commitment"]) for r in golden["blobs"]] + [bytes.fromhex(r["proof"]) for r in golden["blobs"]] + [GEN48, INF48]
    x_off_curve = next(x for x in range(1, 200) if pow((x ** 3 + 4) % P, (P - 1) // 2, P) != 1)
    # an on-curve point outside the subgroup: x = 4 has y^2 = 68, a square; (4, y) has large cofactor order
    x_on = next(x for x in range(1, 200) if pow((x ** 3 + 4) % P, (P - 1) // 2, P) == 1 and not bls.g1_in_subgroup((x, pow((x ** 3 + 4) % P, (P + 1) // 4, P))))
    enc = lambda x, flags=0x80: bytes([x.to_bytes(48, "big")[0] | flags]) + x.to_bytes(48, "big")[1:]  # noqa: E731
    bad = [bytes([GEN48[0] & 0x7F]) + GEN48[1:],  # compression bit clear -> InvalidEncoding
           enc(P),                                  # x >= p -> InvalidEncoding
           bytes([0xC0]) + bytes(46) + b"\x01",     # infinity flag with a non-zero tail -> InvalidEncoding
           enc(x_off_curve),                        # no square root -> NotOnCurve
           enc(x_on)]                               # on the curve, not in the subgroup -> NotInGroup
    pts, st = engine.decompress_g1_batch(good + bad)
    assert st == [0] * len(good) + [3, 3, 3, 4, 5]
    for b, pt in zip(good, pts):
        want = bls.g1_decompress(b)
        if want is None:
            assert pt.is_inf()
        else:
            assert int.from_bytes(pt.affine[:48], "little") == want[0] * (1 << 384) % P and int.from_bytes(pt.affine[48:], "little") == want[1] * (1 << 384) % P
        assert pt.compress() == b
    assert all(p.is_inf() for p in pts[len(good):])
    assert engine.decompress_g1(GEN48).compress() == GEN48
    with pytest.raises(kateth_amd.BlsError, match="NotInGroup"):
        engine.decompress_g1(bad[4])


def test_load_setup_rejects_bad_points_with_the_reference_error():
    """Setup::load_json maps a rejected point to LoadSetupError::Bls(ECGroup(..)) (src/kzg/setup.rs:59-72): the engine
    reports which class through kzg_last_error_code"""
    import kateth_amd
    from oracle.pyref import bls

    raw = json.load(open(TRUSTED_SETUP))
    g1 = [bytes.fromhex(s[2:]) for s in raw["g1_lagrange"]]
    g2 = [bytes.fromhex(s[2:]) for s in raw["g2_monomial"]]
    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    x2 = 1
    while bls._fp_sqrt(x2**3 + 4) is not None:
        x2 += 1
    cases = [
        (7, bls.g1_compress((x, y)), "NotInGroup"),
        (4095, bytes([g1[4095][0] & 0x7F]) + g1[4095][1:], "InvalidEncoding"),
        (0, bytes([0x80]) + x2.to_bytes(48, "big")[1:], "NotOnCurve"),
    ]
    for idx, bad, kind in cases:
        pts = list(g1)
        pts[idx] = bad
        with pytest.raises(kateth_amd.LoadSetupError, match="ECGroupError::" + kind) as e:
            kateth_amd.Setup.from_bytes(pts, g2, window_bits=6)
        assert "g1_lagrange[%d]" % idx in str(e.value)
    q = list(g2)
    q[3] = bytes([g2[3][0] & 0x7F]) + g2[3][1:]
    with pytest.raises(kateth_amd.LoadSetupError, match="ECGroupError::InvalidEncoding") as e:
        kateth_amd.Setup.from_bytes(g1, q, window_bits=6)
    assert "g2_monomial[3]" in str(e.value)


@pytest.mark.parametrize("workload", ["commit", "proof", "verify"])
def test_bench_rank_launcher_two_real_engine_ranks_on_one_card(workload):
    """`python bench.py --gpus 2` starts two rank processes itself (before anything in the launcher touches HIP); here they
    share the one card and exchange through gloo (RCCL needs a GPU per rank): the REAL engine in every rank -- blob-sharded
    commitments all-gathered in rank order, and batch verification through dist.verify_blob_proof_batch_sharded (roots +
    first-error records gathered, global indices for r^i, partial sums gathered, one pairing)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile

    logdir = tempfile.mkdtemp(prefix="bench_ranks_")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", workload, "--batch", "96", "--window-bits", "8",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--rank-logs", logdir]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    # every rank's stdout and stderr are kept (rank 0's stdout is also what was relayed)
    assert sorted(os.listdir(logdir)) == ["rank0.err", "rank0.out", "rank1.err", "rank1.out"]
    assert open(os.path.join(logdir, "rank0.out")).read() == out.stdout
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["blobs_per_gpu"] == 96 and rec["config"]["backend"] == "gloo"
    assert rec["roofline"]["kernel"] == ("k_challenge*" if workload == "verify" else "k_msm_comb30")
    # a rendezvous that disagrees with --gpus must fail loudly instead of silently running one rank
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stdout + bad.stderr)
    # two ranks on ONE card over the nccl backend must be refused before any collective hangs (RCCL needs a GPU per rank)
    if workload == "commit":
        dup = subprocess.run([c if c != "gloo" else "nccl" for c in cmd], capture_output=True, text=True, timeout=300, env=env)
        assert dup.returncode != 0
        errs = "".join(open(os.path.join(logdir, f)).read() for f in ("rank0.err", "rank1.err")) + dup.stderr
        assert "GPUs visible" in errs or "share a GPU" in errs


@pytest.mark.parametrize("workload", ["commit", "verify"])
def test_bench_rccl_calls_with_a_one_rank_group(workload):
    """`bench.py --gpus 1 --dist-single --backend nccl`: a ONE-rank process group over RCCL on the real GPU, so that the calls
    the N-rank path makes -- init_process_group("nccl"), all_gather_object of the device seats, all_gather_into_tensor of the
    48-byte results on the device, the all-gathers of dist.verify_blob_proof_batch_sharded, the MAX all-reduce of the elapsed
    time, barrier -- execute against RCCL itself (every multi-rank run so far went through gloo: a one-GPU box)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dist-single", "--backend", "nccl", "--workload", workload, "--batch", "256",
           "--window-bits", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-live-traffic"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "nccl" and rec["value"] > 0


def test_bench_launcher_single_rank_matches_the_direct_run():
    """`bench.py --gpus 1 --spawn` goes through the launcher (child process, per-rank logs, relayed JSON line); its number must
    be the direct run's within noise -- the launcher adds nothing to the timed region"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "1024", "--window-bits", "8", "--steps", "8", "--warmup", "2",
            "--no-cpu-baseline", "--no-extra"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    vals = {}
    for name, extra in (("direct", []), ("spawn", ["--spawn"])):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        assert rec["n_gpus"] == 1 and rec["config"]["blobs_per_gpu"] == 1024
        vals[name] = rec["value"]
    assert abs(vals["spawn"] / vals["direct"] - 1.0) < 0.10, vals


def test_latency_comb_of_the_class_22_table(engine, torch_cuda, monkeypatch):
    """a class-22 context carries a second, small comb (blocks of 8 points, 64 plane groups) for calls of at most 16 blobs:
    commitments and proofs of 1, 3, 16 (latency comb) and 17 (main comb) blobs must equal the class-8 engine's, with the
    latency comb switched off (KATETH_AMD_LAT_TABLE=0) as well, and an invalid blob keeps its status"""
    import kateth_amd

    torch = torch_cuda
    n = 17
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x1A7, 9, n, d_blobs.data_ptr())
    d_blobs[2 * 131072 + 32 * 100: 2 * 131072 + 32 * 100 + 32] = 0xFF  # blob 2, element 100: not canonical
    want_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    want_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, want_c.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    want_st = st.cpu().tolist()
    assert want_st[2] == 2 and sum(1 for v in want_st if v) == 1
    want_c[2 * 48: 3 * 48] = want_c[0:48]  # a decodable commitment for the bad blob
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), want_c.data_ptr(), n, want_p.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    want_pst = st.cpu().tolist()
    wc, wp = want_c.cpu().numpy().tobytes(), want_p.cpu().numpy().tobytes()
    for env in ({}, {"KATETH_AMD_LAT_TABLE": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        big = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=22)
        try:
            for m in (1, 3, 16, 17):
                c = torch.zeros(m * 48, dtype=torch.uint8, device="cuda")
                p = torch.zeros(m * 48, dtype=torch.uint8, device="cuda")
                s2 = torch.full((m,), -7, dtype=torch.int32, device="cuda")
                big.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, c.data_ptr(), s2.data_ptr())
                torch.cuda.synchronize()
                assert s2.cpu().tolist() == want_st[:m], (env, m)
                got = c.cpu().numpy().tobytes()
                for i in range(m):
                    if i != 2:
                        assert got[48 * i: 48 * i + 48] == wc[48 * i: 48 * i + 48], (env, m, i)
                big.compute_blob_proof_batch_dev(d_blobs.data_ptr(), want_c.data_ptr(), m, p.data_ptr(), s2.data_ptr())
                torch.cuda.synchronize()
                assert s2.cpu().tolist() == want_pst[:m], (env, m)
                gotp = p.cpu().numpy().tobytes()
                for i in range(m):
                    if want_pst[i] == 0:
                        assert gotp[48 * i: 48 * i + 48] == wp[48 * i: 48 * i + 48], (env, m, i)
        finally:
            big.close()


def test_half_wave_mode_on_an_odd_batch(engine, torch_cuda):
    """when two blobs per wave (32 lanes each) is the cheaper launch shape -- 4,095 blobs = 2,048 waves = one round -- an odd
    batch leaves the last wave half empty.  Commitments and proofs of 4,095 blobs equal, item for item, those of the same blobs
    computed in small batches (one blob per wave / several waves per blob), among them 2,050 blobs, which run as four split
    units per blob: five short rounds instead of a second long round for two waves (engine.hip, msm_shape)."""
    torch = torch_cuda
    n = 4095
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    engine.synth_blobs_dev(0x0DD, 0, n, d_blobs.data_ptr())
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    big_c, big_p = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    for first, m in ((0, 3), (2047, 130), (4086, 9), (0, 2050)):  # 2,050 blobs: four split units per blob (five short rounds)
        c2 = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
        p2 = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
        engine.blob_to_commitment_batch_dev(d_blobs.data_ptr() + first * 131072, m, c2.data_ptr(), d_st.data_ptr())
        engine.compute_blob_proof_batch_dev(d_blobs.data_ptr() + first * 131072, c2.data_ptr(), m, p2.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert c2.cpu().numpy().tobytes() == big_c[48 * first:48 * (first + m)], first
        assert p2.cpu().numpy().tobytes() == big_p[48 * first:48 * (first + m)], first
    assert engine.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
