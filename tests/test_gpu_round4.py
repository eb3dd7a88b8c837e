"""Round-4 engine changes, through the C ABI on the GPU:
  * batch verification with its phases interleaved (sorting beside the point decoder, infinity handled in the bucket chains,
    statuses read after the bucket kernels are enqueued) -- src/kzg/setup.rs:115-161, 247-275;
  * commitment / proof calls kept in flight on two streams (workspace slots) -- bit-exact against one call at a time;
  * Polynomial::evaluate (src/kzg/poly.rs:10-33) for more pairs than one staging chunk;
  * the spill-free point decoder at batch size against the oracle's decisions (src/bls.rs:505-531);
  * the measurement aids bench.py prices its roofline with."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, TRUSTED_SETUP  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
INF48 = bytes([0xC0]) + bytes(47)


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
def golden():
    return json.load(open(os.path.join(GOLDEN, "kzg_vectors.json")))


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


@pytest.mark.parametrize("n", [33000, 700])
def test_verify_with_infinity_and_repeated_points(n, engine, torch_cuda):
    """the flat variable-base MSM (both lincombs from 32,768 terms on) and the classic one: a batch in which several blobs are
    all-zero (commitment = proof = the point at infinity, src/bls.rs:505-531 decodes 0xc0..: the sorting kernels no longer see
    the decoder's infinity flags, the bucket chains skip the zero entries), several triples are repeated (P + P in a bucket),
    and one constant blob has an infinity PROOF with a finite commitment.  true; then one finite proof replaced by infinity ->
    false; then a commitment that is not on the curve -> the reference's error although the bucket kernels had been enqueued."""
    import kateth_amd

    torch = torch_cuda
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0x1F1F + n)
    zero_items = [0, 5, n // 2, n - 1]
    for i in zero_items:
        d_blobs[i * 131072:(i + 1) * 131072] = 0
        d_c[i * 48:(i + 1) * 48] = torch.frombuffer(bytearray(INF48), dtype=torch.uint8).cuda()
        d_p[i * 48:(i + 1) * 48] = torch.frombuffer(bytearray(INF48), dtype=torch.uint8).cuda()
    # a constant polynomial: every element 7 -> commitment [7]G (finite), proof = infinity
    const_blob = torch.frombuffer(bytearray((7).to_bytes(32, "big") * 4096), dtype=torch.uint8).cuda()
    d_blobs[9 * 131072:10 * 131072] = const_blob
    c9 = engine.blob_to_commitment(bytes(const_blob.cpu().numpy().tobytes()))
    d_c[9 * 48:10 * 48] = torch.frombuffer(bytearray(c9), dtype=torch.uint8).cuda()
    d_p[9 * 48:10 * 48] = torch.frombuffer(bytearray(INF48), dtype=torch.uint8).cuda()
    # repeated triples: items 20..23 are copies of item 19
    for i in range(20, 24):
        d_blobs[i * 131072:(i + 1) * 131072] = d_blobs[19 * 131072:20 * 131072]
        d_c[i * 48:(i + 1) * 48] = d_c[19 * 48:20 * 48]
        d_p[i * 48:(i + 1) * 48] = d_p[19 * 48:20 * 48]
    torch.cuda.synchronize()
    args = (d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
    assert engine.verify_blob_proof_batch_dev(*args) is True
    # the sharded entry points (phase 1 complete, then phase 2) take the other order of the same pieces
    sess, root, err6 = engine.verify_phase1_dev(*args)
    assert err6[0] == err6[2] == err6[4] == -1
    part = engine.verify_phase2_dev(sess, root, 0, n)
    engine.verify_session_destroy(sess)
    assert engine.verify_batch_finish(part) is True
    keep = d_p[30 * 48:31 * 48].clone()
    d_p[30 * 48:31 * 48] = torch.frombuffer(bytearray(INF48), dtype=torch.uint8).cuda()
    assert engine.verify_blob_proof_batch_dev(*args) is False
    d_p[30 * 48:31 * 48] = keep
    assert engine.verify_blob_proof_batch_dev(*args) is True
    # x = 5 is not on the curve / not in the group: whichever, it is commitment n - 3's error and it wins over nothing else
    bad = bytes([0x80]) + bytes(46) + bytes([5])
    d_c[(n - 3) * 48:(n - 2) * 48] = torch.frombuffer(bytearray(bad), dtype=torch.uint8).cuda()
    with pytest.raises(kateth_amd.KzgError) as err:
        engine.verify_blob_proof_batch_dev(*args)
    want = engine.decompress_g1_batch(bad)[1][0]
    assert want in (4, 5) and err.value.inner.inner.kind == {4: "NotOnCurve", 5: "NotInGroup"}[want]
    sess, _, err6 = engine.verify_phase1_dev(*args)
    engine.verify_session_destroy(sess)
    assert err6[2] == n - 3 and err6[3] == want and err6[0] == -1 and err6[4] == -1
    # the session pool is reusable after a rejected call
    d_c[(n - 3) * 48:(n - 2) * 48] = d_c[19 * 48:20 * 48]
    assert engine.verify_blob_proof_batch_dev(*args) is False  # wrong commitment, well-formed
    del d_blobs, d_c, d_p
    torch.cuda.empty_cache()


def test_host_buffer_verify_takes_the_fused_path(engine, torch_cuda):
    """kzg_verify_blob_proof_batch from host buffers (chunked staging, then the interleaved phases): same answers as the device
    entry point, for a multi-chunk batch and for single items"""
    torch = torch_cuda
    n = 1300
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0xB0B)
    hb, hc, hp = (t.cpu().numpy().tobytes() for t in (d_blobs, d_c, d_p))
    assert engine.verify_blob_proof_batch_host(hb, hc, hp, n) is True
    assert engine.verify_blob_proof_batch_host(hb, hc, hp[48:] + hp[:48], n) is False
    assert engine.verify_blob_proof(hb[:131072], hc[:48], hp[:48]) is True
    assert engine.verify_blob_proof(hb[:131072], hc[:48], hp[48:96]) is False
    assert engine.verify_blob_proof(bytes(131072), INF48, INF48) is True


def test_calls_in_flight_on_several_streams_are_bit_exact(engine, torch_cuda, golden):
    """successive commitment / proof calls take the context's workspace slots (three) in turn: calls enqueued on two, then four
    streams (one more than there are slots: a call waits for the slot's previous user) and on one, with no synchronisation in
    between -- 20 of them, alternating sizes so that the slots are regrown under way -- return what one call at a time returns"""
    torch = torch_cuda
    n = 600
    d_blobs, d_c, d_p = _triples(engine, torch, n, golden["seed"])
    hc, hp = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    for rec in golden["blobs"]:
        assert hc[48 * rec["index"]:48 * rec["index"] + 48].hex() == rec["commitment"]
    streams = [torch.cuda.Stream() for _ in range(4)]
    sizes = [600, 37, 512, 600, 1, 300, 600, 64, 600, 129, 600, 600, 600, 600, 450, 600, 2, 600, 600, 77]
    outs = []
    for k, m in enumerate(sizes):
        st = streams[k % 2] if k < 8 else streams[0] if k < 12 else streams[k % 4]
        with torch.cuda.stream(st):  # the fills run on the call's own stream (torch's pool streams do not order against its default stream)
            o = torch.zeros(m * 48, dtype=torch.uint8, device="cuda")
            s = torch.full((m,), -7, dtype=torch.int32, device="cuda")
            if k % 3 == 2:
                engine.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, o.data_ptr(), s.data_ptr(), st.cuda_stream)
                outs.append((o, s, hc[: 48 * m], k))
            else:
                engine.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), m, o.data_ptr(), s.data_ptr(), st.cuda_stream)
                outs.append((o, s, hp[: 48 * m], k))
    torch.cuda.synchronize()
    for o, s, want, k in outs:
        assert int(s.abs().sum()) == 0, k
        assert o.cpu().numpy().tobytes() == want, k


def test_evaluate_blobs_across_staging_chunks(engine, torch_cuda):
    """kzg_evaluate_blobs streams the blobs through the staging arena in chunks of 2,048: 2,100 pairs, evaluated at each blob's
    own Fiat-Shamir challenge, must equal the evaluations batch verification computed for the same blobs (session introspection),
    and a rejected item beyond the first chunk keeps its place"""
    torch = torch_cuda
    n = 2100
    d_blobs, d_c, d_p = _triples(engine, torch, n, 0xE7A2)
    sess, _, err6 = engine.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
    z, y = engine.verify_session_zy(sess, 0, n)
    engine.verify_session_destroy(sess)
    hb = d_blobs.cpu().numpy().tobytes()
    ys, st = engine.evaluate_blobs(hb, z)
    assert st == [0] * n and ys == y
    bad = bytearray(hb[: 2060 * 131072])
    bad[2055 * 131072 + 64:2055 * 131072 + 96] = R.to_bytes(32, "big")
    ys2, st2 = engine.evaluate_blobs(bytes(bad), z[: 2060 * 32])
    assert st2[2055] == 2 and sum(1 for v in st2 if v) == 1
    assert ys2[: 2055 * 32] == y[: 2055 * 32] and ys2[2055 * 32:2056 * 32] == bytes(32) and ys2[2056 * 32:] == y[2056 * 32:2060 * 32]


def test_point_decoder_at_batch_size_matches_the_oracle_decisions(engine):
    """P1::decompress (src/bls.rs:505-531) for 4,096 encodings in one call -- valid points, x with no square root, on-curve points
    outside the subgroup, x >= p, flag errors, infinity with stray bits: the engine's status equals the oracle's decision for
    every one, and valid points round-trip through P1.compress"""
    import random

    import kateth_amd
    from oracle.pyref import bls

    rng = random.Random(0xDEC0DE)
    raw = json.load(open(TRUSTED_SETUP))
    valid = [bytes.fromhex(s[2:]) for s in raw["g1_lagrange"][:512]]
    items, want = [], []
    for k in range(4096):
        kind = k % 8
        if kind in (0, 1, 2):
            enc = valid[rng.randrange(512)]
        elif kind == 3:  # random x: about half have no square root, the rest are almost surely outside the subgroup
            enc = bytearray(rng.randrange(bls.P).to_bytes(48, "big"))
            enc[0] = (enc[0] & 0x1F) | 0x80 | (0x20 if rng.random() < 0.5 else 0)
            enc = bytes(enc)
        elif kind == 4:  # flip the sign flag of a valid point: still valid (the other root)
            v = bytearray(valid[rng.randrange(512)])
            v[0] ^= 0x20
            enc = bytes(v)
        elif kind == 5:  # x >= p (p's top byte is 0x1a)
            enc = bytes([0x9F]) + bytes(rng.randrange(256) for _ in range(47)) if k % 16 else bytes([0x80 | (bls.P >> 376)]) + bls.P.to_bytes(48, "big")[1:]
        elif kind == 6:  # compression flag clear
            v = bytearray(valid[rng.randrange(512)])
            v[0] &= 0x7F
            enc = bytes(v)
        else:  # infinity, sometimes with a stray bit
            enc = bytes([0xC0]) + bytes(46) + bytes([1 if k % 16 == 15 else 0])
        items.append(enc)
        try:
            bls.g1_decompress(enc)
            want.append(0)
        except bls.ECGroupError as err:
            want.append({"InvalidEncoding": 3, "NotOnCurve": 4, "NotInGroup": 5}[err.kind])
    pts, status = engine.decompress_g1_batch(items)
    assert status == want
    assert set(want) >= {0, 3, 4, 5}
    for k in range(0, 4096, 97):
        if want[k] == 0:
            assert pts[k].compress() == items[k]
    assert isinstance(pts[0], kateth_amd.P1)


def test_measurement_aids(engine, torch_cuda):
    """kzg_microbench_valu_issue and the clock probe answer with finite, positive figures.  bench.py prices SQ_INSTS_VALU with them
    at the start of a process of its own, where they have been stable (4.125 cycles per instruction, 2.0-2.3 GHz).  Late in THIS
    suite's run -- minutes of other contexts, an idle stretch while the previous test runs the Python oracle -- the same calls have
    returned 2.5, 2.7 and 115 cycles per instruction, in this process and in a fresh child process alike, so the bounds here are
    those of a smoke test, not of the measurement (the readings go to gpurun_out/measurement_aids_in_suite.json)."""
    import math

    torch_cuda.cuda.synchronize()
    cyc, ghz = engine.microbench_valu_issue(2, 4000)
    cyc1, ghz1 = engine.microbench_valu_issue(1, 4000)
    engine.clock_probe_launch(20000)
    mean, lo, hi = engine.clock_probe_read()
    got = {"two_waves": [cyc, ghz], "one_wave": [cyc1, ghz1], "probe_ghz_mean_lo_hi": [mean, lo, hi]}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):  # what they said this time (the GPU box merges the directory back)
        json.dump(got, open(os.path.join(out, "measurement_aids_in_suite.json"), "w"))
    for v in (cyc, ghz, cyc1, ghz1, mean, lo, hi):
        assert math.isfinite(v) and v >= 0.0, got
    assert cyc > 0.0 and cyc1 > 0.0 and hi > 0.0, got


def test_out_of_memory_at_call_time_is_reported_and_recoverable(torch_cuda, golden):
    """a call whose workspace cannot be allocated (the card is filled up first) fails with KZG_FAIL_HIP -- and the failure does not
    leak into the next call: a small call on the same thread, with the card still full, succeeds (the runtime's last-error slot is
    cleared when a failure is reported), and once memory is back the large call does too"""
    import kateth_amd

    torch = torch_cuda
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)  # a context of its own: its workspace has not grown yet
    try:
        n = 8192
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        s.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
        d_c = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
        s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), 2, d_c.data_ptr(), d_st.data_ptr())  # the workspace of a 2-blob call exists now
        torch.cuda.synchronize()
        small = d_c[:96].cpu().numpy().tobytes()
        assert small[:48].hex() == golden["blobs"][0]["commitment"]
        torch.cuda.empty_cache()
        # the ballast comes straight from hipMalloc: torch's caching allocator would serve part of it from segments it already
        # holds (in a long test process it does) and leave the DEVICE with more free memory than asked for
        import ctypes

        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        ballast = []
        for _ in range(8):  # until 0.6 GiB are left (the runtime's "free" figure moves by more than one allocation's size): two 1.1-GiB workspaces cannot grow
            free = torch.cuda.mem_get_info()[0]
            if free < 700 << 20:
                break
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(free - (600 << 20))) == 0
            ballast.append(p)
        assert torch.cuda.mem_get_info()[0] < 700 << 20
        try:
            with pytest.raises(kateth_amd.kzg.EngineError, match="out of memory|hipMalloc|HIP"):
                s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
            d_c[:96] = 0
            s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), 2, d_c.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
            assert d_c[:96].cpu().numpy().tobytes() == small and int(d_st[:2].abs().sum()) == 0
        finally:
            for p in ballast:
                assert hip.hipFree(p) == 0
        s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        assert int(d_st.abs().sum()) == 0 and d_c[:96].cpu().numpy().tobytes() == small
        for rec in golden["blobs"]:
            assert d_c[48 * rec["index"]:48 * rec["index"] + 48].cpu().numpy().tobytes().hex() == rec["commitment"]
    finally:
        s.close()
        torch.cuda.empty_cache()


def test_single_item_verification_host_lincomb_agrees_with_the_batch_machinery(engine, torch_cuda, golden, monkeypatch):
    """n = 1 (verify_blob_proof, verify_proof -- src/kzg/setup.rs:96-113, 208-221) ends on the host: commitment - [y]G + [z]proof as
    one double-scalar multiplication, then the pairing.  KATETH_AMD_SINGLE_VIA_BATCH sends the same calls through the batch
    machinery (transcript, two variable-base MSMs): same booleans and same errors on valid triples, a wrong proof, a wrong y, a
    constant blob (proof = infinity), the zero blob (commitment = proof = infinity) and rejected encodings."""
    import kateth_amd

    torch = torch_cuda
    monkeypatch.setenv("KATETH_AMD_SINGLE_VIA_BATCH", "1")
    batch = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    monkeypatch.delenv("KATETH_AMD_SINGLE_VIA_BATCH")
    try:
        n = 3
        d_blobs, d_c, d_p = _triples(engine, torch, n, golden["seed"])
        hb, hc, hp = (t.cpu().numpy().tobytes() for t in (d_blobs, d_c, d_p))
        blobs = [hb[i * 131072:(i + 1) * 131072] for i in range(n)]
        cs = [hc[i * 48:(i + 1) * 48] for i in range(n)]
        ps = [hp[i * 48:(i + 1) * 48] for i in range(n)]
        const_blob = (7).to_bytes(32, "big") * 4096
        c7 = engine.blob_to_commitment(const_blob)
        cases = [(blobs[0], cs[0], ps[0], True), (blobs[1], cs[1], ps[1], True), (blobs[0], cs[0], ps[1], False), (blobs[0], cs[1], ps[0], False),
                 (const_blob, c7, INF48, True), (const_blob, c7, ps[0], False), (bytes(131072), INF48, INF48, True), (bytes(131072), INF48, ps[2], False)]
        for blob, c, p, want in cases:
            assert engine.verify_blob_proof(blob, c, p) is want
            assert batch.verify_blob_proof(blob, c, p) is want
            d = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in (blob, c, p)]
            assert engine.verify_blob_proof_batch_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 1) is want
        import random

        rnd = random.Random(0x0E17)
        for _ in range(24):  # every combination of the three triples' parts, and a blob with one element changed
            i, j, k = (rnd.randrange(n) for _ in range(3))
            blob = blobs[i]
            tampered = rnd.random() < 0.25
            if tampered:
                e = rnd.randrange(4096)
                blob = blob[:32 * e] + (1).to_bytes(32, "big") + blob[32 * e + 32:]
            want = (i == j == k) and not tampered
            assert engine.verify_blob_proof(blob, cs[j], ps[k]) is want
            assert batch.verify_blob_proof(blob, cs[j], ps[k]) is want
        # verify_proof: (z, y) from the engine's own proof at a caller's point
        z = (0x1234567).to_bytes(32, "big")
        prf, y = engine.proof(blobs[0], z)
        y_bad = ((int.from_bytes(y, "big") + 1) % R).to_bytes(32, "big")
        for s_ in (engine, batch):
            assert s_.verify_proof(prf, cs[0], z, y) is True
            assert s_.verify_proof(prf, cs[0], z, y_bad) is False
            assert s_.verify_proof(prf, cs[1], z, y) is False
            assert s_.verify_proof(INF48, c7, z, (7).to_bytes(32, "big")) is True  # a constant polynomial: y = 7 everywhere, proof = infinity
        bad = bytes([0x80]) + bytes(46) + bytes([5])  # not on the curve / not in the group
        for s_ in (engine, batch):
            for args in ((blobs[0], bad, ps[0]), (blobs[0], cs[0], bad)):
                with pytest.raises(kateth_amd.KzgError):
                    s_.verify_blob_proof(*args)
            with pytest.raises(kateth_amd.KzgError):
                s_.verify_proof(bad, cs[0], z, y)
            with pytest.raises(kateth_amd.KzgError):
                s_.verify_blob_proof(R.to_bytes(32, "big") + blobs[0][32:], cs[0], ps[0])
    finally:
        batch.close()
