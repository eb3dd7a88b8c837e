"""GROUP contexts (kzg_config.devices / ndev: the multi-GPU split behind the C ABI), configuration
precedence and the background table build -- through the C ABI, on ONE card: a group may list the same ordinal more than once
(two members = two complete class-8 contexts on the card), which exercises every line of the sharding, the first-error merge
and the phase1 / roots / phase2 / finish protocol that a node with 8 MI355X runs with 8 different ordinals.

Reference shape: blobs are independent (src/kzg/setup.rs:235-242), errors are first-wins in the order blobs, commitments,
proofs (src/kzg/setup.rs:259-271), one pairing check per batch (src/kzg/setup.rs:152-160)."""
import ctypes
import json
import os
import subprocess
import threading

import pytest

pytestmark = pytest.mark.gpu

from conftest import GOLDEN, ROOT, TRUSTED_SETUP  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "kzg_vectors.json")))


@pytest.fixture(scope="module")
def single():
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8)
    yield s
    s.close()


@pytest.fixture(scope="module")
def group2():
    """two members on device 0"""
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0, 0])
    yield s
    s.close()


@pytest.fixture(scope="module")
def group3():
    import kateth_amd

    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0, 0, 0])
    yield s
    s.close()


@pytest.fixture(scope="module")
def triples(single, torch_cuda, golden):
    """37 synthetic (blob, commitment, proof) triples from the single-device context (odd: ragged shares)"""
    torch = torch_cuda
    n = 37
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    single.synth_blobs_dev(golden["seed"], 0, n, d_blobs.data_ptr())
    torch.cuda.synchronize()
    hb = d_blobs.cpu().numpy().tobytes()
    hc, st = single.blob_to_commitment_batch(hb)
    assert st == [0] * n
    hp, st = single.compute_blob_proof_batch(hb, hc)
    assert st == [0] * n
    for rec in golden["blobs"]:
        if rec["index"] < n:
            assert hc[48 * rec["index"]:48 * rec["index"] + 48].hex() == rec["commitment"]
            assert hp[48 * rec["index"]:48 * rec["index"] + 48].hex() == rec["proof"]
    return hb, hc, hp, n


def _split(buf, size):
    return [buf[i:i + size] for i in range(0, len(buf), size)]


def test_group_introspection(single, group2, group3):
    assert single.members == 1 and single.member_device(0) == 0
    assert group2.members == 2 and [group2.member_device(k) for k in range(2)] == [0, 0]
    assert group3.members == 3
    assert group2.window_bits == 8 and group2.member(1).window_bits == 8 and group2.member(1).table_bytes == single.table_bytes
    with pytest.raises(IndexError):
        group2.member(2)


def test_group_of_one_is_the_single_device_context(triples):
    import kateth_amd

    hb, hc, hp, n = triples
    g = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0])
    try:
        assert g.members == 1
        assert g.blob_to_commitment_batch(hb[: 5 * 131072]) == (hc[: 5 * 48], [0] * 5)
        assert g.compute_blob_proof_batch(hb[: 5 * 131072], hc[: 5 * 48]) == (hp[: 5 * 48], [0] * 5)
        assert g.verify_blob_proof_batch(_split(hb, 131072)[:5], _split(hc, 48)[:5], _split(hp, 48)[:5]) is True
    finally:
        g.close()
    ga = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices="all")  # KZG_ALL_DEVICES: every visible device (one here)
    try:
        import torch

        assert ga.members == torch.cuda.device_count()
        assert ga.blob_to_commitment(hb[:131072]) == hc[:48]
    finally:
        ga.close()


@pytest.mark.parametrize("which", ["group2", "group3"])
def test_group_producers_are_bit_exact(which, request, single, triples, golden):
    """commitments, blob proofs and proofs at points over 2 and 3 members == the single-device context, for batch sizes
    that split evenly, raggedly, into one item per member, and into fewer items than members"""
    g = request.getfixturevalue(which)
    hb, hc, hp, n = triples
    zs = b"".join(bytes.fromhex(golden["blobs"][0]["kzg_proof_at"]["z"]) for _ in range(n))
    want_pz, want_y, st = single.compute_proof_batch(hb[: 7 * 131072], zs[: 7 * 32])
    assert st == [0] * 7
    for m in (n, 8, 3, 2, 1):
        assert g.blob_to_commitment_batch(hb[: m * 131072]) == (hc[: m * 48], [0] * m), m
        assert g.compute_blob_proof_batch(hb[: m * 131072], hc[: m * 48]) == (hp[: m * 48], [0] * m), m
    assert g.compute_proof_batch(hb[: 7 * 131072], zs[: 7 * 32]) == (want_pz, want_y, [0] * 7)
    # the point-returning forms (Commitment = Proof = P1, src/kzg/mod.rs:9-10)
    assert g.blob_to_commitment_batch_affine(hb[: 9 * 131072]) == single.blob_to_commitment_batch_affine(hb[: 9 * 131072])
    assert g.compute_blob_proof_batch_affine(hb[: 9 * 131072], hc[: 9 * 48]) == single.compute_blob_proof_batch_affine(hb[: 9 * 131072], hc[: 9 * 48])
    # reference-shaped single-item methods (served by a rotating member)
    for i in range(5):
        assert g.blob_to_commitment(hb[i * 131072:(i + 1) * 131072]) == hc[48 * i:48 * i + 48]
        assert g.blob_proof(hb[i * 131072:(i + 1) * 131072], hc[48 * i:48 * i + 48]) == hp[48 * i:48 * i + 48]
    assert g.decompress_g1_batch(hc[: 11 * 48]) == single.decompress_g1_batch(hc[: 11 * 48])
    assert g.evaluate_blobs(hb[: 5 * 131072], zs[: 5 * 32]) == single.evaluate_blobs(hb[: 5 * 131072], zs[: 5 * 32])


def test_group_rejected_items_land_in_every_share(group2, group3, single, triples):
    """a non-canonical element in one blob of EACH member's range, a bad commitment in another: per-item statuses and zeroed
    outputs equal the single-device call's"""
    hb, hc, hp, n = triples
    m = 12
    blobs = bytearray(hb[: m * 131072])
    for b in (1, 5, 10):  # group2 shares [0,6) [6,12); group3 shares [0,4) [4,8) [8,12)
        blobs[b * 131072 + 64:b * 131072 + 96] = R.to_bytes(32, "big")
    coms = bytearray(hc[: m * 48])
    coms[3 * 48] &= 0x7F  # not compressed -> InvalidEncoding
    coms[9 * 48:10 * 48] = bytes([0x80]) + bytes(46) + bytes([5])  # x = 5: not on the curve or not in the group
    want_c = single.blob_to_commitment_batch(bytes(blobs))
    want_p = single.compute_blob_proof_batch(bytes(blobs), bytes(coms))
    assert [s for s in want_c[1] if s] == [2, 2, 2] and sum(1 for s in want_p[1] if s) == 5
    for g in (group2, group3):
        assert g.blob_to_commitment_batch(bytes(blobs)) == want_c
        assert g.compute_blob_proof_batch(bytes(blobs), bytes(coms)) == want_p


@pytest.mark.parametrize("which", ["group2", "group3"])
def test_group_verification_matches_the_single_device_call(which, request, single, triples):
    """true / false / first-error order (blobs before commitments before proofs, lowest GLOBAL index within a kind --
    src/kzg/setup.rs:259-271) with the offending items spread over different members"""
    import kateth_amd

    g = request.getfixturevalue(which)
    hb, hc, hp, n = triples
    blobs, cs, ps = _split(hb, 131072), _split(hc, 48), _split(hp, 48)
    for m in (n, 12, 4, 3, 2, 1):
        assert g.verify_blob_proof_batch(blobs[:m], cs[:m], ps[:m]) is True, m
    assert g.verify_blob_proof_batch([], [], []) is True
    # one wrong proof in the LAST share, one swapped pair across a share boundary
    assert g.verify_blob_proof_batch(blobs, cs, ps[:-1] + [ps[0]]) is False
    assert g.verify_blob_proof_batch(blobs[:12], cs[:12], ps[:5] + [ps[6], ps[5]] + ps[7:12]) is False
    assert single.verify_blob_proof_batch(blobs[:12], cs[:12], ps[:5] + [ps[6], ps[5]] + ps[7:12]) is False
    bad_blob = bytearray(blobs[10])
    bad_blob[0:32] = R.to_bytes(32, "big")
    not_compressed = bytes([cs[1][0] & 0x7F]) + cs[1][1:]
    off_curve = bytes([0x80]) + bytes(46) + bytes([5])

    def outcome(ctx, bl, c, p):
        try:
            return ctx.verify_blob_proof_batch(bl, c, p)
        except kateth_amd.KzgError as err:
            return str(err)

    cases = [
        # a blob error in the last share beats a commitment error in the first
        (blobs[:10] + [bytes(bad_blob)] + blobs[11:12], [cs[0], not_compressed] + cs[2:12], ps[:12]),
        # a commitment error in the last share beats a proof error in the first
        (blobs[:12], cs[:11] + [not_compressed], [off_curve] + ps[1:12]),
        # two proof errors: the lower global index wins although it is another member's
        (blobs[:12], cs[:12], ps[:2] + [off_curve] + ps[3:9] + [not_compressed] + ps[10:12]),
        # two blob errors in different shares
        (blobs[:3] + [bytes(bad_blob)] + blobs[4:9] + [bytes(bad_blob)] + blobs[10:12], cs[:12], ps[:12]),
    ]
    for bl, c, p in cases:
        want = outcome(single, bl, c, p)
        assert isinstance(want, str)
        assert outcome(g, bl, c, p) == want
    assert "InvalidFieldElement" in outcome(g, *cases[0]) and "InvalidEncoding" in outcome(g, *cases[1])
    # single-item verifiers on a group
    assert g.verify_blob_proof(blobs[2], cs[2], ps[2]) is True and g.verify_blob_proof(blobs[2], cs[2], ps[3]) is False


def test_group_verify_proof_single(group2, golden):
    rec = golden["blobs"][0]
    at = rec["kzg_proof_at"]
    args = [bytes.fromhex(at["proof"]), bytes.fromhex(rec["commitment"]), bytes.fromhex(at["z"]), bytes.fromhex(at["y"])]
    for _ in range(3):  # rotates over the members
        assert group2.verify_proof(*args) is True
    assert group2.verify_proof(args[0], args[1], args[2], (int.from_bytes(args[3], "big") ^ 1).to_bytes(32, "big")) is False


def test_group_device_entry_points_act_on_member_zero(group2, single, triples, torch_cuda, golden):
    torch = torch_cuda
    hb, hc, hp, n = triples
    m = 6
    d_blobs = torch.frombuffer(bytearray(hb[: m * 131072]), dtype=torch.uint8).cuda()
    d_c = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(m, dtype=torch.int32, device="cuda")
    group2.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert d_c.cpu().numpy().tobytes() == hc[: m * 48]
    group2.member(1).blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, d_c.data_ptr(), d_st.data_ptr())  # a borrowed member: data resident on ITS device
    torch.cuda.synchronize()
    assert d_c.cpu().numpy().tobytes() == hc[: m * 48]


def test_group_concurrent_callers(group3, triples):
    """`Arc<Setup>` + `&self` (src/kzg/setup.rs:323): four host threads on one group context, batch and single-item calls mixed"""
    hb, hc, hp, n = triples
    blobs, cs, ps = _split(hb, 131072), _split(hc, 48), _split(hp, 48)
    errors = []

    def worker(tid):
        try:
            for it in range(5):
                k = (tid + it) % 4
                if k == 0:
                    assert group3.blob_to_commitment_batch(hb[: 10 * 131072]) == (hc[: 10 * 48], [0] * 10)
                elif k == 1:
                    assert group3.compute_blob_proof_batch(hb[: 7 * 131072], hc[: 7 * 48]) == (hp[: 7 * 48], [0] * 7)
                elif k == 2:
                    assert group3.verify_blob_proof_batch(blobs[:11], cs[:11], ps[:11]) is True
                    assert group3.verify_blob_proof_batch(blobs[:11], cs[:11], ps[1:12]) is False
                else:
                    assert group3.blob_to_commitment(blobs[tid]) == cs[tid]
                    assert group3.verify_blob_proof(blobs[tid], cs[tid], ps[tid]) is True
        except BaseException as err:  # noqa: BLE001
            errors.append((tid, repr(err)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


def test_group_creation_errors():
    import kateth_amd
    from kateth_amd import kzg

    with pytest.raises(kateth_amd.kzg.EngineError, match="out of range"):
        kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, devices=[0, 99])
    # malformed configuration structs are refused, not guessed at: an unknown flag bit, a reserved field in use, a device list
    # announced but not given
    lib = kzg.load_library()
    raw0 = json.load(open(TRUSTED_SETUP))
    g1b = b"".join(bytes.fromhex(x[2:]) for x in raw0["g1_lagrange"])
    g2b = b"".join(bytes.fromhex(x[2:]) for x in raw0["g2_monomial"])
    for cfg, what in ((kzg._Config.new(0, 8, 0x40, 0, 0, None, 0, 0), b"flags"), (kzg._Config.new(0, 8, 0, 0, 0, None, 0, 7), b"reserved"),
                      (kzg._Config.new(0, 8, 0, 0, 0, None, 2, 0), b"devices"), (kzg._Config.new(0, 8, 0, 3, 0, None, 0, 0), b"plane groups")):
        out = ctypes.c_void_p()
        assert lib.kzg_ctx_create(g1b, g2b, ctypes.byref(cfg), ctypes.byref(out)) == -1 and not out.value  # KZG_FAIL_ARGUMENT
        assert what in lib.kzg_last_error()
    # a struct of another size (a caller compiled against another revision of the header) is refused before any field is believed
    for size in (0, 16, 40, 56):
        cfg = kzg._Config.new(0, 8, 0, 0, 0, None, 0, 0)
        cfg.struct_size = size
        out = ctypes.c_void_p()
        assert lib.kzg_ctx_create(g1b, g2b, ctypes.byref(cfg), ctypes.byref(out)) == -1 and not out.value
        assert b"struct_size" in lib.kzg_last_error()
    # a rejected setup point is reported like the single-device creation reports it (LoadSetupError::Bls, src/kzg/setup.rs:59-64)
    raw = json.load(open(TRUSTED_SETUP))
    g1 = [bytes.fromhex(s[2:]) for s in raw["g1_lagrange"]]
    g2 = [bytes.fromhex(s[2:]) for s in raw["g2_monomial"]]
    g1[7] = bytes([g1[7][0] & 0x7F]) + g1[7][1:]
    with pytest.raises(kateth_amd.LoadSetupError, match="InvalidEncoding"):
        kzg.Setup.from_bytes(g1, g2, window_bits=8, devices=[0, 0])


def test_group_through_the_c_abi_from_c(tmp_path, triples):
    """a C program: kzg_ctx_create_multi with the ordinal listed twice, batch commitment + proof + verification through the
    plain host-buffer entry points, compared with a single-device context created in the same process"""
    from kateth_amd import kzg

    hb, hc, hp, n = triples
    raw = json.load(open(TRUSTED_SETUP))
    (tmp_path / "g1.bin").write_bytes(b"".join(bytes.fromhex(s[2:]) for s in raw["g1_lagrange"]))
    (tmp_path / "g2.bin").write_bytes(b"".join(bytes.fromhex(s[2:]) for s in raw["g2_monomial"]))
    m = 9
    (tmp_path / "blobs.bin").write_bytes(hb[: m * 131072])
    src = tmp_path / "group.c"
    src.write_text(r'''
#include "kateth_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned char* slurp(const char* p, size_t n) { FILE* f = fopen(p, "rb"); unsigned char* b = malloc(n); if (!f || fread(b, 1, n, f) != n) exit(9); fclose(f); return b; }
int main(int argc, char** argv) {
  (void)argc;
  const size_t m = 9;
  unsigned char *g1 = slurp(argv[1], 4096 * 48), *g2 = slurp(argv[2], 65 * 96), *blobs = slurp(argv[3], m * KZG_BYTES_PER_BLOB);
  const int32_t devices[2] = {0, 0};
  kzg_config cfg = KZG_CONFIG_INIT;
  cfg.window_bits = 8;
  kzg_ctx *group = NULL, *one = NULL;
  if (kzg_ctx_create_multi(g1, g2, devices, 2, &cfg, &group) != 0) { printf("create_multi: %s\n", kzg_last_error()); return 1; }
  if (kzg_ctx_create(g1, g2, &cfg, &one) != 0) { printf("create: %s\n", kzg_last_error()); return 1; }
  if (kzg_ctx_members(group) != 2 || kzg_ctx_members(one) != 1 || kzg_ctx_member(one, 0) != one || kzg_ctx_member_device(group, 1) != 0) return 2;
  unsigned char c1[9 * 48], c2[9 * 48], p1[9 * 48], p2[9 * 48];
  int32_t s1[9], s2[9], ok = 0;
  if (kzg_blob_to_commitment_batch(group, blobs, m, c1, s1) || kzg_blob_to_commitment_batch(one, blobs, m, c2, s2)) return 3;
  if (memcmp(c1, c2, sizeof c1) || memcmp(s1, s2, sizeof s1)) return 4;
  if (kzg_compute_blob_proof_batch(group, blobs, c1, m, p1, s1) || kzg_compute_blob_proof_batch(one, blobs, c2, m, p2, s2)) return 5;
  if (memcmp(p1, p2, sizeof p1)) return 6;
  if (kzg_verify_blob_proof_batch(group, blobs, c1, p1, m, &ok) != 0 || ok != 1) return 7;
  memcpy(p1 + 48 * 8, p1, 48);  /* a wrong proof in the second member's share */
  if (kzg_verify_blob_proof_batch(group, blobs, c1, p1, m, &ok) != 0 || ok != 0) return 8;
  c1[48 * 7] &= 0x7f;           /* a malformed commitment in the second member's share: first error = InvalidEncoding */
  if (kzg_verify_blob_proof_batch(group, blobs, c1, p1, m, &ok) != KZG_ERR_EC_INVALID_ENCODING) return 10;
  for (size_t i = 0; i < 48; i++) printf("%02x", c2[i]);
  printf("\n");
  kzg_ctx_destroy(group);
  kzg_ctx_destroy(one);
  return 0;
}
''')
    exe = str(tmp_path / "group")
    hip = "/opt/rocm/lib/libamdhip64.so"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, kzg.library_path(), hip,
                           "-Wl,-rpath," + os.path.dirname(kzg.library_path()), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path / "g1.bin"), str(tmp_path / "g2.bin"), str(tmp_path / "blobs.bin")], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == hc[:48].hex()


# ---------------------------------------------------------------------------------------------------------------------
# configuration precedence and budget
# ---------------------------------------------------------------------------------------------------------------------
def _create(cfg_fields, env=None):
    """(window_bits, plane_groups, table_bytes) of a context created in a child process with `env` set (the knobs are read at
    kzg_ctx_create; a child keeps this process's environment clean)"""
    code = ("import sys, json; sys.path.insert(0, %r); import kateth_amd\n"
            "s = kateth_amd.Setup.load_json(%r, **%r)\n"
            "print(json.dumps([s.window_bits, s.plane_groups, s.table_bytes])); s.close()\n") % (ROOT, TRUSTED_SETUP, cfg_fields)
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(["python3", "-c", code], capture_output=True, text=True, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    return tuple(json.loads(out.stdout.strip().splitlines()[-1]))


def test_explicit_configuration_beats_the_environment():
    """one rule (include/kateth_amd.h): a non-zero kzg_config field beats the environment, which beats the automatic choice"""
    t8_g16 = 16 * 64 * (8 << 7) * 96
    assert _create({"window_bits": 8, "plane_groups": 16}, {"KATETH_AMD_COMB_GROUPS": "4", "KATETH_AMD_WINDOW_BITS": "4"}) == (8, 16, t8_g16)
    assert _create({"window_bits": 8}, {"KATETH_AMD_COMB_GROUPS": "4"}) == (8, 4, t8_g16 // 4)   # the field is 0: the environment fills it in
    assert _create({"plane_groups": 8}, {"KATETH_AMD_WINDOW_BITS": "4"}) == (4, 8, 8 * 64 * (16 << 3) * 96)


def test_automatic_choice_respects_the_budget(torch_cuda):
    """window_bits = 0: the default budget stops at the 96-GiB table however much HBM is free; table_budget_bytes moves the
    cap down; KZG_CFG_TABLE_MAX lifts it (bench.py opts in -- exercised by test_gpu_headline_shapes.py)"""
    import kateth_amd

    free, _ = torch_cuda.cuda.mem_get_info()
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, table_budget_bytes=20 << 30)
    try:
        assert (s.window_bits, s.plane_groups) == ((16, 16) if free >= 21 << 30 else (8, 16))
    finally:
        s.close()
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, table_budget_bytes=1 << 30)
    try:
        assert (s.window_bits, s.plane_groups) == (8, 16)
    finally:
        s.close()
    # the operator's cap on an unconfigured drop-in (round 5): KATETH_AMD_TABLE_BUDGET_GIB, beaten by an explicit field
    os.environ["KATETH_AMD_TABLE_BUDGET_GIB"] = "16"
    try:
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP)
        try:
            assert (s.window_bits, s.plane_groups) == ((16, 16) if free >= 21 << 30 else (8, 16))
        finally:
            s.close()
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP, table_budget_bytes=1 << 30)
        try:
            assert (s.window_bits, s.plane_groups) == (8, 16)
        finally:
            s.close()
    finally:
        del os.environ["KATETH_AMD_TABLE_BUDGET_GIB"]
    if free >= 136 << 30:
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP)
        try:
            assert (s.window_bits, s.plane_groups, s.table_bytes) == (22, 4, 96 << 30)
        finally:
            s.close()
            torch_cuda.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# KZG_CFG_BUILD_ASYNC: usable at once on the first-use table, identical results after the swap
# ---------------------------------------------------------------------------------------------------------------------
def test_background_table_build_swaps_without_changing_results(triples, golden):
    import time

    import kateth_amd

    hb, hc, hp, n = triples
    t0 = time.time()
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=16, build_async=True)
    t_create = time.time() - t0
    try:
        seen = set()
        rounds = 0
        while True:  # commitments and proofs while the 12.9-GB table is being built, and after it is in
            ready_before = s.ready
            assert s.blob_to_commitment_batch(hb[: 20 * 131072]) == (hc[: 20 * 48], [0] * 20)
            assert s.compute_blob_proof_batch(hb[: 6 * 131072], hc[: 6 * 48]) == (hp[: 6 * 48], [0] * 6)
            assert s.blob_to_commitment(hb[:131072]) == hc[:48]
            seen.add(s.window_bits)
            rounds += 1
            if ready_before:
                break
            assert rounds < 2000
        s.wait_ready()
        assert s.ready and s.window_bits == 16 and s.table_bytes == 16 * 64 * (4 << 15) * 96
        assert seen <= {8, 16} and 16 in seen
        assert s.blob_to_commitment_batch(hb) == (hc, [0] * n)
        print("async create %.2f s, classes seen %r over %d rounds" % (t_create, sorted(seen), rounds))
    finally:
        s.close()
    # destroying a context while its table is still being built cancels the build
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=16, build_async=True)
    assert s.blob_to_commitment(hb[:131072]) == hc[:48]
    s.close()
    # a blocking context is always ready
    s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=8, build_async=True)
    assert s.ready and s.window_bits == 8
    s.close()


def test_background_build_failure_leaves_a_working_context(triples, torch_cuda):
    """an explicit class-22 table that cannot be allocated (the card is filled up to ~30 GiB free): with KZG_CFG_BUILD_ASYNC
    creation still succeeds on the first-use table, kzg_ctx_wait_ready reports the failure, and the context keeps answering
    correctly on the table it has; without the flag the same request fails at creation"""
    import kateth_amd

    torch = torch_cuda
    hb, hc, hp, n = triples
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info()[0]
    if free < 40 << 30:
        pytest.skip("needs a mostly empty card")
    ballast = torch.empty(free - (30 << 30), dtype=torch.uint8, device="cuda")
    try:
        with pytest.raises(kateth_amd.kzg.EngineError):
            kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=22, plane_groups=4)
        s = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=22, plane_groups=4, build_async=True)
        try:
            assert s.blob_to_commitment_batch(hb[: 4 * 131072]) == (hc[: 4 * 48], [0] * 4)
            with pytest.raises(kateth_amd.kzg.EngineError):
                s.wait_ready()
            assert s.ready and s.window_bits == 8
            assert s.compute_blob_proof_batch(hb[: 4 * 131072], hc[: 4 * 48]) == (hp[: 4 * 48], [0] * 4)
        finally:
            s.close()
    finally:
        del ballast
        torch.cuda.empty_cache()


def test_group_with_background_build(triples):
    """both flags together: a group whose members come up on their first-use tables and swap to class 16 behind the calls"""
    import kateth_amd

    hb, hc, hp, n = triples
    g = kateth_amd.Setup.load_json(TRUSTED_SETUP, window_bits=16, devices=[0, 0], build_async=True)
    try:
        assert g.members == 2
        assert g.blob_to_commitment_batch(hb) == (hc, [0] * n)
        g.wait_ready()
        assert g.ready and g.window_bits == 16 and g.member(1).window_bits == 16
        assert g.compute_blob_proof_batch(hb, hc) == (hp, [0] * n)
        assert g.verify_blob_proof_batch_host(hb, hc, hp, n) is True
    finally:
        g.close()
