#!/usr/bin/env python3
"""Generates tests/golden/spec-layout-vectors/: ORACLE-PRODUCED vectors laid out exactly like the reference's
conformance vectors (consensus-spec-tests, tests/general/deneb/kzg/<handler>/kzg-mainnet/<case>/data.yaml;
src/kzg/setup.rs:305-317, shapes of src/kzg/spec.rs:20-220, null-output convention of src/kzg/setup.rs:330-337).

These are NOT the official vectors (that submodule is empty in the reference checkout and there is no network): the
outputs come from oracle/pyref.  Their purpose is to run the spec-test RUNNER (tests/test_spec_vectors.py) end to end --
on the oracle and on the GPU engine -- over the categories the official suite has (valid inputs, every class of
invalid input, length mismatches, the empty batch), so that dropping the official directory in (KZG_SPEC_TESTS=...)
is a data change only.  Files are gzip-compressed YAML (data.yaml.gz); most blobs are structured so they compress.
"""
import gzip
import os
import shutil
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyref import bls, synth  # noqa: E402
from oracle.pyref.setup import KzgError, Setup  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "spec-layout-vectors")
R, P = bls.R, bls.P


def hx(b):
    return "0x" + bytes(b).hex()


def be32(v):
    return int(v).to_bytes(32, "big")


def blob_of(f):
    return b"".join(be32(f(i) % R) for i in range(4096))


BLOBS = {
    "zero": bytes(131072),
    "ones": be32(1) * 4096,
    "twos": be32(2) * 4096,
    "ramp": blob_of(lambda i: i),
    "sparse": blob_of(lambda i: (0xDEADBEEF * (i + 1)) if i % 512 == 7 else 0),
    "max": be32(R - 1) * 4096,
    "random": synth.blob_bytes(synth.DEFAULT_SEED, 0),
}


def invalid_blobs():
    b = bytearray(BLOBS["ramp"])
    b[0:32] = be32(R)
    yield "element_equal_to_modulus", bytes(b)
    b = bytearray(BLOBS["ramp"])
    b[32 * 4095:] = b"\xff" * 32
    yield "last_element_all_ones", bytes(b)
    yield "one_byte_short", BLOBS["ramp"][:-1]
    yield "one_byte_long", BLOBS["ramp"] + b"\x00"
    yield "empty", b""


def find_points():
    x = 1
    while bls._fp_sqrt(x**3 + 4) is not None:
        x += 1
    not_on_curve = bytes([0x80]) + x.to_bytes(48, "big")[1:]
    x = 1
    while True:
        y = bls._fp_sqrt(x**3 + 4)
        if y is not None and not bls.g1_in_subgroup((x, y)):
            break
        x += 1
    not_in_group = bls.g1_compress((x, y))
    return not_on_curve, not_in_group


def write_case(handler, name, inp, output):
    d = os.path.join(OUT, "tests", "general", "deneb", "kzg", handler, "kzg-mainnet", "%s_case_%s" % (handler, name))
    os.makedirs(d, exist_ok=True)
    with gzip.GzipFile(os.path.join(d, "data.yaml.gz"), "wb", mtime=0) as fh:
        fh.write(yaml.safe_dump({"input": inp, "output": output}, default_flow_style=False).encode())
    print(handler, name, "null" if output is None else (output if isinstance(output, bool) else "ok"), flush=True)


def attempt(fn):
    """the reference's Result: a value, or None for Err (src/kzg/setup.rs:334-337)"""
    try:
        return fn()
    except (KzgError, Exception) as err:  # noqa: BLE001
        if type(err).__name__ in ("KzgError", "BlobError", "ECGroupError", "FiniteFieldError", "BlsError"):
            return None
        raise


def main():
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    s = Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), subgroup_checks=False)
    not_on_curve, not_in_group = find_points()
    gen = bls.g1_compress(bls.G1_GEN)
    inf = bls.g1_compress(None)
    bad_points = {"not_compressed_flag": bytes([gen[0] & 0x7F]) + gen[1:], "x_not_below_modulus": bytes([0x9A]) + b"\xff" * 47,
                  "infinity_with_sign_bit": bytes([0xE0]) + bytes(47), "not_on_curve": not_on_curve, "not_in_group": not_in_group}
    commitments, proofs = {}, {}
    # ---- blob_to_kzg_commitment --------------------------------------------------------------------------------------
    h = "blob_to_kzg_commitment"
    for name, blob in BLOBS.items():
        commitments[name] = bls.g1_compress(s.blob_to_commitment(blob))
        write_case(h, "valid_blob_" + name, {"blob": hx(blob)}, hx(commitments[name]))
    for name, blob in invalid_blobs():
        assert attempt(lambda: s.blob_to_commitment(blob)) is None
        write_case(h, "invalid_blob_" + name, {"blob": hx(blob)}, None)
    # ---- compute_blob_kzg_proof --------------------------------------------------------------------------------------
    h = "compute_blob_kzg_proof"
    for name in ("zero", "twos", "ramp", "sparse", "random"):
        proofs[name] = bls.g1_compress(s.blob_proof(BLOBS[name], commitments[name]))
        write_case(h, "valid_blob_" + name, {"blob": hx(BLOBS[name]), "commitment": hx(commitments[name])}, hx(proofs[name]))
    for pname, pt in bad_points.items():
        assert attempt(lambda: s.blob_proof(BLOBS["ramp"], pt)) is None
        write_case(h, "invalid_commitment_" + pname, {"blob": hx(BLOBS["ramp"]), "commitment": hx(pt)}, None)
    write_case(h, "invalid_commitment_47_bytes", {"blob": hx(BLOBS["ramp"]), "commitment": hx(commitments["ramp"][:47])}, None)
    for name, blob in list(invalid_blobs())[:3]:
        write_case(h, "invalid_blob_" + name, {"blob": hx(blob), "commitment": hx(commitments["ramp"])}, None)
    # ---- compute_kzg_proof -------------------------------------------------------------------------------------------
    h = "compute_kzg_proof"
    zs = {"zero": 0, "one": 1, "max": R - 1, "root_of_unity": s.roots_of_unity_brp[3], "arbitrary": 0x5EB7004FE57383E6C88B99D839937FDDF3F99279353ADA21A1F0EE1F2B7F7E2A % R}
    kzg = {}
    for bname in ("ramp", "random", "twos"):
        for zname, z in zs.items():
            pi, y = s.proof(BLOBS[bname], be32(z))
            kzg[(bname, zname)] = (bls.g1_compress(pi), bls.fr_to_be_bytes(y))
            write_case(h, "valid_blob_%s_z_%s" % (bname, zname), {"blob": hx(BLOBS[bname]), "z": hx(be32(z))}, [hx(kzg[(bname, zname)][0]), hx(kzg[(bname, zname)][1])])
    for zname, zb in (("equal_to_modulus", be32(R)), ("all_ones", b"\xff" * 32), ("31_bytes", be32(5)[1:]), ("33_bytes", be32(5) + b"\x00")):
        write_case(h, "invalid_z_" + zname, {"blob": hx(BLOBS["ramp"]), "z": hx(zb)}, None)
    for name, blob in list(invalid_blobs())[:3]:
        write_case(h, "invalid_blob_" + name, {"blob": hx(blob), "z": hx(be32(7))}, None)
    # ---- verify_kzg_proof --------------------------------------------------------------------------------------------
    h = "verify_kzg_proof"
    for (bname, zname), (pi, y) in kzg.items():
        if bname == "twos" and zname != "arbitrary":
            continue
        c, z = commitments[bname], be32(zs[zname])
        assert s.verify_proof(pi, c, z, y) is True
        write_case(h, "correct_proof_%s_%s" % (bname, zname), {"commitment": hx(c), "z": hx(z), "y": hx(y), "proof": hx(pi)}, True)
    pi, y = kzg[("ramp", "arbitrary")]
    c, z = commitments["ramp"], be32(zs["arbitrary"])
    wrong_pi = bls.g1_compress(bls.g1_add(bls.g1_uncompress(pi), bls.G1_GEN))
    wrong_y = be32((int.from_bytes(y, "big") + 1) % R)
    for name, args in (("incorrect_proof", (wrong_pi, c, z, y)), ("incorrect_y", (pi, c, z, wrong_y)), ("incorrect_z", (pi, c, be32(zs["one"]), y)),
                       ("incorrect_commitment", (pi, commitments["random"], z, y)), ("proof_is_infinity_for_a_non_constant_polynomial", (inf, c, z, y))):
        out = s.verify_proof(*args)
        assert out is False
        write_case(h, name, {"commitment": hx(args[1]), "z": hx(args[2]), "y": hx(args[3]), "proof": hx(args[0])}, False)
    write_case(h, "correct_proof_constant_polynomial_infinity_proof", {"commitment": hx(commitments["twos"]), "z": hx(z), "y": hx(be32(2)), "proof": hx(inf)},
               s.verify_proof(inf, commitments["twos"], z, be32(2)))
    for pname, pt in bad_points.items():
        write_case(h, "invalid_commitment_" + pname, {"commitment": hx(pt), "z": hx(z), "y": hx(y), "proof": hx(pi)}, None)
        write_case(h, "invalid_proof_" + pname, {"commitment": hx(c), "z": hx(z), "y": hx(y), "proof": hx(pt)}, None)
    write_case(h, "invalid_z_equal_to_modulus", {"commitment": hx(c), "z": hx(be32(R)), "y": hx(y), "proof": hx(pi)}, None)
    write_case(h, "invalid_y_equal_to_modulus", {"commitment": hx(c), "z": hx(z), "y": hx(be32(R)), "proof": hx(pi)}, None)
    write_case(h, "invalid_y_31_bytes", {"commitment": hx(c), "z": hx(z), "y": hx(y[1:]), "proof": hx(pi)}, None)
    write_case(h, "invalid_proof_49_bytes", {"commitment": hx(c), "z": hx(z), "y": hx(y), "proof": hx(pi + b"\x00")}, None)
    # ---- verify_blob_kzg_proof ---------------------------------------------------------------------------------------
    h = "verify_blob_kzg_proof"
    for name in ("zero", "twos", "ramp", "random"):
        assert s.verify_blob_proof(BLOBS[name], commitments[name], proofs[name]) is True
        write_case(h, "correct_proof_" + name, {"blob": hx(BLOBS[name]), "commitment": hx(commitments[name]), "proof": hx(proofs[name])}, True)
    write_case(h, "incorrect_proof", {"blob": hx(BLOBS["ramp"]), "commitment": hx(commitments["ramp"]), "proof": hx(proofs["random"])},
               s.verify_blob_proof(BLOBS["ramp"], commitments["ramp"], proofs["random"]))
    write_case(h, "incorrect_commitment", {"blob": hx(BLOBS["ramp"]), "commitment": hx(commitments["sparse"]), "proof": hx(proofs["ramp"])},
               s.verify_blob_proof(BLOBS["ramp"], commitments["sparse"], proofs["ramp"]))
    for pname, pt in bad_points.items():
        write_case(h, "invalid_commitment_" + pname, {"blob": hx(BLOBS["ramp"]), "commitment": hx(pt), "proof": hx(proofs["ramp"])}, None)
        write_case(h, "invalid_proof_" + pname, {"blob": hx(BLOBS["ramp"]), "commitment": hx(commitments["ramp"]), "proof": hx(pt)}, None)
    for name, blob in list(invalid_blobs())[:3]:
        write_case(h, "invalid_blob_" + name, {"blob": hx(blob), "commitment": hx(commitments["ramp"]), "proof": hx(proofs["ramp"])}, None)
    # ---- verify_blob_kzg_proof_batch ---------------------------------------------------------------------------------
    h = "verify_blob_kzg_proof_batch"
    names = ["ramp", "random", "sparse", "twos", "zero"]

    def batch(ns, cs=None, ps=None, blobs=None):
        return {"blobs": [hx(b) for b in (blobs if blobs is not None else [BLOBS[n] for n in ns])],
                "commitments": [hx(c) for c in (cs if cs is not None else [commitments[n] for n in ns])],
                "proofs": [hx(p) for p in (ps if ps is not None else [proofs[n] for n in ns])]}

    write_case(h, "0_blobs", batch([]), s.verify_blob_proof_batch([], [], []))
    for k in (1, 2, 5):
        ns = names[:k]
        assert s.verify_blob_proof_batch([BLOBS[n] for n in ns], [commitments[n] for n in ns], [proofs[n] for n in ns]) is True
        write_case(h, "%d_blobs" % k, batch(ns), True)
    ns = names[:3]
    swapped = [proofs[ns[1]], proofs[ns[0]], proofs[ns[2]]]
    write_case(h, "incorrect_proofs_swapped", batch(ns, ps=swapped), s.verify_blob_proof_batch([BLOBS[n] for n in ns], [commitments[n] for n in ns], swapped))
    write_case(h, "incorrect_commitment_in_last_position", batch(ns, cs=[commitments[ns[0]], commitments[ns[1]], commitments["zero"]]),
               s.verify_blob_proof_batch([BLOBS[n] for n in ns], [commitments[ns[0]], commitments[ns[1]], commitments["zero"]], [proofs[n] for n in ns]))
    for pname, pt in list(bad_points.items())[:3]:
        write_case(h, "invalid_commitment_" + pname, batch(ns, cs=[commitments[ns[0]], pt, commitments[ns[2]]]), None)
        write_case(h, "invalid_proof_" + pname, batch(ns, ps=[proofs[ns[0]], proofs[ns[1]], pt]), None)
    bad_blob = list(invalid_blobs())[0][1]
    write_case(h, "invalid_blob_in_the_middle", batch(ns, blobs=[BLOBS[ns[0]], bad_blob, BLOBS[ns[2]]]), None)
    write_case(h, "blob_length_different", batch(ns, blobs=[BLOBS[ns[0]], BLOBS[ns[1]][:-1], BLOBS[ns[2]]]), None)
    write_case(h, "commitments_length_mismatch", batch(ns, cs=[commitments[n] for n in ns[:2]]), None)
    write_case(h, "proofs_length_mismatch", batch(ns, ps=[proofs[n] for n in ns] + [proofs["zero"]]), None)
    with open(os.path.join(OUT, "README.md"), "w") as fh:
        fh.write("ORACLE-GENERATED vectors in the consensus-spec-tests layout (see ../make_spec_layout_vectors.py).\n"
                 "They are NOT the official vectors: outputs come from oracle/pyref.  The official suite drops in with KZG_SPEC_TESTS=<dir>.\n")


if __name__ == "__main__":
    main()
