#!/usr/bin/env python3
"""Generates tests/golden/kzg_vectors.json from the CPU oracle (oracle/pyref).

The reference (kateth, Rust + blst) cannot be built or imported in the build
container and its own vectors (consensus-spec-tests) are an empty submodule, so
these vectors are oracle outputs, not reference outputs: they let the GPU box
(where neither /root/reference nor minutes of Python big-int time are wanted)
check the HIP path against committed data.  Blobs are not stored; they are
re-derived from (seed, index) by oracle.pyref.synth / the device generator."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyref import blob as oblob  # noqa: E402
from oracle.pyref import bls, poly, synth  # noqa: E402
from oracle.pyref.setup import Setup  # noqa: E402

N_BLOBS = 6


def main():
    setup = Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), subgroup_checks=False)
    out = {"seed": synth.DEFAULT_SEED, "generator": "element(b,i)=SHA256(seed_le64||b_le64||i_le32) mod r, 32B BE", "blobs": []}
    for b in range(N_BLOBS):
        data = synth.blob_bytes(synth.DEFAULT_SEED, b)
        elements = oblob.from_slice(data)
        c = oblob.commitment(elements, setup)
        c48 = bls.g1_compress(c)
        z = oblob.challenge(elements, c)
        y, pi = poly.prove(elements, z, setup)
        rec = {
            "index": b,
            "blob_sha256": __import__("hashlib").sha256(data).hexdigest(),
            "commitment": c48.hex(),
            "challenge_z": "%064x" % z,
            "eval_y": "%064x" % y,
            "proof": bls.g1_compress(pi).hex(),
        }
        # compute_kzg_proof at an arbitrary point and at an in-domain point (poly.rs:50-64 branch)
        z2 = (0x1234567890ABCDEF << 64 | b) % bls.R
        y2, pi2 = poly.prove(elements, z2, setup)
        rec["kzg_proof_at"] = {"z": "%064x" % z2, "y": "%064x" % y2, "proof": bls.g1_compress(pi2).hex()}
        if b == 0:
            zd = setup.roots_of_unity_brp[5]
            yd, pid = poly.prove(elements, zd, setup)
            rec["kzg_proof_in_domain"] = {"z": "%064x" % zd, "y": "%064x" % yd, "proof": bls.g1_compress(pid).hex()}
        out["blobs"].append(rec)
        print("blob", b, rec["commitment"][:16], rec["proof"][:16], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "kzg_vectors.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
