"""Oracle restatement of kateth `src/kzg/setup.rs`."""
import json

from . import blob as blobmod
from . import bls, domain, poly
from .bls import ECGroupError, FiniteFieldError, R
from .blob import BlobError


class KzgError(Exception):
    """`kzg::Error` (src/kzg/mod.rs:15-31): wraps a blob or bls error."""

    def __init__(self, inner: Exception):
        super().__init__("{}::{}".format(type(inner).__name__, getattr(inner, "kind", "")))
        self.inner = inner


class LoadSetupError(Exception):
    """`LoadSetupError` (src/kzg/setup.rs:20-28)."""


def _unhex(s: str) -> bytes:
    """`Bytes` deserialiser (src/bytes.rs:30-37): optional 0x prefix."""
    return bytes.fromhex(s[2:] if s.startswith("0x") else s)


class Setup:
    """`Setup<G1, G2>` (src/kzg/setup.rs:37-42)."""

    def __init__(self, g1_lagrange_brp, g2_monomial, roots_of_unity_brp):
        self.g1_lagrange_brp = g1_lagrange_brp
        self.g2_monomial = g2_monomial
        self.roots_of_unity_brp = roots_of_unity_brp

    # ---- src/kzg/setup.rs:46-82 -------------------------------------------
    @classmethod
    def load_json(cls, path, g1: int = 4096, g2: int = 65, subgroup_checks: bool = True):
        with open(path) as fh:
            raw = json.load(fh)
        if len(raw["g1_lagrange"]) != g1:
            raise LoadSetupError("InvalidLenG1Lagrange")
        if len(raw["g2_monomial"]) != g2:
            raise LoadSetupError("InvalidLenG2Monomial")
        dec1 = bls.g1_decompress if subgroup_checks else bls.g1_uncompress
        dec2 = bls.g2_decompress if subgroup_checks else bls.g2_uncompress
        try:
            g1_lagrange = [dec1(_unhex(s)) for s in raw["g1_lagrange"]]
            g1_lagrange_brp = domain.bit_reversal_permutation(g1_lagrange)
            g2_monomial = [dec2(_unhex(s)) for s in raw["g2_monomial"]]
        except ECGroupError as err:
            raise LoadSetupError("Bls({})".format(err.kind))
        roots_brp = domain.bit_reversal_permutation(domain.roots_of_unity(g1))
        return cls(g1_lagrange_brp, g2_monomial, roots_brp)

    # ---- src/kzg/setup.rs:84-94 -------------------------------------------
    def verify_proof_inner(self, proof, commitment, point: int, ev: int) -> bool:
        pairing1 = (proof, bls.g2_add(self.g2_monomial[1], bls.g2_mul(bls.g2_neg(bls.G2_GEN), point)))
        pairing2 = (bls.g1_add(commitment, bls.g1_mul(bls.g1_neg(bls.G1_GEN), ev)), bls.G2_GEN)
        return bls.verify_pairings(pairing1, pairing2)

    # ---- src/kzg/setup.rs:96-113 ------------------------------------------
    def verify_proof(self, proof48: bytes, commitment48: bytes, point32: bytes, eval32: bytes) -> bool:
        try:
            proof = bls.g1_decompress(proof48)
            commitment = bls.g1_decompress(commitment48)
            point = bls.fr_from_be_slice(point32)
            ev = bls.fr_from_be_slice(eval32)
        except (ECGroupError, FiniteFieldError) as err:
            raise KzgError(err)
        return self.verify_proof_inner(proof, commitment, point, ev)

    # ---- src/kzg/setup.rs:115-161 -----------------------------------------
    def verify_proof_batch(self, proofs, commitments, points, evals) -> bool:
        assert len(proofs) == len(commitments) == len(points) == len(evals)
        n = len(proofs)
        data = b"RCKZGBATCH___V1_" + len(self.g1_lagrange_brp).to_bytes(16, "big") + n.to_bytes(16, "big")
        r = bls.fr_hash_to(data)  # quirk Q1: does not bind the inputs
        rpowers, points_mul_rpowers, comms_minus_evals = [], [], []
        for i in range(n):
            rpower = bls.fr_pow_reference(r, i)  # quirk Q2: r.pow(0) == r
            rpowers.append(rpower)
            points_mul_rpowers.append(points[i] * rpower % R)
            comms_minus_evals.append(bls.g1_add(commitments[i], bls.g1_mul(bls.g1_neg(bls.G1_GEN), evals[i])))
        proof_lincomb = bls.g1_lincomb(proofs, rpowers)
        proof_z_lincomb = bls.g1_lincomb(proofs, points_mul_rpowers)
        comm_minus_eval_lincomb = bls.g1_lincomb(comms_minus_evals, rpowers)
        return bls.verify_pairings(
            (proof_lincomb, self.g2_monomial[1]),
            (bls.g1_add(comm_minus_eval_lincomb, proof_z_lincomb), bls.G2_GEN),
        )

    # ---- src/kzg/setup.rs:163-171 -----------------------------------------
    def blob_to_commitment(self, blob_bytes: bytes):
        elements = blobmod.from_slice(blob_bytes, len(self.g1_lagrange_brp))
        return blobmod.commitment(elements, self)

    # ---- src/kzg/setup.rs:173-183 -----------------------------------------
    def blob_proof(self, blob_bytes: bytes, commitment48: bytes):
        try:
            elements = blobmod.from_slice(blob_bytes, len(self.g1_lagrange_brp))
            commitment = bls.g1_decompress(commitment48)
        except (BlobError, ECGroupError) as err:
            raise KzgError(err)
        return blobmod.proof(elements, commitment, self)

    # ---- src/kzg/setup.rs:185-194 -----------------------------------------
    def proof(self, blob_bytes: bytes, point32: bytes):
        try:
            elements = blobmod.from_slice(blob_bytes, len(self.g1_lagrange_brp))
            point = bls.fr_from_be_slice(point32)
        except (BlobError, FiniteFieldError) as err:
            raise KzgError(err)
        ev, pi = poly.prove(elements, point, self)
        return pi, ev

    # ---- src/kzg/setup.rs:196-221 -----------------------------------------
    def verify_blob_proof(self, blob_bytes: bytes, commitment48: bytes, proof48: bytes) -> bool:
        try:
            elements = blobmod.from_slice(blob_bytes, len(self.g1_lagrange_brp))
            commitment = bls.g1_decompress(commitment48)
            proof = bls.g1_decompress(proof48)
        except (BlobError, ECGroupError) as err:
            raise KzgError(err)
        z = blobmod.challenge(elements, commitment)
        ev = poly.evaluate(elements, z, self)
        return self.verify_proof_inner(proof, commitment, z, ev)

    # ---- src/kzg/setup.rs:223-275 -----------------------------------------
    def verify_blob_proof_batch(self, blobs, commitments48, proofs48) -> bool:
        assert len(blobs) == len(commitments48) == len(proofs48)  # :256-257 (panics)
        try:
            parsed = [blobmod.from_slice(b, len(self.g1_lagrange_brp)) for b in blobs]
            commitments = [bls.g1_decompress(c) for c in commitments48]
            proofs = [bls.g1_decompress(p) for p in proofs48]
        except (BlobError, ECGroupError) as err:
            raise KzgError(err)
        challenges, evaluations = [], []
        for elements, commitment in zip(parsed, commitments):
            z = blobmod.challenge(elements, commitment)
            challenges.append(z)
            evaluations.append(poly.evaluate(elements, z, self))
        return self.verify_proof_batch(proofs, commitments, challenges, evaluations)
