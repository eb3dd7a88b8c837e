"""CPU oracle for the kateth hot path -- TEST INFRASTRUCTURE, never imported by
the product package `kateth_amd`.  Module layout mirrors the reference:

    bls.py    <- src/bls.rs          domain.py <- src/math.rs
    blob.py   <- src/blob.rs         poly.py   <- src/kzg/poly.rs
    setup.py  <- src/kzg/setup.rs

PARITY STATUS: parity unpinned (no reference-owned golden vectors exist in this
checkout; see bls.py header and DESIGN.md).  Pinned instead by public constants
and identities over trusted_setup_4096.json -- tests/test_oracle_kat.py.
"""
