#!/usr/bin/env python3
"""Soak: commitments (and proofs) of many random batches through the product's comb MSM kernel against round 1's window-table
kernel on 12 x 32-bit limbs (test-only build), byte for byte.  Every batch of 4,096 blobs is 2-3.7e8 mixed additions per
engine, a few thousand of which take the out-of-line complete adder (false alarms of the cheap P == +-Q filter)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = 4096
import __graft_entry__ as g  # noqa: E402

cls = int(sys.argv[2]) if len(sys.argv) > 2 else 22
s28 = kateth_amd.Setup.load_json(SETUP, window_bits=cls)  # the product library: comb kernel (class 22: blocks of 22/21 points; 16: of 16 points)
os.environ["KATETH_AMD_MSM_RADIX"] = "32"  # honoured only by the test-only build (tests/window_msm)
s32 = kateth_amd.Setup.load_json(SETUP, window_bits=12, lib_path=g.build_test_engine())
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
bufs = [torch.empty(n * 48, dtype=torch.uint8, device="cuda") for _ in range(4)]
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
bad = 0
t0 = time.time()
for b in range(batches):
    s28.synth_blobs_dev(0x50A4 + b, b * n, n, d_blobs.data_ptr())
    s28.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, bufs[0].data_ptr(), d_st.data_ptr())
    s32.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, bufs[1].data_ptr(), d_st.data_ptr())
    s28.compute_blob_proof_batch_dev(d_blobs.data_ptr(), bufs[0].data_ptr(), n, bufs[2].data_ptr(), d_st.data_ptr())
    s32.compute_blob_proof_batch_dev(d_blobs.data_ptr(), bufs[0].data_ptr(), n, bufs[3].data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    same_c = bool(torch.equal(bufs[0], bufs[1]))
    same_p = bool(torch.equal(bufs[2], bufs[3]))
    ok = s28.verify_blob_proof_batch_dev(d_blobs.data_ptr(), bufs[0].data_ptr(), bufs[2].data_ptr(), n)
    if not (same_c and same_p and ok):
        bad += 1
        print("MISMATCH in batch %d: commitments equal %s, proofs equal %s, verify %s" % (b, same_c, same_p, ok), flush=True)
    if b % 10 == 9:
        print("batch %d/%d  (%.0f s)  mismatching batches so far: %d" % (b + 1, batches, time.time() - t0, bad), flush=True)
print("TOTAL: %d batches x %d blobs (commit + proof through both kernels, verify): %d mismatching batches" % (batches, n, bad))
s28.close()
s32.close()
sys.exit(1 if bad else 0)
