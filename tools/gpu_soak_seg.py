#!/usr/bin/env python3
"""Soak of the balanced bucket kernel (k_var_buckets_seg + k_var_seg_fixup, the default flat path) against one thread per bucket
(KATETH_AMD_VAR_SEG=0): random batch sizes in [32,768, 140,000] -- share sizes 16 ... 70 -- with repeated triples, zero blobs (points at
infinity) and random share sizes forced through KATETH_AMD_VAR_SEG=E; the two 96-byte partial sums of phase 2 must be identical and both
paths must accept.  usage: gpu_soak_seg.py [batches]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
s = kateth_amd.Setup.load_json(SETUP, window_bits=22)
os.environ["KATETH_AMD_VAR_SEG"] = "0"
ref = kateth_amd.Setup.load_json(SETUP, window_bits=8)
rnd = random.Random(0x5E6)
bad = 0
t0 = time.time()
NMAX = 140000
d_all = torch.empty(NMAX * 131072, dtype=torch.uint8, device="cuda")
for b in range(batches):
    n = rnd.choice([32768, 32769, 65536, 131072, rnd.randrange(32768, NMAX), rnd.randrange(32768, 70000)])
    d_blobs = d_all[: n * 131072]
    s.synth_blobs_dev(rnd.getrandbits(48), rnd.getrandbits(20), n, d_blobs.data_ptr())
    v = d_blobs.view(n, 131072)
    for _ in range(rnd.randrange(0, 4)):  # repeated triples, zero blobs
        i, j = rnd.randrange(n), rnd.randrange(n)
        v[i] = v[j].clone()
    for _ in range(rnd.randrange(0, 3)):
        v[rnd.randrange(n)] = 0
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    E = rnd.choice([1, 1, 16, 17, 23, 40, 64, 129])
    os.environ["KATETH_AMD_VAR_SEG"] = str(E)
    forced = kateth_amd.Setup.load_json(SETUP, window_bits=8)
    sums = []
    ok = int(d_st.abs().sum()) == 0
    for e in (s, ref, forced):
        sess, root, err = e.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
        sums.append(e.verify_phase2_dev(sess, root, 0, n))
        e.verify_session_destroy(sess)
        ok = ok and e.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
    forced.close()
    if not (ok and sums[0] == sums[1] == sums[2]):
        bad += 1
        print("MISMATCH: n = %d, forced E = %d" % (n, E), flush=True)
    if b % 5 == 4:
        print("batch %d/%d (%.0f s), mismatches so far: %d" % (b + 1, batches, time.time() - t0, bad), flush=True)
print("TOTAL: %d batches, sizes 32,768 ... %d: %d mismatches" % (batches, NMAX, bad))
s.close()
ref.close()
sys.exit(1 if bad else 0)
