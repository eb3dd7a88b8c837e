#!/usr/bin/env python3
"""Soak of the whole path across the batch-size regimes (kernel selection changes at 64 / 4,096 / 8,192 / 16,384 / 32,768
items): random batches -> commit -> prove -> verify must accept; one corrupted byte in a random blob, commitment or proof must
make it reject (or raise the reference's decoding error); the flat and the classic variable-base MSM must return the same two
partial sums.  usage: gpu_soak_verify.py [rounds] [window_bits]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
s = kateth_amd.Setup.load_json(SETUP, window_bits=cls)
os.environ["KATETH_AMD_VAR_MSM"] = "classic"
classic = kateth_amd.Setup.load_json(SETUP, window_bits=8)
del os.environ["KATETH_AMD_VAR_MSM"]
rnd = random.Random(0x50AC)
sizes = [1, 2, 63, 64, 65, 130, 1000, 4095, 4096, 4097, 8192, 9000, 16384, 16500, 32767, 32768, 33001, 50000, 65536, 100000, 131072]  # from 32,768 on: the balanced bucket kernel (E = 16 ... 65 entries per lane)
bad = 0
t0 = time.time()
for rd in range(rounds):
    for n in sizes:
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        s.synth_blobs_dev(rnd.getrandbits(48), rnd.getrandbits(20), n, d_blobs.data_ptr())
        d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
        s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize()
        ok = int(d_st.abs().sum()) == 0 and s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is True
        sums = []
        for e in (s, classic):
            sess, root, err = e.verify_phase1_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)
            sums.append(e.verify_phase2_dev(sess, root, 0, n))
            e.verify_session_destroy(sess)
        ok = ok and sums[0] == sums[1]
        # one corrupted byte
        which = rnd.choice(("blob", "commitment", "proof"))
        i = rnd.randrange(n)
        if which == "blob":
            off = i * 131072 + 32 * rnd.randrange(4096) + 31  # low byte of an element: stays canonical
            d_blobs[off] ^= 1 << rnd.randrange(8)
        else:
            t = d_c if which == "commitment" else d_p
            t[i * 48 + 1 + rnd.randrange(47)] ^= 1 << rnd.randrange(8)
        torch.cuda.synchronize()
        try:
            rejected = s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n) is False
        except kateth_amd.kzg.KzgError:
            rejected = True  # the corrupted encoding no longer decodes: the reference returns Err too
        if not (ok and rejected):
            bad += 1
            print("FAIL round %d n=%d corrupted %s[%d]: accept-valid %s reject-corrupt %s" % (rd, n, which, i, ok, rejected), flush=True)
        del d_blobs, d_c, d_p, d_st
    print("round %d done (%.0f s), failures so far: %d" % (rd + 1, time.time() - t0, bad), flush=True)
print("TOTAL: %d rounds x %d batch sizes: %d failures" % (rounds, len(sizes), bad))
s.close()
classic.close()
sys.exit(1 if bad else 0)
