#!/usr/bin/env python3
"""Single-blob calls in a loop (device-resident): run under `rocprofv3 --kernel-trace --stats` to see which kernels make up
the latency of one commitment / proof / verification.  usage: gpu_single_latency.py [window_bits] [reps] [commit|proof|verify]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

c = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c, table_max=True)
d_blob = torch.empty(131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(1, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, 1, d_blob.data_ptr())
only = sys.argv[3] if len(sys.argv) > 3 else None
out = {}
for name, fn in (("commit", lambda: s.blob_to_commitment_batch_dev(d_blob.data_ptr(), 1, d_c.data_ptr(), d_st.data_ptr())),
                 ("proof", lambda: s.compute_blob_proof_batch_dev(d_blob.data_ptr(), d_c.data_ptr(), 1, d_p.data_ptr(), d_st.data_ptr())),
                 ("verify", lambda: s.verify_blob_proof_batch_dev(d_blob.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), 1))):
    if only and name != only:
        continue
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
    out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / reps
print(out)
s.close()
