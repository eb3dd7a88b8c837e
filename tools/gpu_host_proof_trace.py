#!/usr/bin/env python3
"""Host-buffer blob proofs under rocprofv3 --kernel-trace: a few calls of kzg_compute_blob_proof_batch on pageable host blobs.
usage: gpu_host_proof_trace.py [n] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=0, table_max=True)
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
blobs = d_blobs.cpu().numpy().tobytes()
cs = d_c.cpu().numpy().tobytes()
times = []
for _ in range(reps):
    t0 = time.perf_counter()
    s.compute_blob_proof_batch(blobs, cs)
    times.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"n": n, "call_ms": times}))
