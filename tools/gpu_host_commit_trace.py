#!/usr/bin/env python3
"""Host-buffer commitment under rocprofv3 --kernel-trace (tools/gpu_host_commit_trace.sh): a few calls of
kzg_blob_to_commitment_batch on pageable host blobs, wall-clock per call on stdout; the kernel timeline of the last call
is summarised from the trace by the shell wrapper.  usage: gpu_host_commit_trace.py [n] [window_bits] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c)
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
torch.cuda.synchronize()
blobs = d_blobs.cpu().numpy().tobytes()
times = []
for _ in range(reps):
    t0 = time.perf_counter()
    s.blob_to_commitment_batch(blobs, n)
    times.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"n": n, "window_bits": s.window_bits, "call_ms": times}))
