#!/usr/bin/env python3
"""One blob through blob_to_commitment, blob_proof and verify_blob_proof (host API, the shape benches/kzg.rs:35-43 times) under
rocprofv3 --kernel-trace: a few warm calls of each, 20 ms apart, so that tools/trace_timeline_with_copies.py shows the last call
of each kind as a separate group.   usage: gpu_single_item_trace.py [commit|proof|verify]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "verify"
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=0, table_max=True)
d = torch.empty(131072, dtype=torch.uint8, device="cuda")
s.synth_blobs_dev(0x4844, 0, 1, d.data_ptr())
torch.cuda.synchronize()
blob = d.cpu().numpy().tobytes()
c = s.blob_to_commitment(blob)
p = s.blob_proof(blob, c)
assert s.verify_blob_proof(blob, c, p) is True
fn = {"commit": lambda: s.blob_to_commitment(blob), "proof": lambda: s.blob_proof(blob, c), "verify": lambda: s.verify_blob_proof(blob, c, p)}[what]
times = []
for _ in range(6):
    time.sleep(0.02)
    t0 = time.perf_counter()
    fn()
    times.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"what": what, "call_ms": times}))
