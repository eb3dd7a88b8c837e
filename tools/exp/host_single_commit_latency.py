#!/usr/bin/env python3
"""One blob through the host-buffer blob_to_commitment, 300 times; prints the mean latency.  Run under variations of
GPU_MAX_HW_QUEUES / KATETH_AMD_* to see what the host path's overhead over the device-resident call (0.52 ms) is made of."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import kateth_amd  # noqa: E402

s = kateth_amd.Setup.load_json(os.path.join(os.path.dirname(kateth_amd.__file__), "..", "tests", "golden", "trusted_setup_4096.json"), window_bits=int(os.environ.get("WB", "0")))
d = torch.empty(131072, dtype=torch.uint8, device="cuda")
s.synth_blobs_dev(7, 0, 1, d.data_ptr())
torch.cuda.synchronize()
blob = d.cpu().numpy().tobytes()
for _ in range(20):
    c = s.blob_to_commitment(blob)
t0 = time.perf_counter()
for _ in range(300):
    s.blob_to_commitment(blob)
dt = (time.perf_counter() - t0) / 300
d_c = torch.empty(48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(1, dtype=torch.int32, device="cuda")
for _ in range(20):
    s.blob_to_commitment_batch_dev(d.data_ptr(), 1, d_c.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    s.blob_to_commitment_batch_dev(d.data_ptr(), 1, d_c.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
dd = (time.perf_counter() - t0) / 300
# raw copies for scale: 128 KiB up, 48 B down on one stream
h = torch.from_numpy(__import__("numpy").frombuffer(blob, dtype="uint8").copy())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    d.copy_(h)
    torch.cuda.synchronize()
up = (time.perf_counter() - t0) / 300
t0 = time.perf_counter()
for _ in range(300):
    d_c.cpu()
dn = (time.perf_counter() - t0) / 300
print("queues=%s table=%s host %.3f ms  device-resident %.3f ms  | pageable 128 KiB up %.3f ms, 48 B down %.3f ms" % (os.environ.get("GPU_MAX_HW_QUEUES"), s.window_bits, 1e3 * dt, 1e3 * dd, 1e3 * up, 1e3 * dn))
