#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) throughput of the batch entry points: the caller's blobs live in host memory, as in
kateth's byte-slice API.  Pageable memory (a Python bytes object) and pinned memory (torch pin_memory) are both timed;
the device-resident rate of the same batch is printed beside them.  usage: gpu_hostapi_bench.py [n] [window_bits]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = int(sys.argv[2]) if len(sys.argv) > 2 else 12
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c, table_max=True)
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
pin_b = torch.empty(n * 131072, dtype=torch.uint8, pin_memory=True)
pin_c = torch.empty(n * 48, dtype=torch.uint8, pin_memory=True)
pin_p = torch.empty(n * 48, dtype=torch.uint8, pin_memory=True)
pin_b.copy_(d_blobs)
pin_c.copy_(d_c)
pin_p.copy_(d_p)
torch.cuda.synchronize()
blobs = pin_b.numpy().tobytes()  # pageable copies
cs, ps = pin_c.numpy().tobytes(), pin_p.numpy().tobytes()
out = {"n": n, "window_bits": c}


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def rec(name, t):
    out[name + "_ms"], out[name + "_blobs_per_s"] = 1e3 * t, n / t


rec("commit_host_pageable", timed(lambda: s.blob_to_commitment_batch(blobs, n)))
rec("commit_dev", timed(lambda: (s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr()), torch.cuda.synchronize())))
got, st = s.blob_to_commitment_batch(blobs, n)
assert got == cs and not any(st)
rec("proof_host_pageable", timed(lambda: s.compute_blob_proof_batch(blobs, cs)))
rec("proof_dev", timed(lambda: (s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr()), torch.cuda.synchronize())))
assert s.verify_blob_proof_batch_host(blobs, cs, ps, n) is True
assert s.verify_blob_proof_batch_host(pin_b.data_ptr(), pin_c.data_ptr(), pin_p.data_ptr(), n) is True
rec("verify_host_pageable", timed(lambda: s.verify_blob_proof_batch_host(blobs, cs, ps, n)))
rec("verify_host_pinned", timed(lambda: s.verify_blob_proof_batch_host(pin_b.data_ptr(), pin_c.data_ptr(), pin_p.data_ptr(), n)))
rec("verify_dev", timed(lambda: s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n)))
# raw copy rates for reference
t = timed(lambda: (d_blobs.copy_(pin_b, non_blocking=True), torch.cuda.synchronize()))
out["h2d_pinned_GBps"] = n * 131072 / t / 1e9
src = torch.frombuffer(bytearray(blobs), dtype=torch.uint8)
t = timed(lambda: (d_blobs.copy_(src), torch.cuda.synchronize()))
out["h2d_pageable_GBps"] = n * 131072 / t / 1e9
# single-item host API latencies
one_b, one_c, one_p = blobs[:131072], cs[:48], ps[:48]
for name, fn in (("single_commit_ms", lambda: s.blob_to_commitment(one_b)), ("single_proof_ms", lambda: s.blob_proof(one_b, one_c)),
                 ("single_verify_blob_ms", lambda: s.verify_blob_proof(one_b, one_c, one_p))):
    fn()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    out[name] = 1e3 * (time.perf_counter() - t0) / 10
print(json.dumps(out))
s.close()
